#!/bin/bash
# r03 experiment AF: 128- instead of 256-position tiles in the fused round kernel at 64 channels, k = 3 / 7 (two workgroups per CU)
exec < /dev/null
mkdir -p gpurun_out
for nt in 0 128; do
AUR_ROUND_NT=$nt timeout 300 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
AUR_ROUND_NT=$nt timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --steps 2 > gpurun_out/r03_af_bench_$nt.json 2> gpurun_out/r03_af.err
python - $nt <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03_af_bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
rv=d['roofline_vocoder']
print('nt',sys.argv[1], d['ms_per_step'], d['breakdown_ms_per_step']['vocoder_convs'], [round(c['ms']/2,2) for c in rv['by_class']])
PY
done
