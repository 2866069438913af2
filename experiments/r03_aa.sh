#!/bin/bash
# r03 experiment AA: fused round kernel with 32 instead of 64 positions per wave (twice the waves per workgroup)
exec < /dev/null
mkdir -p gpurun_out
for w in 0 1 2 3; do
AUR_ROUND_WN1=$w timeout 300 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
AUR_ROUND_WN1=$w timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --steps 2 > gpurun_out/r03_aa_bench_$w.json 2> gpurun_out/r03_aa_bench.err
python - $w <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03_aa_bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
rv=d['roofline_vocoder']
print('wn1',sys.argv[1], d['ms_per_step'], d['breakdown_ms_per_step']['vocoder_convs'], [round(c['ms']/2,2) for c in rv['by_class']])
PY
done
