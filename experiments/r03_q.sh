#!/bin/bash
# r03 experiment R: transposed convs on the LDS-DMA kernel (fp16 stage inputs)
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_api.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|rms err|Error|assert" | tail -12 > gpurun_out/r03_r_tests.log
echo "tests rc=$?"; cat gpurun_out/r03_r_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_r_bench.json 2> gpurun_out/r03_r_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_r_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
rv=d['roofline_vocoder']
print(rv['achieved'],rv['frac'],rv['avg_launch_ms'],rv.get('frac_of_binding_floors'))
for c in rv['by_class']: print(c['class'],c['launches'],round(c['ms']/3,2),round(c['hbm']['frac'],3),round(c['mfma']['frac'],3))
PY
