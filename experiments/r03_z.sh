#!/bin/bash
# r03 experiment Z: fused ResBlock round kernel on the 64- and 32-channel stages
exec < /dev/null
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|rms err|Error|assert|rror" | tail -12 > gpurun_out/r03_z_tests.log
echo "tests rc=$?"; cat gpurun_out/r03_z_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_z_bench.json 2> gpurun_out/r03_z_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_z_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
rv=d['roofline_vocoder']
print(rv['achieved'],rv['frac'],rv['avg_launch_ms'],rv.get('frac_of_binding_floors'))
for c in rv['by_class']: print(c['class'],c['launches'],round(c['ms']/3,2),round(c['hbm']['frac'],3),round(c['mfma']['frac'],3))
PY
