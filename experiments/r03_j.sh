#!/bin/bash
# r03 experiment J: fp16 residual stream in the vocoder ResBlocks -- parity tests, conv_diag, bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r03_j_tests.log
echo "tests rc=$?"; tail -12 gpurun_out/r03_j_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/conv_diag.hip -o /tmp/conv_diag && timeout 300 /tmp/conv_diag > gpurun_out/r03_j_conv_diag.log 2>&1
cat gpurun_out/r03_j_conv_diag.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03_j_bench.json 2> gpurun_out/r03_j_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_j_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('roofline_vocoder'))[:1500])
PY
