#!/bin/bash
# r03 experiment W: after removing the uninstantiated conv variants: vocoder tests (+ DMA vs register staging equality), bench
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 > gpurun_out/r03_w_tests.log
echo "tests rc=$?"; cat gpurun_out/r03_w_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_w_bench.json 2> gpurun_out/r03_w_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_w_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
PY
