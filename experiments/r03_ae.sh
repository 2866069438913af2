#!/bin/bash
# r03 experiment AE: bench consumes the results as views of the pinned result blocks (poll(copy=False))
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_ae_bench.json 2> gpurun_out/r03_ae.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_ae_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['breakdown_ms_per_step'])
PY
