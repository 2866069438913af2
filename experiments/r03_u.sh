#!/bin/bash
# r03 experiment U: 16-way top-k search in the sampler, finished flag folded into the token read-back word
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "not kv_fp16_report" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12 > gpurun_out/r03_u_tests.log
echo "tests rc=$?"; tail -8 gpurun_out/r03_u_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_u_bench.json 2> gpurun_out/r03_u_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_u_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
PY
