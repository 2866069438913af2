#!/bin/bash
# r03 experiment O: prompt-row GEMMs in the split-bf16 arithmetic (gemm_tile_split_kernel)
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -s -k "not kv_fp16_report" 2>&1 | grep -E "passed|failed|error|Error|split max err|assert" | tail -20 > gpurun_out/r03_o_tests.log
echo "tests rc=$?"; tail -12 gpurun_out/r03_o_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-throughput-mode > gpurun_out/r03_o_bench.json 2> gpurun_out/r03_o_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_o_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
print(json.dumps(d['prefill_roofline'])[:600])
PY
