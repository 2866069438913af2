#!/bin/bash
# r03 experiment X: tile shape of the narrow (N = 1024) prompt-row GEMMs
exec < /dev/null
mkdir -p gpurun_out
for t in 0 1; do
AUR_GT_TILE=$t timeout 600 python bench.py --no-cpu-baseline --no-throughput-mode --steps 2 > gpurun_out/r03_x_bench_$t.json 2> gpurun_out/r03_x_bench.err
python - $t <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03_x_bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('tile',sys.argv[1], d['ms_per_step'], d['breakdown_ms_per_step']['gpt_prefill'])
PY
done
