#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
for bsz in 1 8; do
timeout 300 python bench.py --batch $bsz --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/r02z_b$bsz.json 2> gpurun_out/r02z.err
python - $bsz <<'PY'
import json,sys
j=json.load(open("gpurun_out/r02z_b%s.json"%sys.argv[1])); b=j["breakdown_ms_per_step"]
print("batch",sys.argv[1], "ms per batch", round(j["ms_per_step"],1), "rtf", round(j["rtf"],5), "samples/s", round(j["value"]), {k:round(v,3) for k,v in b.items()})
PY
done
