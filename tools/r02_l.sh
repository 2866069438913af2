#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02l
run() { tag=$1; shift; env $ENVV timeout 400 python bench.py --warmup 1 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02l_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()})
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_X=0" run seq --steps 4
ENVV="AUR_X=0" run pipe --steps 4 --pipeline
ENVV="AUR_STREAM_PRIORITY=0" run pipe_noprio --steps 4 --pipeline
