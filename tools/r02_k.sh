#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02k
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02k_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()}); a=j["decode_attention"]; print("attn", round(a["avg_launch_ms"]*1e3,2),"us", round(a["achieved"]), round(a["frac"],3))
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_ATTN_UN=8" run un8
ENVV="AUR_ATTN_UN=8" run un8_kvh --kv fp16
