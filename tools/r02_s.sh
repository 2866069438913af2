#!/bin/bash
# attention A/B: (steps unrolled, next-iteration prefetch) combinations; conditioning tests after the channel-mean fix
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02s
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02s_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); b=j["breakdown_ms_per_step"]; a=j["decode_attention"]; print(sys.argv[1], round(j["ms_per_step"],1), "decode step", round(b["gpt_ms_per_decode_step"],3), "attn", round(a["avg_launch_ms"]*1e3,2),"us", round(a["achieved"]), round(a["frac"],3))
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_ATTN_UN=4 AUR_ATTN_PREFETCH=0" run un4_pf0
ENVV="AUR_ATTN_UN=2 AUR_ATTN_PREFETCH=1" run un2_pf1
ENVV="AUR_ATTN_UN=2 AUR_ATTN_PREFETCH=0" run un2_pf0
ENVV="AUR_ATTN_UN=2 AUR_ATTN_PREFETCH=1" run un2_pf1_kvh --kv fp16
ENVV="AUR_ATTN_UN=4 AUR_ATTN_PREFETCH=0" run un4_pf0_kvh --kv fp16
timeout 600 python -m pytest tests/test_conditioning.py -m gpu -x -q -s 2>&1 | grep -i "hip\|passed\|failed\|clip" 
