#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02c
bash tools/gemm_bench.sh g 32 2>&1 | grep -v "round-1\|^clock\|noLN" | cut -c1-150 | grep "waves\|chain\|empty"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02c_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()})
except Exception as e: print(f, "ERR", e)
PY
}
run base AUR_X=0
run s2 AUR_DECODE_STREAMS=2
run s2g AUR_DECODE_STREAMS=2 AUR_DECODE_GRAPH=1
run g1 AUR_DECODE_GRAPH=1
