#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02j
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02j_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()}); a=j["decode_attention"]; print("attn", round(a["avg_launch_ms"]*1e3,2),"us", round(a["achieved"]), round(a["frac"],3), [round(g["avg_launch_ms"]*1e3,1) for g in j["decode_gemm_kernels"]])
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_X=0" run base
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > ${T}_gpt.log 2>&1; echo "gpt rc=$?"; tail -4 ${T}_gpt.log
