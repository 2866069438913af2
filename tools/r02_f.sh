#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02f
timeout 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_api.py -m gpu -q --tb=short -p no:cacheprovider > ${T}_edges.log 2>&1; echo "edges rc=$?"; tail -25 ${T}_edges.log
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py -m gpu -q --tb=short -p no:cacheprovider -s > ${T}_gpt.log 2>&1; echo "gpt rc=$?"; grep "kv_fp16 report" ${T}_gpt.log; tail -8 ${T}_gpt.log
run() { tag=$1; shift; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02f_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()}); a=j["decode_attention"]; print("attn", round(a["avg_launch_ms"]*1e3,2),"us", round(a["achieved"]), round(a["frac"],3))
except Exception as e: print(f, "ERR", e)
PY
}
run base
run kvh --kv fp16
