#!/bin/bash
# FC tile policy A/B + kernel / parity tests
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_size.py tests/test_gpu_gpt.py -m gpu -x -q 2>&1 | tail -3
T=gpurun_out/r02v
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.load(open("gpurun_out/r02v_%s.json"%sys.argv[1])); b=j["breakdown_ms_per_step"]
fc=[g for g in j["decode_gemm_kernels"] if "fc" in g["kernel"]][0]
print(sys.argv[1], round(j["ms_per_step"],1), "decode step", round(b["gpt_ms_per_decode_step"],3), "fc us", round(fc["avg_launch_ms"]*1e3,2))
PY
}
ENVV="AUR_GEMM_FC_TILE=2x1" run new
ENVV="AUR_GEMM_FC_TILE=1x2" run old
ENVV="AUR_GEMM_FC_TILE=2x1" run new2
ENVV="AUR_GEMM_FC_TILE=1x2" run old2
