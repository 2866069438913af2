#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02n
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_tile" > ${T}_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 ${T}_kernels.log
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02n_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); print(f, round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()})
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_X=0" run tile64
ENVV="AUR_GEMM_TILE_SMALL_N=4096" run all64
