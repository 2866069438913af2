#!/bin/bash
exec < /dev/null
TAG=${1:-sf}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -12
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_$name.json')); print(round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" 2>&1 | tail -1)"; }
EXTRA="" run fused X=1
EXTRA="" run fused_wg AUR_GELU_FENCE=wg
EXTRA="" run unfused AUR_FUSE_GELU=0
