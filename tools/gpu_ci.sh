#!/bin/bash
# Run every GPU test file in its own process (an abort in one must not hide the others); logs -> gpurun_out/
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=0
for f in tests/test_gpu_kernels.py tests/test_gpu_vocoder.py tests/test_gpu_gpt.py tests/test_gpu_api.py tests/test_gpu_edges.py; do
  n=$(basename $f .py)
  timeout ${GPU_CI_TIMEOUT:-600} python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider -p no:faulthandler "$@" > gpurun_out/$n.log 2>&1
  echo "== $f rc=$?"
  grep -v "^Extension modules" gpurun_out/$n.log | tail -${GPU_CI_TAIL:-40}
done
