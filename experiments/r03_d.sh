#!/bin/bash
# r03 experiment D: gemm_bench after LN folding + kernel unit tests
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
timeout 300 /tmp/gemm_bench 64 > gpurun_out/r03_d_gemm_bench.log 2>&1
echo "rc=$?"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_rows or layernorm" > gpurun_out/r03_d_pytest.log 2>&1
tail -5 gpurun_out/r03_d_pytest.log
