#!/bin/bash
# r03 experiment A: decode GEMM shapes / bf16x3 split / cross-launch prefetch (tools/gemm_bench), M = 64
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
( timeout 300 /tmp/gemm_bench 64; AUR_GEMM_SHAPES=r02 timeout 120 /tmp/gemm_bench 64 1 ) > gpurun_out/r03_a_gemm_bench.log 2>&1
echo "rc=$?"; cat gpurun_out/r03_a_gemm_bench.log
