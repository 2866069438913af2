#!/bin/bash
# r03 experiment B: what bounds a decode GEMM launch (tools/gemm_diag)
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_diag.hip -o /tmp/gemm_diag || exit 1
timeout 300 /tmp/gemm_diag 64 ${DIAG_SKIP:-0} > gpurun_out/r03_b_gemm_diag.log 2>&1
echo "rc=$?"; cat gpurun_out/r03_b_gemm_diag.log
