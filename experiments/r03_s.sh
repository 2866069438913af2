#!/bin/bash
# r03 experiment S: side-stream weight warmer next to the 30-layer decode chain (tools/gemm_bench)
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
timeout 400 /tmp/gemm_bench 64 > gpurun_out/r03_s_gemm_bench.log 2>&1
echo "rc=$?"; grep "^chain" gpurun_out/r03_s_gemm_bench.log
