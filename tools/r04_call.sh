#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
for M in 1 16 64; do timeout 120 /tmp/gemm_bench $M 1 | grep -i "attention alone\|r04 prec=1"; done > gpurun_out/gemm_bench_r04h_attn.log 2>&1; cat gpurun_out/gemm_bench_r04h_attn.log
