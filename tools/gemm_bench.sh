#!/bin/bash
# build + run the decode-GEMM micro-benchmark on the GPU box; output -> gpurun_out/gemm_bench_<tag>.log
exec < /dev/null
TAG=${1:-a}; shift
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
for M in ${@:-64}; do timeout 120 /tmp/gemm_bench $M; done > gpurun_out/gemm_bench_$TAG.log 2>&1
echo "gemm_bench rc=$?"; cat gpurun_out/gemm_bench_$TAG.log
