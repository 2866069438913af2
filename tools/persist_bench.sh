#!/bin/bash
# build + run the persistent-decode prototype benchmark on the GPU box; output -> gpurun_out/persist_bench_<tag>.log
exec < /dev/null
TAG=${1:-a}; shift
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/persist_bench.hip -o /tmp/persist_bench || exit 1
for M in ${@:-64}; do timeout 60 /tmp/persist_bench $M; echo "rc=$?"; done > gpurun_out/persist_bench_$TAG.log 2>&1
cat gpurun_out/persist_bench_$TAG.log
