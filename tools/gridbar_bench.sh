#!/bin/bash
# build + run the grid-barrier micro-benchmark on the GPU box; output -> gpurun_out/gridbar_bench.log
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridbar_bench.hip -o /tmp/gridbar_bench || exit 1
timeout 120 /tmp/gridbar_bench > gpurun_out/gridbar_bench.log 2>&1
echo "gridbar_bench rc=$?"; cat gpurun_out/gridbar_bench.log
