#!/bin/bash
# r03 experiment C: launch floor by geometry (tools/launch_bench)
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_bench.hip -o /tmp/launch_bench || exit 1
timeout 120 /tmp/launch_bench > gpurun_out/r03_c_launch_bench.log 2>&1
echo "rc=$?"; cat gpurun_out/r03_c_launch_bench.log
