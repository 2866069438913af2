#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2bw_bench.hip -o /tmp/l2bw_bench || exit 1
timeout 120 /tmp/l2bw_bench > gpurun_out/l2bw_bench.log 2>&1
echo "rc=$?"; cat gpurun_out/l2bw_bench.log
