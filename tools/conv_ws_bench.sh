#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/conv_ws_bench.hip -o /tmp/conv_ws_bench || exit 1
timeout 200 /tmp/conv_ws_bench > gpurun_out/conv_ws_bench.log 2>&1
echo "rc=$?"; cat gpurun_out/conv_ws_bench.log
