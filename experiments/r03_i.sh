#!/bin/bash
# r03 experiment I: what bounds a 128-channel vocoder conv (tools/conv_diag)
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/conv_diag.hip -o /tmp/conv_diag || exit 1
timeout 300 /tmp/conv_diag > gpurun_out/r03_i_conv_diag.log 2>&1
echo "rc=$?"; cat gpurun_out/r03_i_conv_diag.log
