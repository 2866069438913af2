#!/bin/bash
# r03 experiment V: round check of the final build + conv PMC traffic of the final layouts
exec < /dev/null
bash tools/gpu_round_check.sh r03final2 2>&1 | cut -c1-260
PMC_PASSES='fetch write' PMC_KERNELS='conv1d_' bash tools/pmc.sh r03conv2
