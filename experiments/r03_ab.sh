#!/bin/bash
# r03 experiment AB: round check of the final build + conv PMC traffic of the final kernels
exec < /dev/null
bash tools/gpu_round_check.sh r03final4 2>&1 | cut -c1-260
PMC_PASSES='fetch write' PMC_KERNELS='conv1d_|resblock_round' bash tools/pmc.sh r03conv3
