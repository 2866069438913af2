#!/bin/bash
# GPU call 3 of round 4: gemm_bench at M = 32, 48, 64 (shape policy between the small-M and the full-batch regime; K split at larger M)
exec < /dev/null
mkdir -p gpurun_out
for M in 32 48 64; do bash tools/gemm_bench.sh r04c_m$M $M | grep -v "waves= 8" | grep -i "prec=1\|chain\|rc=" | cut -c1-150; done
