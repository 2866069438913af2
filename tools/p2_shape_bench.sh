#!/bin/bash
# round 6 experiment: the K = 4096 projection at 33-64 rows on a K split with wider row tiles (-DAUR_P2_SHAPE=1|2|3) against the product
# shape, same box, interleaved.  usage: bash tools/p2_shape_bench.sh <tag> [M ...]   -> gpurun_out/p2_shape_<tag>.log
# (the switch it drives is not in the tree: `git apply tools/experiments/gemm_rows_ksplit_wide.patch` first -- tools/experiments/README.md)
exec < /dev/null
TAG=${1:-a}; shift
mkdir -p gpurun_out
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16"
BINS=""
for V in ${P2_SHAPES:-0 1 2 3}; do
  F=""; [ "$V" != 0 ] && F="-DAUR_P2_SHAPE=$V"
  $CC $F tools/gemm_bench.hip -o /tmp/gemm_bench_p2_$V || exit 1
  BINS="$BINS /tmp/gemm_bench_p2_$V"
done
for R in $(seq 1 ${GEMM_BENCH_REPS:-2}); do for M in ${@:-64}; do for B in $BINS; do echo "=== $B M=$M rep $R"; timeout 180 $B $M $GEMM_BENCH_QUICK; done; done; done > gpurun_out/p2_shape_$TAG.log 2>&1
grep -E "^===|prj2|chain" gpurun_out/p2_shape_$TAG.log
