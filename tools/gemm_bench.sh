#!/bin/bash
# build + run the decode-kernel micro-benchmark on the GPU box; output -> gpurun_out/gemm_bench_<tag>.log
# usage: bash tools/gemm_bench.sh <tag> [M ...]
#   GEMM_BENCH_AB=1 also builds (a) the same sources WITHOUT -amdgpu-kernarg-preload-count and (b) the round-4 kernels kept under
#   tools/_r04/ (git-ignored copy of HEAD~'s sources, when present) and runs all three for every M
exec < /dev/null
TAG=${1:-a}; shift
mkdir -p gpurun_out
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result"
$CC -mllvm -amdgpu-kernarg-preload-count=16 tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
BINS="/tmp/gemm_bench"
if [ -n "$GEMM_BENCH_AB" ]; then
  $CC tools/gemm_bench.hip -o /tmp/gemm_bench_nopreload && BINS="$BINS /tmp/gemm_bench_nopreload"
  if [ -f tools/_r04/gemm_bench.hip ]; then
    (cd tools/_r04 && $CC gemm_bench.hip -o /tmp/gemm_bench_r04) && BINS="$BINS /tmp/gemm_bench_r04"
    # the same round-4 kernels with tile loads and MFMAs compiled out (tools/_r04 carries an R04_NODATA guard around them)
    grep -q R04_NODATA tools/_r04/gpt_kernels.hip && (cd tools/_r04 && $CC -DR04_NODATA gemm_bench.hip -o /tmp/gemm_bench_r04_nodata) && BINS="$BINS /tmp/gemm_bench_r04_nodata"
  fi
fi
#   GEMM_BENCH_VARIANT="-DAUR_GR_LN_EARLY=0" also builds the current sources with these extra flags (/tmp/gemm_bench_variant)
#   GEMM_BENCH_PREV=1 also builds tools/_prev/ (git-ignored mirror of an earlier commit's csrc/ + tools/gemm_bench.hip: same-box A/B
#   of a change that has no compile-time switch)
#   GEMM_BENCH_QUICK=1: chains only;  GEMM_BENCH_REPS=n: the whole set of binaries n times, interleaved
if [ -n "$GEMM_BENCH_VARIANT" ]; then
  $CC -mllvm -amdgpu-kernarg-preload-count=16 $GEMM_BENCH_VARIANT tools/gemm_bench.hip -o /tmp/gemm_bench_variant && BINS="$BINS /tmp/gemm_bench_variant"
fi
if [ -n "$GEMM_BENCH_PREV" ] && [ -f tools/_prev/tools/gemm_bench.hip ]; then
  (cd tools/_prev && $CC -mllvm -amdgpu-kernarg-preload-count=16 tools/gemm_bench.hip -o /tmp/gemm_bench_prev) && BINS="$BINS /tmp/gemm_bench_prev"
fi
for R in $(seq 1 ${GEMM_BENCH_REPS:-1}); do for M in ${@:-64}; do for B in $BINS; do echo "=== $B M=$M rep $R"; timeout 180 $B $M $GEMM_BENCH_QUICK; done; done; done > gpurun_out/gemm_bench_$TAG.log 2>&1
echo "gemm_bench rc=$?"; cat gpurun_out/gemm_bench_$TAG.log
