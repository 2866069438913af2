#!/bin/bash
# one PMC pass of SQ activity counters over the conv kernels of the bench command; args: <tag> [bench args]
exec < /dev/null
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES \
   --kernel-include-regex "conv1d_mfma" --output-format csv -d $OUT -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/stdout.log 2>&1
echo "rc=$?"; ls $OUT | head
