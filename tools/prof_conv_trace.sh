#!/bin/bash
# per-launch timeline of the vocoder kernels of one bench step (rocprofv3 kernel trace, filtered to the vocoder stream's kernels)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_convtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- \
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-mode > $OUT/stdout.log 2>&1
echo "rocprofv3 rc=$?"
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
head -1 "$f" > $R/gpurun_out/conv_trace.csv
grep -E "conv1d|conv_post|interp2|gemv_rows" "$f" >> $R/gpurun_out/conv_trace.csv
wc -l $R/gpurun_out/conv_trace.csv
rm -rf $OUT
