#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summary CSVs -> gpurun_out/prof_<tag>/
# usage: tools/prof.sh <tag> [bench args...]
exec < /dev/null
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- \
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side --out $OUT/bench_full.json "$@" > $OUT/stdout.log 2>&1
echo "rocprofv3 rc=$?"
grep -h '^{"metric"' $OUT/stdout.log | cut -c1-400
find $OUT -type f | head -20
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -40 "$f"; fi
# the raw trace is large: keep only the stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
