#!/bin/bash
# GPU session: full -m gpu suite, bench (pipelined decode on / off), rocprof trace.  usage: tools/session_b.sh <tag>
exec < /dev/null
TAG=${1:-sb}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/ -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -25
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
AUR_DECODE_PIPELINE=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_nopipe.json 2> gpurun_out/${TAG}_bench_nopipe.err; echo "bench(nopipe) rc=$?"; cut -c1-330 gpurun_out/${TAG}_bench_nopipe.json
PROF_TIMEOUT=240 bash tools/prof.sh $TAG 2>&1 | head -16
