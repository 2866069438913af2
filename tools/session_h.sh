#!/bin/bash
exec < /dev/null
TAG=${1:-sh}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -8
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_$name.json')); print(round(d['ms_per_step'],1), d['breakdown_ms_per_step'], d['roofline']['launches_sampled'])" 2>&1 | tail -1)"; }
EXTRA="" run graph X=1
EXTRA="" run nograph AUR_DECODE_GRAPH=0
