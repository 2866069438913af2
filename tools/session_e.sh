#!/bin/bash
exec < /dev/null
TAG=${1:-se}
mkdir -p gpurun_out
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_$name.json')); print(round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" 2>&1 | tail -1)"; }
EXTRA="" run graph AUR_DECODE_GRAPH=1
PMC_TIMEOUT=120 bash tools/pmc.sh $TAG
find gpurun_out/pmc_$TAG -name "*.db" -delete
ls -la gpurun_out/pmc_$TAG/*
