#!/bin/bash
# r03 experiment F: full GPU suite after the hygiene refactor + wide goldens + erf GELU + vocoder-launch fault injection
exec < /dev/null
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r03_f_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r03_f_tests.log | tail -25
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r03_f_bench.json 2> gpurun_out/r03_f_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r03_f_bench.json; tail -4 gpurun_out/r03_f_bench.err
