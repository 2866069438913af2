#!/bin/bash
# r03 experiment G: full GPU suite (wide goldens, erf GELU, fault injection, ups epilogue) + kv_fp16 report + bench + conv trace
exec < /dev/null
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/r03_g_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r03_g_tests.log | tail -25
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r03_g_bench.json 2> gpurun_out/r03_g_bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/r03_g_bench.json; tail -2 gpurun_out/r03_g_bench.err
bash tools/prof_conv_trace.sh 2>&1 | tail -2
