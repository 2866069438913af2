#!/bin/bash
# r03 experiment E: full GPU suite + bench with the new decode GEMMs (bf16x3 split, LN folded, one workgroup per CU)
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r03_e_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r03_e_tests.log | tail -15
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r03_e_bench.json; tail -4 gpurun_out/r03_e_bench.err
