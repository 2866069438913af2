#!/bin/bash
# GPU call 1 of round 4: full -m gpu suite, default bench (compact line + c2 + c5s + cpu baseline), rocprofv3 cross-check, small-M gemm_bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04b_tests.log 2>&1; echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04b_tests.log | tail -8
timeout 600 python bench.py --steps 5 --warmup 2 --out gpurun_out/r04b_bench_full.json > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo "bench rc=$?"; cat gpurun_out/r04b_bench.json; grep "^\[bench [0-9]" gpurun_out/r04b_bench.err | tail -12
bash tools/gemm_bench.sh r04b_m1 1 | grep -i "chain\|prj2\|rc="
bash tools/gemm_bench.sh r04b_m16 16 | grep -i "chain\|prj2\|rc="
