#!/bin/bash
# GPU call 4 of round 4: full -m gpu suite on the MFMA prompt attention + new shape policy, default bench, rocprofv3 of the bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04d_tests.log 2>&1; echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04d_tests.log | tail -8
timeout 600 python bench.py --steps 5 --warmup 2 --out gpurun_out/r04d_bench_full.json > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.err; echo "bench rc=$?"; cat gpurun_out/r04d_bench.json; grep "^\[bench [0-9]" gpurun_out/r04d_bench.err | tail -12
PROF_TIMEOUT=300 bash tools/prof.sh r04d 2>&1 | head -24
