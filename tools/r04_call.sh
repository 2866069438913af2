#!/bin/bash
# GPU call 5 of round 4: sampler (per-wave top-k search, shuffle reductions): kernel + baseline-size parity tests, then the bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04e_tests.log 2>&1; echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04e_tests.log | tail -8
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --out gpurun_out/r04e_bench_full.json > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err; echo "bench rc=$?"; cat gpurun_out/r04e_bench.json; grep "^\[bench [0-9]" gpurun_out/r04e_bench.err | tail -12
