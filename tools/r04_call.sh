#!/bin/bash
# GPU call: DMA-staged pre-split weights in the prompt-row GEMMs: kernel tests (bitwise A/B), gpt + baseline-size parity, bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r04i_tests.log 2>&1; echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04i_tests.log | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --out gpurun_out/r04i_bench_full.json > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/r04i_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r04i_bench_full.json')); print(d['breakdown_ms_per_step'], d['kernels']['prefill'])"
AUR_GEMM_BDMA=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-profile-pass --out gpurun_out/r04i_bench_full_reg.json 2>/dev/null | cut -c1-200; python -c "
import json; d=json.load(open('gpurun_out/r04i_bench_full_reg.json')); print('register-staged:', d['breakdown_ms_per_step'])"
