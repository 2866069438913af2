#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/gemm_bench.hip -o /tmp/gemm_bench || exit 1
for M in 1 64; do timeout 120 /tmp/gemm_bench $M 1 | grep -i "attention alone\|r04 prec=1\|attention, proj, fc, proj2) M=64 shapes=r03 prec=1"; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py tests/test_gpu_edges.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "^Extension" | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side --out gpurun_out/r04l.json 2>/dev/null | cut -c1-100; python -c "
import json; d=json.load(open('gpurun_out/r04l.json')); print(d['ms_per_step'], d['breakdown_ms_per_step'], d['kernels']['attention']['avg_launch_us'])"
