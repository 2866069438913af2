#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpt.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r04j_tests.log 2>&1; echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04j_tests.log | tail -4
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-profile-pass --out gpurun_out/r04j_dma.json 2>/dev/null | cut -c1-120; python -c "
import json; d=json.load(open('gpurun_out/r04j_dma.json')); print('pre-split planes:', d['breakdown_ms_per_step'])"
AUR_GEMM_PRESPLIT=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-profile-pass --out gpurun_out/r04j_reg.json 2>/dev/null | cut -c1-120; python -c "
import json; d=json.load(open('gpurun_out/r04j_reg.json')); print('split on the fly:', d['breakdown_ms_per_step'])"
done
