#!/bin/bash
# GPU call: kernel parity of the decode GEMM + engine parity (small, BASELINE size) + bench
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02b
bash tools/gemm_bench.sh e 64 2>&1 | grep -v "round-1\|^clock\|nt=1\|noLN" | cut -c1-220
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or layernorm or sampler" > ${T}_kernels.log 2>&1; echo "kernels rc=$?"; tail -12 ${T}_kernels.log
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py -m gpu -q --tb=short -p no:cacheprovider -x > ${T}_gpt.log 2>&1; echo "gpt rc=$?"; tail -15 ${T}_gpt.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
for f in ("gpurun_out/r02b_bench.json",):
    try:
        j=json.load(open(f)); print(f, j["ms_per_step"], j["breakdown_ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 ${T}_bench.err
