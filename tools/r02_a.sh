#!/bin/bash
# round-2 first GPU call: new decode GEMM chain — kernel parity, engine parity at BASELINE size, bench A/B, rocprof
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02a
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or layernorm" > ${T}_kernels.log 2>&1; echo "kernels rc=$?"; tail -15 ${T}_kernels.log
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > ${T}_gpt.log 2>&1; echo "gpt rc=$?"; tail -15 ${T}_gpt.log
timeout 600 python -m pytest tests/test_gpu_baseline_size.py -m gpu -q --tb=short -p no:cacheprovider --durations=5 > ${T}_c2c3.log 2>&1; echo "c2c3 rc=$?"; tail -25 ${T}_c2c3.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > ${T}_bench_new.json 2> ${T}_bench_new.err; echo "bench new rc=$?"; cut -c1-300 ${T}_bench_new.json; python - <<'PY'
import json
for f in ("gpurun_out/r02a_bench_new.json",):
    try:
        j=json.load(open(f)); print(f, j["ms_per_step"], j["breakdown_ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
AUR_DECODE_GEMM=splitk timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > ${T}_bench_old.json 2> ${T}_bench_old.err; echo "bench old rc=$?"; python - <<'PY'
import json
for f in ("gpurun_out/r02a_bench_old.json",):
    try:
        j=json.load(open(f)); print(f, j["ms_per_step"], j["breakdown_ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
PROF_TIMEOUT=240 bash tools/prof.sh r02a 2>&1 | head -30
