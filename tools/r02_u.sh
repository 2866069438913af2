#!/bin/bash
# vocoder staging prefetch variants: vocoder parity + bench
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/r02u_bench.json 2> gpurun_out/r02u_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r02u_bench.json")); b=j["breakdown_ms_per_step"]; c=j["roofline_second_kernel"]
print(round(j["ms_per_step"],1), {k:round(v,2) for k,v in b.items()}, "conv frac", round(c["frac"],3), round(c["survey_8d_fp32_bytes"]["frac"],3), "mfma", round(c["mfma"]["frac"],3))
PY
