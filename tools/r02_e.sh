#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02e
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > ${T}_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 ${T}_kernels.log
S=$(date +%s); timeout 900 python bench.py > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"; grep "cpu_baseline" ${T}_bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r02e_bench.json"))
print(round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()})
r=j["roofline"]; print("dominant:", r["kernel"][:60], round(r["achieved"]), round(r["frac"],3), r["avg_launch_ms"], r.get("traffic"), r.get("rocprof"))
for g in j["decode_gemm_kernels"]: print(g["kernel"][:50], round(g["achieved"]), round(g["frac"],3), round(g["avg_launch_ms"]*1e3,2), "us", round(g["mfma"]["frac"],3), g.get("rocprof",{}).get("avg_launch_us"))
print("conv", round(j["roofline_second_kernel"]["frac"],3), j["roofline_second_kernel"].get("survey_8d_fp32_bytes"))
print("step", j["decode_step_roofline"])
c=j["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["rtf"], c["thread_sweep_s_per_token"]); print(c["sample"]); print(c["c1_50char_70_tokens"])
PY
