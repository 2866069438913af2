#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02d
bash tools/gemm_bench.sh h 64 2>&1 | grep "prj2\|chain" | grep -v round-1 | cut -c1-130
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > ${T}_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 ${T}_kernels.log
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_baseline_size.py -m gpu -q --tb=short -p no:cacheprovider -x > ${T}_gpt.log 2>&1; echo "gpt rc=$?"; tail -6 ${T}_gpt.log
/usr/bin/time -v timeout 600 python bench.py > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$?"; grep "Elapsed\|cpu_baseline" ${T}_bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r02d_bench.json"))
print(round(j["ms_per_step"],1), {k:round(v,2) for k,v in j["breakdown_ms_per_step"].items()})
r=j["roofline"]; print("dominant:", r["kernel"][:60], round(r["achieved"]), round(r["frac"],3), r["avg_launch_ms"])
for g in j["decode_gemm_kernels"]: print(g["kernel"][:50], round(g["achieved"]), round(g["frac"],3), round(g["avg_launch_ms"]*1e3,2), "us", round(g["mfma"]["frac"],3))
print("conv", round(j["roofline_second_kernel"]["frac"],3), j["roofline_second_kernel"].get("survey_8d_fp32_bytes"))
print("step", j["decode_step_roofline"])
c=j["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["rtf"], c["thread_sweep_s_per_token"]); print(c["sample"])
PY
PROF_TIMEOUT=240 bash tools/prof.sh r02d 2>&1 | head -16
PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' PMC_TIMEOUT=200 bash tools/pmc.sh dec
ls -la gpurun_out/pmc_dec/*/
