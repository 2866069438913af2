#!/bin/bash
# r03 experiment H: profiles of the default bench command: rocprofv3 kernel stats, PMC FETCH/WRITE passes of the decode kernels
# (--tokens 40) and of the vocoder convs
exec < /dev/null
mkdir -p gpurun_out
PROF_TIMEOUT=300 bash tools/prof.sh r03 2>&1 | tail -3
PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' PMC_TIMEOUT=240 bash tools/pmc.sh r03dec
PMC_PASSES='fetch write util' PMC_KERNELS='conv1d_mfma' PMC_TIMEOUT=240 bash tools/pmc.sh r03conv
python tools/pmc_r03_summary.py gpurun_out/pmc_r03dec 40 gpurun_out/pmc_r03conv > gpurun_out/r03_pmc_summary.log 2>&1
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_r03.json
tail -40 gpurun_out/r03_pmc_summary.log
ls -la gpurun_out/prof_r03 gpurun_out/pmc_r03dec/* gpurun_out/pmc_r03conv/* | head -30
