#!/bin/bash
# round-4 evidence run: driver sequence (tests, smoke, bench, torchrun bench, rocprofv3 stats) + PMC fetch / write passes of the decode kernels
exec < /dev/null
bash tools/gpu_round_check.sh r04final
PMC_TIMEOUT=200 PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' bash tools/pmc.sh r04dec
ls gpurun_out/pmc_r04dec/*/ | head; find gpurun_out/pmc_r04dec -name "*.csv" -size +30M -delete
