#!/bin/bash
exec < /dev/null
PMC_PASSES="fetch write" PMC_KERNELS="gemm_splitk_kernel<false" PMC_TIMEOUT=150 bash tools/pmc.sh gemm
find gpurun_out/pmc_gemm -name "*.db" -delete
ls -la gpurun_out/pmc_gemm/*; tail -2 gpurun_out/pmc_gemm/fetch/stdout.log | cut -c1-300
