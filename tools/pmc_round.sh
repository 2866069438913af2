# PMC passes of a round (here r06): decode kernels at the bench's own --tokens 280, vocoder convs; summary folded into a copy of hbm_traffic.json
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
PMC_TIMEOUT=500 PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 280' bash tools/pmc.sh r06dec
PMC_TIMEOUT=400 PMC_PASSES='fetch write' PMC_KERNELS='conv1d_|resblock_round' bash tools/pmc.sh r06conv
cd $R
PMC_ROUND=r06 python tools/pmc_round_summary.py gpurun_out/pmc_r06dec 280 gpurun_out/pmc_r06conv > gpurun_out/r06_pmc_summary.log 2>&1
tail -40 gpurun_out/r06_pmc_summary.log
cp profiles/hbm_traffic.json gpurun_out/r06_hbm_traffic.json
ls -la gpurun_out/pmc_r06dec/*/ gpurun_out/pmc_r06conv/*/ | head -30
# keep the merge small: the raw counter CSVs of the decode passes are tens of MB
find gpurun_out/pmc_r06dec gpurun_out/pmc_r06conv -name "*.csv" -size +8M -delete
find gpurun_out/pmc_r06dec gpurun_out/pmc_r06conv -name "*.db" -delete
