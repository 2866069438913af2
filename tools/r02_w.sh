#!/bin/bash
# PMC refresh of the decode kernels' HBM traffic (fetch / write passes, 40-token runs) with the final kernels
exec < /dev/null
PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' PMC_TIMEOUT=200 bash tools/pmc.sh dec2
cd ${GRAFT_REPO_ROOT:-.}
python tools/pmc_decode_summary.py gpurun_out/pmc_dec2 40 | tail -12
mkdir -p gpurun_out/pmc_dec2_keep; for p in fetch write; do f=$(find gpurun_out/pmc_dec2/$p -name "*counter_collection.csv" | head -1); python - "$f" gpurun_out/pmc_dec2_keep/$p.csv <<'PY'
import csv,sys
# keep one row per (kernel, dispatch) with the columns the summary reads: small enough to commit
rows=list(csv.DictReader(open(sys.argv[1])))
w=csv.writer(open(sys.argv[2],"w")); w.writerow(["Kernel_Name","Counter_Name","Counter_Value"])
for r in rows: w.writerow([r["Kernel_Name"][:80],r["Counter_Name"],r["Counter_Value"]])
PY
done
ls -la gpurun_out/pmc_dec2_keep; cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_new.json
find gpurun_out/pmc_dec2 -name "*.db" -delete
