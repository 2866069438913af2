#!/bin/bash
# PMC passes behind DESIGN section 3 item 6 (what a decode GEMM launch waits for): L1 -> L2 read latency and count, L1 stalls on
# its pending-request limit, L2 hit / miss, outstanding vector-memory instructions.  One counter group per run (rocprofv3 --pmc
# alone), restricted to the decode kernels (PMC_KERNELS=<regex> for others, e.g. the vocoder convs; PMC_BENCH_ARGS for the bench command;
# a fourth pass with the LDS and matrix-pipe counters when PMC_LDS=1).
# usage: tools/pmc_latency.sh <tag>  -> gpurun_out/pmc_<tag>/<pass>/pmc_counter_collection.csv
exec < /dev/null
TAG=${1:-r04lat}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
            ${PMC_LDS:+"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"}; do
  i=$((i+1)); OUT=$R/gpurun_out/pmc_$TAG/pass$i; mkdir -p $OUT
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "${PMC_KERNELS:-paged_attention_kernel|gemm_rows_kernel}" --output-format csv -d $OUT -o pmc -- \
      python $R/bench.py --steps 1 --warmup 0 ${PMC_BENCH_ARGS:---tokens 40} --no-cpu-baseline --no-side --no-profile-pass --out /tmp/pmc_bench_full.json > $OUT/stdout.log 2>&1
  echo "pass $i ($pass) rc=$?"; tail -2 $OUT/stdout.log | cut -c1-200
done
python - <<PY
import csv, collections, re, glob
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/pass*/pmc_counter_collection.csv")):
    g = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void aur::", ""), r["Counter_Name"])
        g[k][0] += 1
        g[k][1] += float(r["Counter_Value"])
    print(f.split("/")[-2])
    for (k, c), (n, t) in sorted(g.items()):
        print(f"  {k:62s} {c:34s} dispatches {n:5d} mean {t / n:16.1f}")
PY
