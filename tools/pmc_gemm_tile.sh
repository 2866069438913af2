#!/bin/bash
# two PMC passes of SQ activity counters over the prompt-row GEMM kernels (gemm_tile_split_kernel) of one bench step; args: <tag>
exec < /dev/null
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcgt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
   --kernel-include-regex "gemm_tile_split" --output-format csv -d $OUT/a -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-profile-pass --out /tmp/pmc_full.json "$@" > $OUT/a.log 2>&1
echo "rc=$?"
timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
   --kernel-include-regex "gemm_tile_split" --output-format csv -d $OUT/b -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-profile-pass --out /tmp/pmc_full.json "$@" > $OUT/b.log 2>&1
echo "rc=$?"
python3 - $OUT <<'PY'
import csv, sys, os, collections, glob
out = sys.argv[1]
for sub in ("a", "b"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void aur::", "")[:60]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in per.items():
            print(sub, k, {n: (len(v), round(sum(v) / len(v))) for n, v in c.items()})
PY
