#!/bin/bash
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcsq_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
   --kernel-include-regex "conv1d_mfma" --output-format csv -d $OUT -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-mode --tokens 120 > $OUT/stdout.log 2>&1
echo "rc=$?"; ls $OUT | head
python3 - <<'PY'
import csv, collections, os, glob
f=glob.glob(os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/pmcsq_r02/*counter_collection.csv")) or glob.glob("/root/repo/gpurun_out/pmcsq_r02/*counter_collection.csv")
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void aur::","")+" g"+r["Grid_Size"]
    d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["SQ_BUSY_CYCLES"])[:14]:
    wc=v["SQ_WAVE_CYCLES"] or 1
    print(f"{k[:52]:52s} n={n[k]:2d} wait_any {v['SQ_WAIT_ANY']/wc:5.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:5.2f} lds_act {v['SQ_ACTIVE_INST_LDS']/wc:5.2f} vmem_act {v['SQ_ACTIVE_INST_VMEM']/wc:5.2f} mfma_busy/busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['SQ_BUSY_CYCLES'] or 1):6.3f} bankconf/wavecyc {v['SQ_LDS_BANK_CONFLICT']/wc:6.3f}")
PY
