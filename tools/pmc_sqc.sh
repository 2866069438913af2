#!/bin/bash
# Instruction- and scalar-cache counters of the decode kernels (is a launch's start paid in cold caches?): one rocprofv3 --pmc pass.
# usage: tools/pmc_sqc.sh <tag>  -> gpurun_out/pmc_<tag>/sqc/pmc_counter_collection.csv + a per-kernel summary on stdout
exec < /dev/null
TAG=${1:-r05sqc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
            "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); OUT=$R/gpurun_out/pmc_$TAG/sqc$i; mkdir -p $OUT
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "${PMC_KERNELS:-paged_attention_kernel|gemm_rows_kernel}" --output-format csv -d $OUT -o pmc -- \
      python $R/bench.py --steps 1 --warmup 0 --tokens 40 --no-cpu-baseline --no-side --no-profile-pass --out /tmp/pmc_bench_full.json > $OUT/stdout.log 2>&1
  echo "pass $i ($pass) rc=$?"; tail -2 $OUT/stdout.log | cut -c1-200
done
python - <<PY
import csv, collections, re, glob
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/sqc*/pmc_counter_collection.csv")):
    g = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void aur::", ""), r["Counter_Name"])
        g[k][0] += 1
        g[k][1] += float(r["Counter_Value"])
    for (k, c), (n, t) in sorted(g.items()):
        print(f"  {k:62s} {c:34s} dispatches {n:5d} mean {t / n:14.1f}")
PY
