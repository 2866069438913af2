#!/bin/bash
# PMC passes (one counter group per run, as the microarch guide prescribes) restricted to the vocoder conv kernels.
# usage: tools/pmc.sh <tag> [bench args]; outputs gpurun_out/pmc_<tag>/{fetch,write,util}/*counter_collection.csv
exec < /dev/null
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "util SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  case " ${PMC_PASSES:-fetch write util} " in *" $name "*) ;; *) continue;; esac
  OUT=$R/gpurun_out/pmc_$TAG/$name
  mkdir -p $OUT
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --kernel-include-regex "${PMC_KERNELS:-conv1d_mfma}" --output-format csv -d $OUT -o pmc -- \
      python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side --no-profile-pass --out /tmp/pmc_bench_full.json $PMC_BENCH_ARGS > $OUT/stdout.log 2>&1
  echo "pass $name rc=$?"
done
