#!/bin/bash
# r03 experiment N: intermediate round check (tests, smoke, bench, rocprof stats) + conv PMC traffic + --pipeline / --gemm f32 bench lines
exec < /dev/null
bash tools/gpu_round_check.sh r03n 2>&1 | cut -c1-300
PMC_PASSES='fetch write' PMC_KERNELS='conv1d_' bash tools/pmc.sh r03conv
for extra in "--pipeline" "--gemm f32"; do
  tag=$(echo $extra | tr -d ' -')
  timeout 400 python bench.py --no-cpu-baseline --no-throughput-mode $extra > gpurun_out/r03n_bench_$tag.json 2> gpurun_out/r03n_bench_$tag.err
  python - "$tag" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03n_bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
PY
done
