#!/bin/bash
# round 6: the last, quarter-full round of the prompt-row FC / mlp-projection GEMMs on smaller tiles in a second launch on the SAME stream
# (AUR_PREFILL_TAIL=1, default) against one launch (=0): same box, interleaved.  -> gpurun_out/prefill_tail_ab_<tag>.log
# (the switch it drives is not in the tree: `git apply tools/experiments/prefill_tail_same_stream.patch` first -- tools/experiments/README.md)
exec < /dev/null
TAG=${1:-a}
mkdir -p gpurun_out
for R in 1 2 3; do for T in 1 0; do
  AUR_PREFILL_TAIL=$T python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side --out /tmp/pt_$T.json > /dev/null 2>&1
  python - <<PY
import json
r = json.load(open("/tmp/pt_$T.json"))
print("AUR_PREFILL_TAIL=$T rep $R: ms_per_step %.2f  prefill %.2f ms per batch  (gpt_prefill %.2f)" % (r["ms_per_step"], r["kernels"]["prefill"]["ms_per_batch"], r["breakdown_ms_per_step"]["gpt_prefill"]))
PY
done; done 2>&1 | tee gpurun_out/prefill_tail_ab_$TAG.log
