#!/bin/bash
# Multi-GPU self-check on one node: bench.py at N = 1, 2, 4, 8 ranks (one process per GPU over RCCL), one JSON line per N with
# multi_gpu.{rccl_ranks, conditioning_hash_equal_across_ranks, output_hash_equal_across_ranks, native_route, per_rank_ms_per_step} and, for N > 1, c4 (512 utterances, strong scaling).
# usage: bash tools/scale_check.sh [N ...]      -> gpurun_out/scale_<N>.json
exec < /dev/null
mkdir -p gpurun_out
NS=${@:-"1 2 4 8"}
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
for N in $NS; do
  if [ "$N" -gt "$HAVE" ]; then echo "skip N=$N (only $HAVE GPUs visible)"; continue; fi
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline --out gpurun_out/scale_${N}_full.json > gpurun_out/scale_$N.json 2> gpurun_out/scale_$N.err
  echo "N=$N rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/scale_$N.json"))
    print({k: d.get(k) for k in ("n_gpus", "value", "ms_per_step")}, d.get("multi_gpu"), d.get("c4"))
except Exception as e:
    print("no JSON line:", e)
PY
done
