#!/bin/bash
# c5s (BASELINE configs[4] at one GPU's share) by slot count / admission group; one record per line -> gpurun_out/c5s_sweep_<tag>.jsonl
# usage: bash tools/c5s_sweep.sh <tag> "<slots>[:admit_min_batch[:vocoder_min_batch]] ..."
exec < /dev/null
TAG=${1:-a}; shift
mkdir -p gpurun_out
OUT=gpurun_out/c5s_sweep_$TAG.jsonl
: > $OUT
for cfg in ${@:-64 96 128}; do
  IFS=: read S A V <<< "$cfg"
  timeout 300 python bench.py --workload c5s --warmup 1 --c5s-slots $S --admit-min-batch ${A:-0} --vocoder-min-batch ${V:-0} 2> gpurun_out/c5s_sweep_${TAG}_$S.err | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('slots','samples_per_s','rtf','slot_occupancy','decode_steps','decode_ms_per_step','prefill_ms','vocoder_ms','vocoder_batches','wall_s','first_chunk_s','in_order','chunks')} | {'cfg':'$cfg'}))" >> $OUT
  echo "rc=$? ($cfg)"; tail -1 $OUT | cut -c1-400
done
