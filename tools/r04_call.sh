#!/bin/bash
# A/B: the sampler writes the step's tokens into the pinned block itself vs a D2H copy per step
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_edges.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "^Extension" | tail -3
for i in 1 2; do
for z in 1 0; do
AUR_ZERO_COPY_TOKENS=$z timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass --c5-chars 6000 --out gpurun_out/r04k_$z.json 2>/dev/null | cut -c1-60; python -c "
import json; d=json.load(open('gpurun_out/r04k_$z.json')); print('zero_copy=$z', d['ms_per_step'], d['breakdown_ms_per_step']['gpt_ms_per_decode_step'], 'c2 step', d['c2']['decode_step']['ms'], 'tta', d['c2']['time_to_audio_ms'])"
done; done
