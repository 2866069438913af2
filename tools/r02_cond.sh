#!/bin/bash
# GPU: HIP conditioning parity tests, end-to-end cloning, ABI, rocprof of one conditioning call
exec < /dev/null
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/cond_once.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from auralis_amd._lib import NativeEngine
from auralis_amd.weights import pack_conditioning
from auralis_amd.checkpoint import make_synthetic_conditioning_weights
from auralis_amd.config import XTTSDims
import torch
sd = make_synthetic_conditioning_weights(XTTSDims(), seed=99); sd["mel_stats"] = torch.ones(80)
g = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/cond_female_6s.npz"))
eng = NativeEngine(n_layer=1, max_seqs=1); eng.load_weights(pack_conditioning(sd))
pcm = g["pcm16"].astype(np.float32) / 32767.0
for _ in range(4): eng.compute_conditioning([pcm], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
eng.close()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cond -o cond -- python /tmp/cond_once.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cond.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_cond -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/cond_kernel_stats.csv
head -32 $GRAFT_REPO_ROOT/gpurun_out/cond_kernel_stats.csv | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/prof_cond -name "*.db" -delete; find $GRAFT_REPO_ROOT/gpurun_out/prof_cond -name "*kernel_trace.csv" -delete
