"""Scratch timing on the GPU box: vocoder conv throughput and GPT decode step time (prints JSON lines)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests.gpu_util import SPK_KEY, make_engine  # noqa: E402
from auralis_amd.checkpoint import make_synthetic_text_ids  # noqa: E402
from auralis_amd.config import XTTSDims  # noqa: E402


def main():
    n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dims = XTTSDims()
    t0 = time.time()
    e, *_ = make_engine(n_layer, max_seqs=B, profile=True)
    print(json.dumps({"setup_s": time.time() - t0}), flush=True)
    lat = np.random.default_rng(0).standard_normal((B, 280, 1024)).astype(np.float32)
    for it in range(2):
        e.reset_stats()
        t0 = time.time()
        e.vocode(lat, None, SPK_KEY)
        dt = time.time() - t0
        s = e.stats()
        print(json.dumps({"vocode_wall_s": dt, "B": B, "conv_ms": s["conv_ms"], "vocoder_ms": s["vocoder_ms"],
                          "conv_tflops": s["conv_flops"] / (s["conv_ms"] * 1e-3) / 1e12,
                          "conv_GBps": s["conv_bytes"] / (s["conv_ms"] * 1e-3) / 1e9,
                          "launches": s["conv_launches"]}), flush=True)
    ids = make_synthetic_text_ids(dims, n_text=70)
    for it in range(2):
        e.reset_stats()
        for b in range(B):
            e.submit(ids, SPK_KEY, temperature=0.75, max_tokens=ntok, seed=b, ignore_stop=True)
        t0 = time.time()
        out = e.run_until_done()
        dt = time.time() - t0
        s = e.stats()
        audio_s = sum(len(o["wav"]) for o in out) / 24000.0
        print(json.dumps({"e2e_wall_s": dt, "B": B, "tokens": ntok, "gpt_ms": s["gpt_ms"], "steps": s["steps"],
                          "ms_per_step": s["gpt_ms"] / max(1, s["steps"]), "vocoder_ms": s["vocoder_ms"],
                          "audio_s": audio_s, "rtf": dt / audio_s}), flush=True)


if __name__ == "__main__":
    main()
