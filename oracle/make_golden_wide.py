"""Wider BASELINE-size goldens: tests/golden/c2w_L30_T280.npz, c3w_L30_T280.npz, ragged_L30.npz.

TEST INFRASTRUCTURE ONLY.  Usage (build container, ~40 min on 8 cores):  python -m oracle.make_golden_wide [c2w] [c3w] [ragged]

c2w   : the C2 configuration (BASELINE.json configs[1]: 30 layers, 70 text ids, greedy, rep-pen 5.0, 280 tokens) on two more
        prompts (text seeds 12 and 13): ids, per-step top-2 margins, literal-second-pass latents (XTTSv2.py:617-687), and the
        waveform of prompt 12 (hifigan_decoder.py:776-802).
c3w   : the C3 configuration (configs[2]: T 0.75 / top_p 0.85 / top_k 50 / rep-pen 5.0) for ALL 64 seeds of the bench prompt
        (text seed 11), 280 ids each, with the race ratio of every step.
ragged: a 30-layer ragged batch for the over-subscribed 64-slot engine: 80 sequences, 6..120 text ids, max_tokens 20..150,
        natural stop (stop id 1025 ends a sequence and stays in its ids, XTTSv2.py:737) made reachable on the synthetic
        checkpoint by mel_head.bias[1025] = STOP_BIAS (the test applies the same edit to the packed weights), every third
        sequence greedy, the others sampled with their own seed.

The weights are the seeded synthetic checkpoint (seed 1234), so only ids and reference OUTPUTS are stored.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_text_ids,  # noqa: E402
                                    make_synthetic_xtts)
from auralis_amd.config import XTTSDims  # noqa: E402
from oracle import xtts_oracle as O  # noqa: E402
from oracle.make_golden_c2 import sampled_with_margins  # noqa: E402

N_LAYER = 30
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
STOP_BIAS = 1.35
SAMPLING = dict(temperature=0.75, top_k=50, top_p=0.85, repetition_penalty=5.0)


def ragged_specs():
    """(n_text, max_tokens, greedy, seed, text_seed) of the 80 sequences; deterministic"""
    rng = np.random.RandomState(2026)
    out = []
    for i in range(80):
        out.append((int(rng.randint(6, 121)), int(rng.randint(20, 151)), i % 3 == 0, 1000 + i, 200 + i))
    return out


def main():
    what = set(sys.argv[1:]) or {"c2w", "c3w", "ragged"}
    torch.set_num_threads(os.cpu_count() or 8)
    dims = XTTSDims()
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=N_LAYER)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    cond, spk = make_synthetic_conditioning(dims)
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    os.makedirs(OUT_DIR, exist_ok=True)

    if "c2w" in what:
        rec = {}
        for ts in (12, 13):
            ids = list(make_synthetic_text_ids(dims, n_text=70, seed=ts))
            c = gpt.build_cond(cond, ids)
            t0 = time.time()
            ref = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=280, ignore_stop=True), return_debug=True)
            lat = gpt.second_pass_latents(c, ref["tokens"])
            m = np.asarray(ref["margins"], np.float32)
            print(f"c2w text seed {ts}: {time.time() - t0:.1f}s; min margin {m.min():.3e} at step {int(m.argmin())}; "
                  f"distinct ids {len(set(ref['tokens']))}", flush=True)
            rec[f"text_ids_{ts}"] = np.asarray(ids, np.int32)
            rec[f"tokens_{ts}"] = np.asarray(ref["tokens"], np.int32)
            rec[f"margins_{ts}"] = m
            rec[f"latents_{ts}"] = lat[0].numpy().astype(np.float32)
            if ts == 12:
                rec["wav_12"] = O.hifi_decoder_forward(O.vocoder_effective_weights(xtts_sd), lat, spk).reshape(-1).numpy().astype(np.float32)
        np.savez_compressed(os.path.join(OUT_DIR, "c2w_L30_T280.npz"), **rec)

    if "c3w" in what:
        ids = list(make_synthetic_text_ids(dims, n_text=70, seed=11))
        c = gpt.build_cond(cond, ids)
        toks, rat = [], []
        for s in range(64):
            t0 = time.time()
            tk, r = sampled_with_margins(gpt, c, O.SamplingCfg(max_tokens=280, ignore_stop=True, seed=s, **SAMPLING))
            toks.append(tk)
            rat.append(r)
            print(f"c3w seed {s}: {time.time() - t0:.1f}s; closest race ratio {max(r):.6f}", flush=True)
        np.savez_compressed(os.path.join(OUT_DIR, "c3w_L30_T280.npz"), text_ids=np.asarray(ids, np.int32),
                            seeds=np.arange(64, dtype=np.int32), tokens=np.asarray(toks, np.int32),
                            race_ratio=np.asarray(rat, np.float32))

    if "ragged" in what:
        sd = {k: v.clone() for k, v in gpt_sd.items()}
        sd["mel_head.bias"][1025] = STOP_BIAS
        g2 = O.GPTOracle(sd, xtts_sd)
        specs = ragged_specs()
        toks, lens, stopped = np.full((len(specs), 150), -1, np.int32), [], []
        t0 = time.time()
        for i, (n_text, mt, greedy, seed, tseed) in enumerate(specs):
            ids = list(make_synthetic_text_ids(dims, n_text=n_text, seed=tseed))
            c = g2.build_cond(cond, ids)
            cfg = (O.SamplingCfg(temperature=0.0, max_tokens=mt) if greedy else O.SamplingCfg(max_tokens=mt, seed=seed, **SAMPLING))
            tk = g2.generate(c, cfg)["tokens"]
            toks[i, :len(tk)] = tk
            lens.append(len(tk))
            stopped.append(tk[-1] == 1025)
            print(f"ragged {i}: n_text {n_text} max {mt} {'greedy' if greedy else 'sampled'} -> {len(tk)} ids, stop {tk[-1] == 1025} "
                  f"({time.time() - t0:.0f}s)", flush=True)
        np.savez_compressed(os.path.join(OUT_DIR, "ragged_L30.npz"), specs=np.asarray([(a, b, int(g), s, t) for a, b, g, s, t in specs], np.int32),
                            tokens=toks, lengths=np.asarray(lens, np.int32), stopped=np.asarray(stopped, np.bool_),
                            stop_bias=np.float32(STOP_BIAS))
        print(f"ragged: {sum(stopped)} of {len(specs)} sequences ended on the stop id; lengths {min(lens)}..{max(lens)}")


if __name__ == "__main__":
    main()
