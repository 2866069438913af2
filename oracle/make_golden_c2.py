"""Generate the BASELINE-size goldens: tests/golden/c2_L30_T280.npz and tests/golden/c3_L30_T280.npz.

TEST INFRASTRUCTURE ONLY.  Usage (build container, ~3 min on 8 cores):  python -m oracle.make_golden_c2 [--f64-check]

C2 (BASELINE.json configs[1]; SURVEY §8 working sizes): 30 layers, 70 text ids (seed 11, the bench's prompt), greedy with
repetition penalty 5.0, 280 mel tokens in fixed-length mode -> the reference path end to end as the oracle restates it
(XTTSv2.py:762-814): AR tokens, LITERAL second pass (XTTSv2.py:617-687), HiFi-GAN (hifigan_decoder.py:776-802).
Stored: tokens, per-step top-2 margin of the penalised logits (to explain any greedy flip), second-pass latents, waveform.

C3 (configs[2]): the same prompt sampled with T 0.75 / top_p 0.85 / top_k 50 / rep-pen 5.0, seeds 0..7, 280 tokens each,
under the counter-hash Exp(1) noise the HIP sampler shares (oracle.exp_noise).  Stored: tokens per seed and, per step,
the ratio of the two largest race values probs/e (a ratio near 1 marks a step where rounding noise may pick the other id).

The weights are the seeded synthetic checkpoint (seed 1234), so only ids and reference OUTPUTS are stored.
--f64-check re-runs the greedy loop in float64 and reports whether fp32 rounding alone changes any id (a fragility probe
for the parity contract; informational).
"""
import argparse
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

N_LAYER, N_TEXT, TEXT_SEED = 30, 70, 11
C2_TOKENS = 280
C3_TOKENS, C3_SEEDS = 280, (0, 1, 2, 3, 4, 5, 6, 7)
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def sampled_with_margins(gpt, cond, cfg):
    """GPTOracle.generate for a sampled config, additionally recording the race margin of every step."""
    n0 = [None]
    ratios = []
    orig = O.sample_token

    def spy(logits, temperature, top_k, top_p, noise):
        tok = orig(logits, temperature, top_k, top_p, noise)
        z = logits.to(torch.float32) / temperature
        V = z.shape[0]
        zs, zi = z.sort()
        zs = zs.masked_fill(zs < zs[V - top_k], float("-inf"))
        ps = zs.softmax(-1)
        m = ps.cumsum(-1) <= (1.0 - top_p)
        m[-1] = False
        zs = zs.masked_fill(m, float("-inf"))
        zz = torch.empty_like(zs).scatter_(0, zi, zs)
        q = torch.softmax(zz, -1) / torch.from_numpy(noise)
        t2 = torch.topk(q, 2).values
        ratios.append(float(t2[1] / t2[0]))
        return tok

    O.sample_token = spy
    try:
        out = gpt.generate(cond, cfg)
    finally:
        O.sample_token = orig
    return out["tokens"], ratios


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f64-check", action="store_true")
    ap.add_argument("--skip-c3", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    dims = XTTSDims()
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=N_LAYER)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    cond, spk = make_synthetic_conditioning(dims)
    ids = list(make_synthetic_text_ids(dims, n_text=N_TEXT, seed=TEXT_SEED))
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    c = gpt.build_cond(cond, ids)
    os.makedirs(OUT_DIR, exist_ok=True)

    t0 = time.time()
    ref = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=C2_TOKENS, ignore_stop=True), return_debug=True)
    t1 = time.time()
    lat = gpt.second_pass_latents(c, ref["tokens"])
    t2 = time.time()
    wav = O.hifi_decoder_forward(O.vocoder_effective_weights(xtts_sd), lat, spk).reshape(-1).numpy()
    t3 = time.time()
    stash = gpt.latents_from_decode_rows(ref["decode_rows"], len(ref["tokens"]))[0].numpy()
    margins = np.asarray(ref["margins"], dtype=np.float32)
    print(f"C2: {len(ref['tokens'])} tokens in {t1 - t0:.1f}s, second pass {t2 - t1:.1f}s, vocoder {t3 - t2:.1f}s; "
          f"min margin {margins.min():.3e} at step {int(margins.argmin())}; stop id emitted: {1025 in ref['tokens']}; "
          f"wav rms {np.sqrt(np.mean(wav ** 2)):.4f}; |stash - second pass| max {np.abs(stash - lat[0].numpy()).max():.2e}")
    np.savez_compressed(os.path.join(OUT_DIR, "c2_L30_T280.npz"), text_ids=np.asarray(ids, np.int32),
                        tokens=np.asarray(ref["tokens"], np.int32), margins=margins,
                        latents=lat[0].numpy().astype(np.float32), wav=wav.astype(np.float32),
                        timing_s=np.asarray([t1 - t0, t2 - t1, t3 - t2], np.float32), cores=np.int32(torch.get_num_threads()))

    if args.f64_check:
        g64 = O.GPTOracle(gpt_sd, xtts_sd)
        g64.w = {k: v.double() for k, v in g64.w.items()}
        g64.x = {k: v.double() for k, v in xtts_sd.items() if k.startswith("text_")}
        r64 = g64.generate(c.double(), O.SamplingCfg(temperature=0.0, max_tokens=C2_TOKENS, ignore_stop=True))
        same = sum(int(a == b) for a, b in zip(r64["tokens"], ref["tokens"]))
        first = next((i for i, (a, b) in enumerate(zip(r64["tokens"], ref["tokens"])) if a != b), None)
        print(f"f64 check: {same}/{C2_TOKENS} ids equal to the fp32 oracle; first difference at step {first}"
              + ("" if first is None else f" (fp32 margin there {margins[first]:.3e})"))

    if not args.skip_c3:
        toks, rat = [], []
        for s in C3_SEEDS:
            t0 = time.time()
            cfg = O.SamplingCfg(temperature=0.75, top_k=50, top_p=0.85, repetition_penalty=5.0, max_tokens=C3_TOKENS,
                                ignore_stop=True, seed=s)
            tk, r = sampled_with_margins(gpt, c, cfg)
            toks.append(tk)
            rat.append(r)
            print(f"C3 seed {s}: {len(tk)} tokens in {time.time() - t0:.1f}s; closest race ratio {max(r):.6f}")
        np.savez_compressed(os.path.join(OUT_DIR, "c3_L30_T280.npz"), text_ids=np.asarray(ids, np.int32),
                            seeds=np.asarray(C3_SEEDS, np.int32), tokens=np.asarray(toks, np.int32),
                            race_ratio=np.asarray(rat, np.float32))


if __name__ == "__main__":
    main()
