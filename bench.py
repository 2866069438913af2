#!/usr/bin/env python3
"""bench.py — realtime-factor / audio-samples-per-second of the generate_speech() hot path on MI355X.

Headline workload (default, `--workload c3`) = BASELINE.json configs[2], the configuration `metric` is quoted on: 64 concurrent
200-char utterances per GPU, shared speaker latent, temperature 0.75 / top_p 0.85 / top_k 50 / repetition penalty 5.0;
one "step" = one pass of the hot path over one such batch: 70 text tokens -> prefill of 103 rows -> 280 mel tokens (fixed-length
mode) -> latent stash -> HiFi-GAN -> 312 064 samples per utterance, waveforms delivered to the host.  Weights (seeded
synthetic, true shapes) and speaker conditioning are resident in HBM before the timed region.  The timed region runs the engine
exactly as the parity tests do (profile mode OFF); the per-kernel roofline numbers come from a profile pass right after it (the
same workload once more with aur_set_profile on: HIP events on the stream the kernels run on, see include/auralis_amd.h).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Other workloads: `--workload c3f` (the headline workload as 64 concurrent TTSRequests through the TTS facade: the drop-in boundary on
the clock), `--workload c2` (configs[1]: ONE 200-char utterance, greedy, batch 1: time to audio), `--workload c5s` (configs[4]
at single-GPU scale: one GPU's eighth of the ~450 k characters, mixed en/fr/de through longform.stream_longform, natural stop, ragged),
`--workload c4` (configs[3]: 512 utterances dealt 64 at a time to the ranks by parallel.shard_units, strong scaling).  The default
run also measures c3f, c2 and c5s once after the headline (`--no-side` skips them) and, under torchrun with N > 1, c4.

Output: the LAST stdout line is one compact JSON object (< 4 KB: the contract fields + roofline + cpu_baseline + one summary per
side workload); everything else goes to gpurun_out/bench_full.json and stderr.

Multi-GPU: utterances are independent, so each rank runs its own batches; the only collective is one RCCL broadcast of the speaker
conditioning (133 120 B) from rank 0 before the timed region.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 achievable)
FP32_MFMA_PEAK_TFLOPS = 157.3    # exact-f32 MFMA / vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA
MATMUL_PARAMS_PER_LAYER = 12_582_912   # c_attn + c_proj + c_fc + mlp.c_proj
VOC_BYTES_8D_FP32, VOC_BYTES_8D_FP16 = 21301.0, 10650.0   # vocoder activation bytes per output sample
GEMM_KINDS = ["qkv", "proj", "fc", "proj2", "head"]
STOP_BIAS_C5S = 0.8              # mel_head.bias[1025]: makes the stop id reachable on the synthetic checkpoint (1.35, the value of the ragged
                                 # parity test in tests/test_gpu_baseline_size.py, ends a sequence after ~59 tokens; speech runs ~1.3 tokens per character)


def _log(msg: str):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _r(x, n=4):
    """round for the compact line"""
    if x is None:
        return None
    if isinstance(x, (list, tuple)):
        return [_r(v, n) for v in x]
    return float(f"{x:.{n}g}")


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(gpt_sd, xtts_sd, dims, cond, spk, text_ids, n_tokens: int = 280, with_c1: bool = False, activation: str = "gelu_new"):
    """CPU baseline (kind "port": the torch-CPU fp32 restatement of the reference path in oracle/) on this box's host cores.

    Sample = BASELINE configs[1] (C2) IN FULL: one 200-char utterance, 70 text ids, greedy, `n_tokens` = 280 mel tokens ->
    prefill + 279 decode steps + the reference's literal second pass (XTTSv2.py:617-687) + HiFi-GAN -> 312 064 samples,
    with per-stage times.  The thread count is swept first on a 4-token probe (more threads than the GEMV-sized matmuls can
    use makes the decode loop slower: round 1 measured 0.35 s/token at 64 threads against 36 ms/token at 8)."""
    from oracle import xtts_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gpt = O.GPTOracle(gpt_sd, xtts_sd, activation=activation)
    w = O.vocoder_effective_weights(xtts_sd)
    c = gpt.build_cond(cond, text_ids)
    sweep = {}
    for nt in sorted({t for t in (8, 16, 32) if t <= cores} | {min(cores, 8)}):
        torch.set_num_threads(nt)
        gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=2, ignore_stop=True))          # warm-up
        t0 = time.perf_counter()
        gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=4, ignore_stop=True))
        sweep[nt] = (time.perf_counter() - t0) / 4.0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    _log("cpu_baseline: thread sweep (s per token incl. prefill share) " + ", ".join(f"{k}: {v:.3f}" for k, v in sweep.items())
         + f" -> {best} threads")

    def run(ids, n):
        cc = gpt.build_cond(cond, ids)
        t0 = time.perf_counter()
        out = gpt.generate(cc, O.SamplingCfg(temperature=0.0, max_tokens=n, ignore_stop=True))
        t1 = time.perf_counter()
        lat = gpt.second_pass_latents(cc, out["tokens"])
        t2 = time.perf_counter()
        wav = O.hifi_decoder_forward(w, lat, spk)
        t3 = time.perf_counter()
        ns = wav.numel()
        return {"samples": ns, "wall_s": t3 - t0, "ar_tokens_s": t1 - t0, "second_pass_s": t2 - t1, "vocoder_s": t3 - t2,
                "samples_per_s": ns / (t3 - t0), "rtf": (t3 - t0) / (ns / 24000.0)}

    c2 = run(text_ids, n_tokens)
    out = {
        "value": c2["samples_per_s"], "unit": "audio-samples/s", "cores": best, "kind": "port",
        "sample": f"C2 in full: 1 utterance, 70 text ids, {n_tokens} greedy mel tokens -> {c2['samples']} samples in "
                  f"{c2['wall_s']:.1f} s (AR {c2['ar_tokens_s']:.1f}, second pass {c2['second_pass_s']:.1f}, HiFi-GAN "
                  f"{c2['vocoder_s']:.1f}); torch CPU fp32 oracle, {best} threads",
        "rtf": c2["rtf"], "host_cores": cores, "c2": c2, "thread_sweep_s_per_token": {str(k): v for k, v in sweep.items()},
    }
    if with_c1:   # BASELINE configs[0]: 50-char utterance, 18 text ids, 70 tokens
        out["c1_50char_70_tokens"] = run(list(text_ids[:17]) + [text_ids[-1]], 70)
    return out


# ------------------------------------------------------------------------------------------------ rooflines from a profile pass
def _traffic():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return {}


def kernel_rooflines(args, st, dims, samples_in_pass):
    """Per-kernel numbers from the engine's HIP-event timing of ONE profile pass (`st` = aur_stats of that pass).
    Decode GEMMs / attention: replay batches (one event pair per n_layer back-to-back launches of one kind, the step's real
    operands, outputs to scratch).  Vocoder convs: a pair per launch (0.2-2 ms each).  Prefill: the whole phase."""
    tj = _traffic()
    pmc_key = next((k for k in ("r06_decode", "r05_decode", "r04_decode", "r03_decode") if k in tj), "r03_decode")
    pmc = tj.get(pmc_key, {})
    gemm_peak = FP32_MFMA_PEAK_TFLOPS if args.gemm == "f32" else BF16_MFMA_PEAK_TFLOPS / 6.0

    def roof(name, ms, n, nbytes, flops, mfma_peak, pmc_name):
        if not n or ms <= 0:
            return None
        us = ms / n * 1e3
        gbps = (nbytes / n) / (us * 1e-6) / 1e9
        r = {"kernel": name, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
             "avg_launch_us": us, "launches_timed": n, "algorithmic_bytes_per_launch": nbytes / n, "traffic": None}
        pm = pmc.get(pmc_name) if pmc_name else None
        if pm:   # PMC FETCH_SIZE (x2) + WRITE_SIZE of committed rocprofv3 --pmc passes; the ratio carries over to this run's bytes
            r["traffic"] = pm["ratio_to_algorithmic"] * nbytes / n
            r["traffic_ratio"] = pm["ratio_to_algorithmic"]
            r["traffic_from_committed_profile"] = f"profiles/hbm_traffic.json[{pmc_key}]"
        if flops:
            tf = (flops / n) / (us * 1e-6) / 1e12
            t_hbm = (nbytes / n) / (HBM_PEAK_GBPS * 1e9)
            t_mfma = (flops / n) / (mfma_peak * 1e12)
            r["mfma"] = {"achieved": tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tf / mfma_peak}
            r["binding_roof"] = "hbm" if t_hbm >= t_mfma else "mfma"
            r["binding_floor_us"] = max(t_hbm, t_mfma) * 1e6
            r["binding_floor_frac"] = max(t_hbm, t_mfma) / (us * 1e-6)
        return r

    gemms = [roof("gemm_rows_kernel: " + GEMM_KINDS[k], st["gemm_kind_ms"][k], st["gemm_kind_launches"][k], st["gemm_kind_bytes"][k],
                  st["gemm_kind_flops"][k], gemm_peak, "gemm_" + GEMM_KINDS[k]) for k in range(5)]
    n_dec = max(1, st["decode_steps"])
    fam = None
    if all(gemms):
        # the family as it runs in a decode step: n_layer launches of each of the four block GEMMs.  The mel head (one launch per
        # step) is reported in per_kind_us but left OUT of the family's time and bytes: its replay batch re-reads ONE weight matrix
        # n_layer times, so all but the first launch are L2 / Infinity-Cache warm, unlike the launch inside a step
        L = args.layers
        w = [L, L, L, L, 0]
        us = sum(g["avg_launch_us"] * k for g, k in zip(gemms, w)) / sum(w)
        by = sum(g["algorithmic_bytes_per_launch"] * k for g, k in zip(gemms, w)) / sum(w)
        fl = sum(st["gemm_kind_flops"][k] / st["gemm_kind_launches"][k] * w[k] for k in range(5)) / sum(w)
        gbps = by / (us * 1e-6) / 1e9
        t_hbm, t_mfma = by / (HBM_PEAK_GBPS * 1e9), fl / (gemm_peak * 1e12)
        tr = [g.get("traffic_ratio") for g in gemms]
        fam = {"kernel": "gemm_rows_kernel (decode-regime GEMMs: qkv, attn proj, fc, mlp proj, mel head; every instantiation, "
                         "weighted as they occur in a decode step)",
               "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
               "avg_launch_us": us, "launches_timed": int(sum(g["launches_timed"] for g in gemms)),
               "algorithmic_bytes_per_launch": by,
               "frac_survey8d_fp16_bytes": gbps / 2 / HBM_PEAK_GBPS,
               "binding_roof": "hbm" if t_hbm >= t_mfma else "mfma", "binding_floor_frac": max(t_hbm, t_mfma) / (us * 1e-6),
               "four_gemms_per_layer_us": sum(g["avg_launch_us"] for g in gemms[:4]),
               "per_kind_us": {GEMM_KINDS[k]: gemms[k]["avg_launch_us"] for k in range(5)},
               "per_kind_frac": {GEMM_KINDS[k]: gemms[k]["frac"] for k in range(5)},
               "traffic": None,
               "measured_by": "HIP events around n_layer back-to-back launches per kind (replay of the step's launches, outputs to "
                              "scratch): the launch-to-launch period, as a kernel trace reports it"}
        if all(t is not None for t in tr):
            ratio = sum(gemms[k]["traffic_ratio"] * gemms[k]["algorithmic_bytes_per_launch"] * w[k] for k in range(5)) / (by * sum(w))
            fam["traffic"], fam["traffic_ratio"] = ratio * by, ratio
            fam["traffic_from_committed_profile"] = f"profiles/hbm_traffic.json[{pmc_key}]"
    attn = roof("paged_attention_kernel (decode: one query row per sequence against its paged K/V)", st["attn_ms"], st["attn_launches"],
                st["attn_bytes"], 0.0, 1.0, "attention")
    if attn and attn.get("traffic_ratio") is not None:
        # the algorithmic bytes count every sequence's whole context; the shared speaker prefix (32 tokens, two K/V blocks per voice) is
        # served from cache for all but the first reader, so fewer bytes MOVE: the fraction of the HBM peak on the PMC traffic
        # (FETCH_SIZE x 2 + WRITE_SIZE of the committed rocprofv3 --pmc passes, run at this bench's own --tokens)
        attn["frac_on_traffic"] = attn["frac"] * attn["traffic_ratio"]
        attn["pmc_tokens"] = pmc.get("tokens")
    # vocoder
    voc = None
    if st["conv_ms"] > 0:
        ms = st["conv_ms"]
        mf_peak = FP32_MFMA_PEAK_TFLOPS if args.vocoder == "fp32" else BF16_MFMA_PEAK_TFLOPS
        tf = st["conv_flops"] / (ms * 1e-3) / 1e12
        gb = st["conv_bytes"] / (ms * 1e-3) / 1e9
        names = ["resblocks_256ch", "resblocks_128ch", "resblocks_64ch", "resblocks_32ch", "conv_pre_and_transposed"]
        cls = {}
        for k in range(5):
            n, m = st["conv_class_launches"][k], st["conv_class_ms"][k]
            if n and m > 0:
                cls[names[k]] = {"launches": n, "ms": m, "frac_mfma": st["conv_class_flops"][k] / (m * 1e-3) / 1e12 / mf_peak,
                                 "frac_hbm_as_stored": st["conv_class_bytes"][k] / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        conv_pmc = (tj.get("r06_conv") or tj.get("r05_conv") or tj.get("r04_conv") or tj.get("r03_conv") or {}).get(f"conv_{args.vocoder}")
        voc = {"kernel": "conv1d_dma_f16_kernel + resblock_round_f16_kernel + conv1d_mfma_f16_kernel (HiFi-GAN convs)"
                         if args.vocoder == "fp16" else "conv1d_mfma_kernel",
               "ms_per_batch": ms / max(1, st["vocoder_batches"]), "launches": st["conv_launches"],
               "bound": "mfma", "achieved": tf, "peak": mf_peak, "unit": "TFLOP/s", "frac_mfma": tf / mf_peak,
               "frac_hbm_as_stored": gb / HBM_PEAK_GBPS, "bytes_per_sample_as_stored": st["conv_bytes"] / max(1, samples_in_pass),
               "frac_hbm_8d_fp16": VOC_BYTES_8D_FP16 * samples_in_pass / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "frac_hbm_8d_fp32": VOC_BYTES_8D_FP32 * samples_in_pass / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "by_class": cls,
               "traffic_ratio_as_stored": (conv_pmc["bytes_per_launch"] * st["conv_launches"] / st["conv_bytes"]) if conv_pmc and st["conv_bytes"] else None}
    # prefill (north_star: ">= 40 % MFMA util on GPT prefill"): matmul flops of the prompt rows over the event-timed prefill phases
    prefill = None
    if st["prefill_ms"] > 0:
        fl = 2.0 * MATMUL_PARAMS_PER_LAYER * args.layers * st["prefill_rows"]
        tf = fl / (st["prefill_ms"] * 1e-3) / 1e12
        prefill = {"kernel": "gemm_tile_split_kernel (+ prompt attention, LayerNorm, embedding launches of the phase)" if args.gemm != "f32"
                             else "gemm_tile_kernel", "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac_f32_peak": tf / FP32_MFMA_PEAK_TFLOPS, "prefill_rows": st["prefill_rows"],
                   "ms_per_batch": st["prefill_ms"] / max(1, st["vocoder_batches"]),
                   "matrix_pipe_busy": (6.0 * tf / BF16_MFMA_PEAK_TFLOPS) if args.gemm != "f32" else tf / FP32_MFMA_PEAK_TFLOPS}
    est = {}
    if fam and attn:
        est = {"gemm_rows_kernel": fam["avg_launch_us"] * (4 * args.layers + 1) * n_dec * 1e-3,
               "paged_attention_kernel": attn["avg_launch_us"] * args.layers * n_dec * 1e-3,
               "vocoder_convs": st["conv_ms"], "prefill_phase": st["prefill_ms"]}
    return {"gemm_family": fam, "gemm_kinds": gemms, "attention": attn, "vocoder": voc, "prefill": prefill,
            "est_gpu_ms_in_pass": est, "event_pair_overhead_us": st["event_pair_overhead_ms"] * 1e3}


def decode_step_roofline(st):
    if not st["decode_steps"]:
        return None
    ms = st["decode_ms"] / st["decode_steps"]
    b32 = (st["decode_weight_bytes"] + st["decode_kv_bytes"]) / st["decode_steps"]
    return {"ms": ms, "steps": st["decode_steps"], "bytes_per_step_as_stored": b32,
            "frac_as_stored": b32 / ms / 1e6 / HBM_PEAK_GBPS, "frac_8d_fp16": b32 / 2 / ms / 1e6 / HBM_PEAK_GBPS,
            "rows_per_step": st["decode_rows"] / st["decode_steps"],
            "note": "fp32 weights and K/V are what this engine stores and streams (bit-exact contract against the fp32 CPU path); "
                    "SURVEY 8(d) quotes the reference GPU path's fp16 storage, i.e. half the bytes for the same step"}


# ------------------------------------------------------------------------------------------------ workloads
class Bench:
    def __init__(self, args, rank, world, local_rank, use_dist):
        self.args, self.rank, self.world, self.local_rank, self.use_dist = args, rank, world, local_rank, use_dist
        from auralis_amd._lib import NativeEngine
        from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_text_ids,
                                            make_synthetic_xtts)
        from auralis_amd.config import XTTSDims
        from auralis_amd.weights import pack_all
        self.dims = XTTSDims()
        gelu_erf = False
        if args.checkpoint:   # real weights (the reference's on-disk format, XTTSv2.py:276-308); shapes and config are checked on load
            from auralis_amd.checkpoint import load_checkpoint, read_checkpoint_config
            _log(f"loading checkpoint {args.checkpoint}")
            self.gpt_sd, self.xtts_sd = load_checkpoint(args.checkpoint)
            ck = read_checkpoint_config(args.checkpoint, self.gpt_sd)
            args.layers, gelu_erf = ck.n_layer, ck.gelu_erf
        else:
            _log("building synthetic checkpoint")
            self.gpt_sd = make_synthetic_gpt(self.dims.gpt, seed=1234, n_layer=args.layers)
            self.xtts_sd = make_synthetic_xtts(self.dims, seed=1234, gpt_sd=self.gpt_sd)
        # profile=False: the engine configuration the parity tests run (tests/test_gpu_baseline_size.py)
        self.eng = NativeEngine(n_layer=args.layers, max_seqs=args.batch, device=local_rank, profile=False,
                                vocoder_fp16=(args.vocoder == "fp16"), return_latents=False,   # audio + tokens, as TTSOutput
                                kv_fp16=(args.kv == "fp16"), gemm_f32_exact=(args.gemm == "f32"), admit_min_batch=args.admit_min_batch,
                                vocoder_min_batch=args.vocoder_min_batch, gelu_erf=gelu_erf)
        self.gelu_erf = gelu_erf
        self.packed = pack_all(self.gpt_sd, self.xtts_sd)
        self.eng.load_weights(self.packed)
        _log("weights resident")
        self.cond, self.spk = make_synthetic_conditioning(self.dims)
        if args.speaker:      # .npz with gpt_cond_latent [1,32,1024] and speaker_embedding [1,512,1] (XTTSv2Engine.get_audio_conditioning's output)
            z = np.load(args.speaker)
            self.cond = torch.from_numpy(np.asarray(z["gpt_cond_latent"], np.float32).reshape(1, 32, 1024))
            self.spk = torch.from_numpy(np.asarray(z["speaker_embedding"], np.float32).reshape(1, 512, 1))
        self.text_ids = make_synthetic_text_ids(self.dims, n_text=70, seed=11)
        tok_file = os.path.join(args.checkpoint, "tokenizer.json") if args.checkpoint else ""
        if tok_file and os.path.isfile(tok_file):   # the checkpoint's own tokenizer on a 200-character English sentence
            from auralis_amd.api.text import XTTSTokenizer
            txt = ("The old lighthouse keeper climbed the spiral stairs every evening, counted the ships on the horizon, wrote their names "
                   "in a worn leather book, and wondered which of them would still be sailing when the winter storms arrived.")
            self.text_ids = XTTSTokenizer(tok_file, vocab_size=self.xtts_sd["text_embedding.weight"].shape[0]).batch_encode_with_split(txt, "en")[0]
            _log(f"text ids from the checkpoint's tokenizer: {len(self.text_ids)} ids for {len(txt)} characters")
        self.make_ids = lambda n, seed: make_synthetic_text_ids(self.dims, n_text=n, seed=seed)
        self.SPK = 1

    def fence(self):
        if self.use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        self.eng.sync()

    def run_batches(self, keys, batch=None, sampled=True):
        """one batch of `batch` utterances per key, one after the other; results consumed in place (views of the engine's pinned
        result blocks: the D2H copies are part of the step), released once counted"""
        a, eng = self.args, self.eng
        batch = batch or a.batch
        total = 0
        groups = [list(keys)] if a.pipeline else [[k] for k in keys]
        for grp in groups:
            t_grp = time.perf_counter()
            for k in grp:
                for b in range(batch):
                    eng.submit(self.text_ids, self.SPK, temperature=0.75 if sampled else 0.0, top_p=0.85, top_k=50, repetition_penalty=5.0,
                               max_tokens=a.tokens, seed=(self.rank * 100003 + (k + 7) * 1009 + b), ignore_stop=True)
            if os.environ.get("AUR_BENCH_STEP_TRACE"):   # host time of every aur_step of the batch: where the wall time outside the GPU phases goes
                outs, marks, t_sub = [], [], time.perf_counter()
                while True:
                    t0 = time.perf_counter()
                    live, fin = eng.step()
                    t1 = time.perf_counter()
                    got = eng.poll(cap=64, copy=False) if fin or live == 0 else []
                    outs.extend(got)
                    marks.append((t1 - t0, time.perf_counter() - t1, live, len(got)))
                    if live == 0:
                        break
                tot = time.perf_counter() - t_sub
                mid = sorted(m[0] for m in marks[2:-4])
                _log(f"step trace: submits {(t_sub - t_grp) * 1e3:.2f} ms; {len(marks)} aur_step calls in {tot * 1e3:.2f} ms; first two {marks[0][0] * 1e3:.2f} / {marks[1][0] * 1e3:.2f} ms, "
                     f"median of the middle {mid[len(mid) // 2] * 1e3:.3f} ms (sum {sum(mid) * 1e3:.1f}), last four "
                     + " / ".join(f"{m[0] * 1e3:.2f}+poll {m[1] * 1e3:.2f} (live {m[2]}, got {m[3]})" for m in marks[-4:]))
            else:
                outs = eng.run_until_done(max_steps=len(grp) * (a.tokens + 16) + 64, copy=False)
            assert len(outs) == batch * len(grp)
            total += sum(len(o["wav"]) for o in outs)
            for o in outs:
                assert o["error"] == 0
                eng.release(o["seq_id"])
        return total

    # -- c2: BASELINE configs[1], one utterance, greedy, batch 1 (the regime the reference publishes latencies for, README.md:483-485)
    def workload_c2(self, reps=3):
        a, eng = self.args, self.eng
        walls, stl = [], []
        for r in range(reps + 1):
            eng.reset_stats()
            self.fence()
            t0 = time.perf_counter()
            eng.submit(self.text_ids, self.SPK, temperature=0.0, top_p=0.85, top_k=50, repetition_penalty=5.0, max_tokens=a.tokens,
                       seed=r, ignore_stop=True)
            outs = eng.run_until_done(copy=False)
            dt = time.perf_counter() - t0
            ns = len(outs[0]["wav"])
            eng.release(outs[0]["seq_id"])
            if r:   # first repetition = warm-up (buffer growth)
                walls.append(dt)
                stl.append(eng.stats())
        i = int(np.argsort(walls)[len(walls) // 2])
        st, dt = stl[i], walls[i]
        ds = decode_step_roofline(st)
        return {"workload": "BASELINE configs[1]: 1 x 200-char utterance (70 text ids), greedy, rep_pen 5.0, 280 mel tokens fixed-length, batch 1",
                "time_to_audio_ms": dt * 1e3, "samples": ns, "audio_s": ns / 24000.0, "rtf": dt / (ns / 24000.0), "samples_per_s": ns / dt,
                "prefill_ms": st["prefill_ms"], "decode_ms": st["decode_ms"], "vocoder_ms": st["vocoder_ms"],
                "decode_step": ds, "wall_ms_all_reps": [w * 1e3 for w in walls],
                "hbm_floor_ms_per_step": (ds["bytes_per_step_as_stored"] / (HBM_PEAK_GBPS * 1e9) * 1e3) if ds else None}

    # -- c3f: the headline workload at the DROP-IN boundary: the same 64 utterances as 64 concurrent TTSRequests through the facade
    #    (TTS.generate_speech_async -> TwoPhaseScheduler -> XTTSv2Engine plugin -> EngineDriver thread -> C ABI), what a user of the
    #    reference's generate_speech() gets (core/tts.py:196-233, XTTSv2.py:762-814).  Same engine, same sampling, fixed length.
    C3F_TEXT = ("the old lighthouse keeper climbed the spiral stairs every evening and counted the ships on the horizon while the "
                "wind pushed grey clouds over the waters and the gulls went quiet above the harbour wall")   # 200 characters

    def workload_c3f(self, reps=3):
        import asyncio

        from auralis_amd import TTS, TTSRequest
        from auralis_amd.api.text import XTTSTokenizer
        from auralis_amd.api.xtts_engine import XTTSv2Engine
        a, eng = self.args, self.eng
        tok = XTTSTokenizer(None, vocab_size=self.xtts_sd["text_embedding.weight"].shape[0], synthetic=True)
        n_ids = len(tok.batch_encode_with_split(self.C3F_TEXT, "en")[0])
        assert n_ids == len(self.text_ids), f"c3f text gives {n_ids} ids, the headline uses {len(self.text_ids)}"
        xe = XTTSv2Engine(eng, tok, max_concurrency=a.batch, gpt_max_audio_tokens=a.tokens)
        xe.fixed_length = True
        tts = TTS(scheduler_max_concurrency=a.batch).with_engine(xe)
        voice = {"gpt_cond_latent": self.cond.numpy(), "speaker_embedding": self.spk.numpy()}

        async def batch(k):
            reqs = [TTSRequest(text=self.C3F_TEXT, speaker_files=[voice], language="en", temperature=0.75, top_p=0.85, top_k=50,
                               repetition_penalty=5.0, seed=(self.rank * 100003 + (k + 7) * 1009 + b)) for b in range(a.batch)]
            outs = await asyncio.gather(*[tts.generate_speech_async(r) for r in reqs])
            return sum(len(o.array) for o in outs), sum(int(o.token_length or 0) for o in outs)

        def run(k):
            return asyncio.run_coroutine_threadsafe(batch(k), tts._loop).result()
        try:
            run(-1)   # warm-up
            eng.reset_stats()
            self.fence()
            t0 = time.perf_counter()
            ns = nt = 0
            for k in range(reps):
                s_, t_ = run(k)
                ns += s_
                nt += t_
            self.fence()
            dt = time.perf_counter() - t0
            st = eng.stats()
        finally:
            tts.close(keep_engine=True)
        return {"workload": f"BASELINE configs[2] through the facade: {a.batch} concurrent TTSRequest(text=<{len(self.C3F_TEXT)} chars>, "
                            f"speaker_files=[voice], seed=...) -> TTS(scheduler_max_concurrency={a.batch}).generate_speech_async, {n_ids} text ids, "
                            f"{a.tokens} mel tokens fixed-length, T=0.75 top_p=0.85 top_k=50 rep_pen=5.0; {reps} batches",
                "reps": reps, "ms_per_step": dt / reps * 1e3, "samples": ns, "tokens": nt, "samples_per_s": ns / dt, "rtf": dt / (ns / 24000.0),
                "prefill_batches_per_step": st["prefill_batches"] / reps, "vocoder_batches_per_step": st["vocoder_batches"] / reps,
                "decode_steps_per_step": st["decode_steps"] / reps, "failed_steps": xe.driver.failed_steps}

    # -- c5s: BASELINE configs[4] at single-GPU scale: mixed-language long form through the facade, natural stop, ragged
    def workload_c5s(self, chars=56250, window=None):
        from auralis_amd import TTS
        from auralis_amd.api.text import XTTSTokenizer
        from auralis_amd.api.xtts_engine import XTTSv2Engine
        from auralis_amd.longform import build_requests, default_window, stream_longform
        a, eng = self.args, self.eng
        slots = a.c5s_slots or a.batch
        own = None
        if slots != a.batch:   # the long-form knob is the facade's scheduler_max_concurrency (core/tts.py:20-51): its own engine, same weights
            from auralis_amd._lib import NativeEngine
            own = eng = NativeEngine(n_layer=a.layers, max_seqs=slots, device=self.local_rank, profile=False, vocoder_fp16=(a.vocoder == "fp16"),
                                     return_latents=False, kv_fp16=(a.kv == "fp16"), gemm_f32_exact=(a.gemm == "f32"),
                                     admit_min_batch=a.admit_min_batch, vocoder_min_batch=a.vocoder_min_batch, gelu_erf=self.gelu_erf,
                                     urgent_rows=a.urgent_rows)
            eng.load_weights(self.packed)
            eng.set_conditioning(self.SPK, self.cond.numpy(), self.spk.numpy())
        EN = ("It was a bright cold day in April, and the clocks were striking thirteen. Nobody in the street seemed to notice, "
              "and the wind kept pushing the dust along the old road as if nothing had happened at all. ")
        FR = ("Il était une fois, dans une petite ville que nous ne connaissons pas, un homme qui avait beaucoup d'idées et très peu "
              "de temps pour les écrire. Il marchait chaque matin le long de la rivière avec son chien. ")
        DE = ("Es war einmal ein Mann, der nicht mit dem Zug fahren wollte und auch nicht zu Fuß gehen konnte, weil der Weg durch "
              "den Wald zu lang war. Also blieb er zu Hause und schrieb Briefe an seine Freunde. ")
        rng = np.random.default_rng(5)
        paras = []
        while sum(len(p) for p in paras) < chars:
            for s in (EN, FR, DE):
                paras.append((s * int(rng.integers(1, 4)))[: int(rng.integers(90, 600))].strip())
        # the stop id becomes reachable (as in the ragged baseline-size parity test); sequences end on it or at 605 tokens
        hb = np.array(self.packed["mel_head.b"], dtype=np.float32, copy=True)
        hb[1025] = STOP_BIAS_C5S
        eng.load_weights({"mel_head.b": hb})
        voice = {"gpt_cond_latent": self.cond.numpy(), "speaker_embedding": self.spk.numpy()}
        reqs = build_requests(paras, [voice], seed=3)   # request defaults: T 0.75 / top_p 0.85 / top_k 50 / rep_pen 5.0
        xe = XTTSv2Engine(eng, XTTSTokenizer(None, vocab_size=self.xtts_sd["text_embedding.weight"].shape[0], synthetic=True),
                          max_concurrency=slots)
        tts = TTS(scheduler_max_concurrency=slots).with_engine(xe)
        inflight = tts.scheduler.second_phase_concurrency   # the facade's gate = slots, as in the reference; the window's other chunks queue inside the engine
        window = window or default_window(tts)              # paragraphs in flight (>= 1 chunk each)
        try:
            eng.reset_stats()
            self.fence()
            t0 = time.perf_counter()
            first, n_chunks, ns, order_ok, last = None, 0, 0, True, -1
            toks, stamps = [], []
            for i, c in stream_longform(tts, reqs, window=window):
                stamps.append(time.perf_counter() - t0)
                if first is None:
                    first = stamps[0]
                order_ok &= i >= last
                last = i
                n_chunks += 1
                ns += len(c.array)
                toks.append(int(c.token_length or 0))
            dt = time.perf_counter() - t0
            st = eng.stats()
        finally:
            tts.close(keep_engine=True)   # the bench still needs the native engine
            if own is not None:
                own.close()
            else:
                eng.load_weights({"mel_head.b": np.asarray(self.packed["mel_head.b"], np.float32)})
        occ = st["decode_rows"] / max(1, st["decode_steps"]) / slots
        gaps = np.diff(np.asarray(stamps)) if len(stamps) > 1 else np.zeros(1)
        return {"first_chunk_tokens": toks[0] if toks else None, "p95_chunk_gap_s": float(np.percentile(gaps, 95)), "max_chunk_gap_s": float(gaps.max()),
                "workload": f"BASELINE configs[4] at 1-GPU scale: {sum(len(p) for p in paras)} chars, {len(paras)} paragraphs en/fr/de "
                            f"(language=auto), {n_chunks} chunks, natural stop (mel_head.bias[1025] = {STOP_BIAS_C5S}), {window} paragraphs in flight (every chunk of them submitted; facade gate {inflight}) "
                            f"on {slots} slots (admit_min_batch {a.admit_min_batch or max(1, slots // 8)}), streamed in (paragraph, chunk) order "
                            f"through TTS / longform.stream_longform",
                "slots": slots, "chars": sum(len(p) for p in paras), "paragraphs": len(paras), "chunks": n_chunks, "in_order": bool(order_ok),
                "wall_s": dt, "first_chunk_s": first, "samples": ns, "audio_s": ns / 24000.0, "samples_per_s": ns / dt, "rtf": dt / max(1e-9, ns / 24000.0),
                "chars_per_s": sum(len(p) for p in paras) / dt, "slot_occupancy": occ,
                "tokens_per_chunk": {"mean": float(np.mean(toks)), "min": int(np.min(toks)), "max": int(np.max(toks))} if toks else None,
                "decode_steps": st["decode_steps"], "decode_ms_per_step": st["decode_ms"] / max(1, st["decode_steps"]),
                "gpt_ms": st["gpt_ms"], "prefill_ms": st.get("prefill_ms"), "vocoder_ms": st["vocoder_ms"], "vocoder_batches": st["vocoder_batches"]}

    # -- c4: BASELINE configs[3]: 512 utterances dealt 64 at a time to the ranks (strong scaling; 8 GPUs -> one batch each)
    def workload_c4(self, n_units=512):
        from auralis_amd.parallel import merge_ordered, shard_units
        a, eng = self.args, self.eng
        mine = shard_units(n_units, self.world, self.rank, per_gpu_batch=a.batch)
        self.fence()
        t0 = time.perf_counter()
        digests, ns = [], 0
        for start in range(0, len(mine), a.batch):
            blk = mine[start:start + a.batch]
            ids = {}
            for u in blk:
                ids[eng.submit(self.text_ids, self.SPK, temperature=0.75, top_p=0.85, top_k=50, repetition_penalty=5.0, max_tokens=a.tokens,
                               seed=u, ignore_stop=True)] = u
            for o in eng.run_until_done(copy=False):
                assert o["error"] == 0
                hh = hashlib.blake2b(digest_size=8)
                hh.update(np.asarray(o["tokens"], np.int32).tobytes())
                digests.append((ids[o["seq_id"]], len(o["wav"]), hh.hexdigest()))
                ns += len(o["wav"])
                eng.release(o["seq_id"])
        self.fence()
        dt = time.perf_counter() - t0
        if self.use_dist:
            per = [None] * self.world
            torch.distributed.all_gather_object(per, digests)
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        else:
            per = [digests]
        merged = merge_ordered(per)
        total = sum(n for _, n, _ in merged)
        return {"workload": f"BASELINE configs[3]: {n_units} x 200-char utterances dealt {a.batch} at a time to {self.world} rank(s) by "
                            "parallel.shard_units, per-unit seeds, results merged by unit index (parallel.merge_ordered)",
                "units": n_units, "units_returned_in_order": [u for u, _, _ in merged] == list(range(n_units)),
                "wall_s": dt, "samples": total, "samples_per_s": total / dt, "rtf": dt / (total / 24000.0), "scaling": "strong",
                "ids_digest": hashlib.blake2b("".join(h for _, _, h in merged).encode(), digest_size=8).hexdigest()}


def compact(line, full_path):
    """The driver-facing line: contract fields + roofline + cpu_baseline + one summary per side workload, < 4 KB."""
    k = line["kernels"] or {}
    fam, attn, voc, pre = k.get("gemm_family"), k.get("attention"), k.get("vocoder"), k.get("prefill")
    out = {key: line[key] for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data", "rtf", "config")}
    out["value"], out["ms_per_step"], out["rtf"] = _r(out["value"], 6), _r(out["ms_per_step"], 6), _r(out["rtf"], 4)
    if fam:
        out["roofline"] = {"kernel": "gemm_rows_kernel (decode GEMM family)", "bound": fam["bound"], "achieved": _r(fam["achieved"]),
                           "peak": fam["peak"], "unit": fam["unit"], "frac": _r(fam["frac"]), "traffic": _r(fam["traffic"]),
                           "traffic_ratio": _r(fam.get("traffic_ratio")), "traffic_from_committed_profile": fam.get("traffic_from_committed_profile"),
                           "frac_survey8d_fp16_bytes": _r(fam["frac_survey8d_fp16_bytes"]), "binding_roof": fam["binding_roof"],
                           "binding_floor_frac": _r(fam["binding_floor_frac"]), "avg_launch_us": _r(fam["avg_launch_us"]),
                           "launches_timed": fam["launches_timed"], "four_gemms_per_layer_us": _r(fam["four_gemms_per_layer_us"]),
                           "per_kind_us": {a: _r(b, 3) for a, b in fam["per_kind_us"].items()},
                           "share_of_gpu_time": _r(line.get("dominant_share")), "in_run": "HIP events, replay batches in a profile pass"}
    else:
        out["roofline"] = None
    if attn:
        out["roofline_attention"] = {"frac": _r(attn["frac"]), "frac_on_traffic": _r(attn.get("frac_on_traffic")), "achieved": _r(attn["achieved"]),
                                     "avg_launch_us": _r(attn["avg_launch_us"]), "traffic_ratio": _r(attn.get("traffic_ratio")),
                                     "traffic_measured_at_tokens": attn.get("pmc_tokens")}
    if voc:
        out["roofline_vocoder"] = {"frac_mfma": _r(voc["frac_mfma"]), "frac_hbm_as_stored": _r(voc["frac_hbm_as_stored"]),
                                   "frac_hbm_8d_fp16": _r(voc["frac_hbm_8d_fp16"]), "frac_hbm_8d_fp32": _r(voc["frac_hbm_8d_fp32"]),
                                   "conv_ms_per_batch": _r(voc["ms_per_batch"]),
                                   "frac_mfma_by_class": {a: _r(b["frac_mfma"], 3) for a, b in voc["by_class"].items()}}
    if pre:
        out["prefill_roofline"] = {"frac_f32_peak": _r(pre["frac_f32_peak"]), "matrix_pipe_busy": _r(pre["matrix_pipe_busy"]),
                                   "ms_per_batch": _r(pre["ms_per_batch"])}
    ds = line.get("decode_step")
    if ds:
        out["decode_step"] = {"ms": _r(ds["ms"]), "frac_as_stored": _r(ds["frac_as_stored"]), "frac_8d_fp16": _r(ds["frac_8d_fp16"])}
    if ds and fam and attn and out.get("roofline"):
        # IN SITU: what the four GEMMs of a layer cost inside the timed region's own decode steps (HIP events around every step, profile
        # mode off) = (step - tail) / layers - attention, tail = embed + final norms + mel head + sampler (the head's replay time + 25 us).
        # The replay batches above re-issue one kind back to back; in a step the kinds alternate (proj runs ~15 % slower there).
        L = line["config"]["gpt_layers"]
        tail_us = fam["per_kind_us"]["head"] + 25.0
        g4 = (ds["ms"] * 1e3 - tail_us) / L - attn["avg_launch_us"]
        by4 = 4.0 * fam["algorithmic_bytes_per_launch"]   # (the family average is over the four block GEMMs)
        out["roofline"]["four_gemms_per_layer_us_in_situ"] = _r(g4)
        out["roofline"]["frac_in_situ"] = _r(by4 / (g4 * 1e-6) / 1e9 / HBM_PEAK_GBPS)
    out["breakdown_ms_per_step"] = {a: _r(b) for a, b in line["breakdown_ms_per_step"].items()}
    c2 = line.get("c2")
    if c2 and "error" not in c2:
        out["c2"] = {"time_to_audio_ms": _r(c2["time_to_audio_ms"]), "rtf": _r(c2["rtf"]), "decode_ms_per_step": _r(c2["decode_step"]["ms"]),
                     "decode_frac_hbm_as_stored": _r(c2["decode_step"]["frac_as_stored"]), "prefill_ms": _r(c2["prefill_ms"]),
                     "vocoder_ms": _r(c2["vocoder_ms"])}
    elif c2:
        out["c2"] = c2
    c3f = line.get("c3f")
    if c3f and "error" not in c3f:
        out["c3f"] = {"ms_per_step": _r(c3f["ms_per_step"], 5), "samples_per_s": _r(c3f["samples_per_s"], 5),
                      "facade_overhead_ms": _r(c3f["ms_per_step"] - line["ms_per_step"], 3),
                      "facade_overhead_frac": _r(c3f["ms_per_step"] / line["ms_per_step"] - 1.0, 3),
                      "prefill_batches_per_step": _r(c3f["prefill_batches_per_step"], 3),
                      "vocoder_batches_per_step": _r(c3f["vocoder_batches_per_step"], 3)}
    elif c3f:
        out["c3f"] = c3f
    c5 = line.get("c5s")
    if c5 and "error" not in c5:
        out["c5s"] = {"slots": c5.get("slots"), "chars": c5["chars"], "chunks": c5["chunks"], "samples_per_s": _r(c5["samples_per_s"]), "rtf": _r(c5["rtf"]),
                      "slot_occupancy": _r(c5["slot_occupancy"]), "decode_ms_per_step": _r(c5.get("decode_ms_per_step")), "first_chunk_s": _r(c5["first_chunk_s"]),
                      "first_chunk_tokens": c5.get("first_chunk_tokens"), "p95_chunk_gap_s": _r(c5.get("p95_chunk_gap_s")), "in_order": c5["in_order"]}
    elif c5:
        out["c5s"] = c5
    c4 = line.get("c4")
    if c4:
        out["c4"] = {a: (_r(b) if isinstance(b, float) else b) for a, b in c4.items() if a != "workload"}
    cb = line.get("cpu_baseline")
    out["cpu_baseline"] = ({"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "host_cores": cb["host_cores"], "kind": cb["kind"],
                            "rtf": _r(cb["rtf"]), "sample": cb["sample"]} if cb else None)
    if line.get("multi_gpu"):
        out["multi_gpu"] = line["multi_gpu"]
    out["full_report"] = full_path
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c3", "c3f", "c2", "c5s", "c4"], default="c3",
                    help="c3 (default) = BASELINE configs[2], the headline; c3f (the same through the TTS facade) / c2 / c5s / c4 run only "
                         "that workload and print its record")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--tokens", type=int, default=280, help="mel tokens per utterance (fixed-length mode)")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--vocoder", choices=["fp32", "fp16"], default="fp16",
                    help="MFMA input type of the HiFi-GAN convs (fp32 accumulate either way; GPT is fp32)")
    ap.add_argument("--kv", choices=["fp32", "fp16"], default="fp32",
                    help="paged K/V pool dtype: fp32 = the bit-exact parity mode (default, the reported metric); fp16 = opt-in "
                         "throughput mode (aur_config.kv_fp16), half the attention bytes, ids may differ after a near-tie")
    ap.add_argument("--gemm", choices=["bf16x3", "f32"], default="bf16x3",
                    help="GEMM arithmetic: bf16x3 = exact 3-way bf16 split of the fp32 operands, 6 bf16 MFMAs per product, fp32 "
                         "accumulate (default, what the parity tests run); f32 = v_mfma_f32_*_f32 (aur_config.gemm_f32_exact)")
    ap.add_argument("--bcast", choices=["auto", "native", "torch"], default="auto",
                    help="multi-GPU launches, the conditioning broadcast BEFORE the measurements.  auto (default) = the ncclBroadcast inside the "
                         "library on the engine's own RCCL communicator (aur_comm_init / aur_broadcast_conditioning) when aur_comm_init "
                         "succeeded on EVERY rank (agreed with one all_reduce before any rank enters the collective), otherwise "
                         "torch.distributed.broadcast into a device buffer registered with aur_set_conditioning_device, with the reason in "
                         "multi_gpu.native_route_failed.  native = the same without the torch fallback for a failed init; torch = the torch "
                         "route (the in-library one is then exercised after the measurements, multi_gpu.native_route).  The native "
                         "broadcast runs under a watchdog (--native-check-timeout): a hang ends the run with a diagnostic instead of a silent stall")
    ap.add_argument("--native-check-timeout", type=float, default=90.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the c2 / c5s (and, N > 1, c4) measurements after the headline")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--profile-every", type=int, default=8, help="profile pass: replay batches after every n-th decode step")
    ap.add_argument("--c5-chars", type=int, default=56250,
                    help="characters of the c5s long-form text; default = one GPU's eighth of BASELINE configs[4]'s ~450 k characters")
    ap.add_argument("--c5s-slots", type=int, default=192,
                    help="slots (TTS scheduler_max_concurrency, the reference's long-form knob, core/tts.py:20-51) of the c5s measurement; "
                         "0 = --batch.  A value other than --batch builds a second engine for that workload (same weights, its own K/V "
                         "pool).  Default 192: the decode GEMMs stream the weights once per step whatever the rows, so a long-form job is "
                         "cheaper per token on more rows (profiles/r05_c5s_slots_sweep.jsonl: 64 slots 26.3 M samples/s, 128: 28.4, 192: 30.3)")
    ap.add_argument("--urgent-rows", type=int, default=0,
                    help="aur_config.urgent_rows (c5s: running sequences the engine fills up to while the stream's head chunk -- TTSRequest.priority, "
                         "set by longform.build_requests -- waits or runs; 0 = the engine's default, slots / 4)")
    ap.add_argument("--admit-min-batch", type=int, default=0,
                    help="aur_config.admit_min_batch (0 = the engine's default, slots / 8; 1 = admit one by one); matters for c5s only")
    ap.add_argument("--vocoder-min-batch", type=int, default=0,
                    help="aur_config.vocoder_min_batch: finished sequences wait for this many before a vocoder batch is launched (0/1 = at once)")
    ap.add_argument("--checkpoint", default="",
                    help="a checkpoint directory in the reference's on-disk format (XTTSv2.py:276-308) instead of the seeded synthetic "
                         "weights: layer count, activation and tokenizer come from it.  The line's `data` then says so; generation stays "
                         "in fixed-length mode (--tokens) so that runs are comparable")
    ap.add_argument("--speaker", default="", help=".npz with gpt_cond_latent / speaker_embedding to use instead of the seeded synthetic voice")
    ap.add_argument("--pipeline", action="store_true",
                    help="queue all steps at once so the vocoder of batch k overlaps the GPT of batch k+1 (measured neutral)")
    ap.add_argument("--cpu-tokens", type=int, default=280, help="mel tokens of the CPU-baseline utterance (280 = C2 in full)")
    ap.add_argument("--cpu-c1", action="store_true", help="also time BASELINE configs[0] on the CPU oracle")
    ap.add_argument("--out", default=os.path.join("gpurun_out", "bench_full.json"))
    args = ap.parse_args()

    # stdout carries exactly ONE line, the compact JSON record: everything native libraries print to file descriptor 1 (RCCL's
    # version banner at communicator creation, for one) is sent to stderr instead, and the record goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # any torchrun launch, even N = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:   # N ranks build the synthetic checkpoint at the same time on one host: share the cores instead of N x all of them
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    B = Bench(args, rank, world, local_rank, use_dist)
    eng, dims, SPK = B.eng, B.dims, B.SPK

    # speaker conditioning: computed on rank 0, RCCL-broadcast over xGMI, registered from the device buffer
    multi = {}
    route = args.bcast
    dev = torch.device("cuda", local_rank)
    if use_dist and route in ("auto", "native"):
        # the collective inside the library: one ncclBroadcast of 133 120 B on the engine's own RCCL communicator.  First every rank
        # builds its communicator and the ranks AGREE that all of them are up (parallel.comm_init_agreed); only then does anyone
        # enter the broadcast.  The route has only ever run at world size 1 in the build environment, so the broadcast itself runs
        # under a watchdog: a rank stuck in it cannot be recovered, the run then ends with a diagnostic on stderr.
        import threading
        from auralis_amd.parallel import broadcast_conditioning_native, comm_init_agreed

        def watched(fn, what):   # run a step that contains a collective under the watchdog; a rank stuck inside it cannot be recovered
            box = {}

            def run():
                try:
                    torch.cuda.set_device(local_rank)   # the current device is per thread: object collectives stage through it
                    box["v"] = fn()
                except Exception as ex:   # noqa: BLE001 - handed back to the caller
                    box["e"] = ex
            th = threading.Thread(target=run, daemon=True)
            th.start()
            th.join(args.native_check_timeout)
            if th.is_alive():
                _log(f"rank {rank}: {what} did not return within {args.native_check_timeout:.0f} s; re-run with --bcast torch")
                sys.stderr.flush()
                os._exit(3)
            if "e" in box:
                raise box["e"]
            return box.get("v")
        err = watched(lambda: comm_init_agreed(eng, dev), "aur_comm_init (ncclCommInitRank on the engine's communicator)")
        multi["native_route_failed"] = False
        if err:
            if route == "native":
                raise SystemExit(f"--bcast native: {err}")
            multi["native_route_failed"] = err[:160]
            _log(f"in-library RCCL route not available ({err}); falling back to torch.distributed.broadcast")
            route = "torch"
        else:
            res = {}
            try:
                watched(lambda: broadcast_conditioning_native(eng, SPK, B.cond if rank == 0 else None, B.spk if rank == 0 else None, src=0,
                                                              device=dev), "ncclBroadcast on the engine's communicator")
                res["ok"] = True
            except Exception as ex:   # noqa: BLE001 - agreed on below
                res["err"] = f"{type(ex).__name__}: {ex}"
            bad = torch.tensor([0 if res.get("ok") else 1], device=dev, dtype=torch.int32)
            torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
            if int(bad.item()):   # the collective returned an error somewhere: the voice is re-sent by the torch route on every rank
                multi["native_route_failed"] = (res.get("err") or "aur_broadcast_conditioning failed on another rank")[:160]
                _log(f"in-library broadcast failed ({multi['native_route_failed']}); falling back to torch.distributed.broadcast")
                route = "torch"
            else:
                route = "native"
                multi["rccl_ranks"], multi["rccl_rank_of_rank0"] = eng.comm_info()   # what the communicator itself reports
    if use_dist and route == "torch":
        from auralis_amd.parallel import broadcast_conditioning
        broadcast_conditioning(eng, SPK, B.cond if rank == 0 else None, B.spk if rank == 0 else None, src=0,
                               device=torch.device("cuda", local_rank))
        multi["rccl_ranks"] = torch.distributed.get_world_size()
    elif not use_dist:
        eng.set_conditioning(SPK, B.cond.numpy(), B.spk.numpy())
    if use_dist:
        # self-verification of the multi-GPU path, outside the timed region: (1) the voice every rank holds in device memory is
        # byte-identical to rank 0's; (2) the SAME small workload (same prompts, same seeds, greedy and sampled) produces
        # byte-identical ids and PCM on every rank (batch invariance makes that a requirement, not a hope)
        from auralis_amd.parallel import all_ranks_equal
        multi["bcast_route"] = route
        multi["conditioning_hash_equal_across_ranks"] = all_ranks_equal(eng.conditioning_checksum(SPK))
        for k in range(4):
            eng.submit(B.make_ids(20 + 5 * k, 900 + k), SPK, temperature=(0.0 if k < 2 else 0.75),
                       top_p=0.85, top_k=50, repetition_penalty=5.0, max_tokens=24, seed=4242 + k, ignore_stop=True)
        outs = sorted(eng.run_until_done(), key=lambda o: o["seq_id"])
        hh = hashlib.blake2b(digest_size=16)
        for o in outs:
            hh.update(np.asarray(o["tokens"], np.int32).tobytes())
            hh.update(np.asarray(o["wav"], np.float32).tobytes())
        multi["output_hash_equal_across_ranks"] = all_ranks_equal(hh.hexdigest())

    def side(name, fn):
        try:
            t0 = time.perf_counter()
            r = fn()
            _log(f"{name} done in {time.perf_counter() - t0:.1f} s")
            return r
        except Exception as ex:   # noqa: BLE001 - never lose the headline over a side measurement
            _log(f"{name} FAILED: {type(ex).__name__}: {ex}")
            return {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}

    if args.workload != "c3":   # ad-hoc: one workload, its own record
        if args.warmup:
            B.run_batches(range(-args.warmup, 0), batch=(1 if args.workload == "c2" else None))
        rec = {"c2": B.workload_c2, "c5s": lambda: B.workload_c5s(args.c5_chars), "c4": B.workload_c4,
               "c3f": lambda: B.workload_c3f(max(1, args.steps))}[args.workload]()
        if rank == 0:
            rec["n_gpus"] = world
            print(json.dumps(rec), file=json_out, flush=True)
        eng.close()
        if use_dist:
            torch.distributed.destroy_process_group()
        return

    # ---- headline: c3, timed region with the engine as the parity tests run it
    if args.warmup:
        B.run_batches(range(-args.warmup, 0))
        _log(f"{args.warmup} warmup step(s) done")
    eng.reset_stats()
    B.fence()
    t0 = time.perf_counter()
    samples = B.run_batches(range(args.steps))
    _log(f"{args.steps} timed step(s) done at +{time.perf_counter() - t0:.3f}s")
    B.fence()
    dt = time.perf_counter() - t0
    st = eng.stats()

    if use_dist:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, dt / args.steps * 1e3)
        multi["per_rank_ms_per_step"] = [_r(v, 5) for v in per_rank]
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(s, op=torch.distributed.ReduceOp.SUM)
        samples = float(s.item())

    # ---- profile pass (every rank runs it, so that the ranks stay in step; rank 0 reports): the same batch once more with profile on
    kernels = None
    if not args.no_profile_pass:
        eng.set_profile(args.profile_every)
        eng.reset_stats()
        ps = B.run_batches([args.steps + 1000])
        eng.sync()
        pst = eng.stats()
        eng.set_profile(0)
        kernels = kernel_rooflines(args, pst, dims, ps)
        _log("profile pass done")

    line = None
    if rank == 0:
        audio_s = samples / 24000.0
        est = (kernels or {}).get("est_gpu_ms_in_pass") or {}
        line = {
            "metric": "audio_samples_per_s (64-way batch; rtf = wall_s / audio_s alongside)",
            "value": samples / dt, "unit": "audio-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (GPT; GEMMs " + ("exact-f32 MFMA" if args.gemm == "f32" else "as exact bf16x3 splits, f32 acc") + ")"
                     + ("" if args.vocoder == "fp32" else " + f16-in/f32-acc vocoder") + (" + f16 K/V (NOT the parity mode)" if args.kv == "fp16" else ""),
            "data": "synthetic" if not args.checkpoint else "real checkpoint weights (" + os.path.basename(os.path.normpath(args.checkpoint)) + "), synthetic text / fixed-length generation",
            "rtf": dt / audio_s,
            "config": {"workload": f"BASELINE configs[2]: {args.batch} concurrent 200-char utterances per GPU (70 text tokens -> {args.tokens} mel "
                                   f"tokens fixed-length -> {dims.voc.samples_for_latents(args.tokens)} samples each), T=0.75 top_p=0.85 top_k=50 "
                                   "rep_pen=5.0, shared speaker latent, continuous batching"
                                   + ("; steps pipelined" if args.pipeline else ""),
                       "utterances_per_gpu": args.batch, "mel_tokens": args.tokens, "gpt_layers": args.layers,
                       "vocoder_mfma_inputs": args.vocoder, "kv_cache": args.kv, "gemm_arithmetic": args.gemm, "engine_profile_mode_in_timed_region": False,
                       "parallelism": f"dp{world} (independent utterances, 1 RCCL broadcast of conditioning)"},
            "kernels": kernels,
            "dominant_share": (est.get("gemm_rows_kernel", 0.0) / sum(est.values())) if est else None,
            "decode_step": decode_step_roofline(st),
            "breakdown_ms_per_step": {"gpt": st["gpt_ms"] / args.steps, "gpt_prefill": st["prefill_ms"] / args.steps,
                                      "gpt_decode": st["decode_ms"] / args.steps, "vocoder": st["vocoder_ms"] / args.steps,
                                      "gpt_ms_per_decode_step": st["decode_ms"] / max(1, st["decode_steps"])},
            "audio_sec_per_wall_sec": audio_s / dt,
        }
        if use_dist:
            line["multi_gpu"] = multi
    # ---- side workloads
    c2 = c5 = c4 = c3f = None
    if not args.no_side:
        if world == 1:
            c3f = side("c3f", lambda: B.workload_c3f(max(1, min(args.steps, 10))))
            c2 = side("c2", B.workload_c2)
            c5 = side("c5s", lambda: B.workload_c5s(args.c5_chars))
        else:
            c4 = B.workload_c4()   # collective: no try/except around a path every rank must walk together
    # ---- the in-library RCCL route, after every measurement is in hand: a second voice key is broadcast with aur_comm_init /
    # aur_broadcast_conditioning on the engine's own communicator and must arrive byte-identical to the first one.  Under a
    # watchdog: a hang ends in "timeout" in the line (and a hard exit after it has been printed), not in a lost run.
    hard_exit = False
    if use_dist and route == "torch" and args.bcast == "torch":
        import threading
        from auralis_amd.parallel import broadcast_conditioning_native
        res = {}

        def native():
            try:
                torch.cuda.set_device(local_rank)   # per thread
                broadcast_conditioning_native(eng, SPK + 1, B.cond if rank == 0 else None, B.spk if rank == 0 else None, src=0, device=dev)
                res["ranks"], res["rank0"] = eng.comm_info()
                res["same_bytes"] = eng.conditioning_checksum(SPK + 1) == eng.conditioning_checksum(SPK)
                res["status"] = "ok" if res["same_bytes"] and res["ranks"] == world else "mismatch"
            except Exception as ex:   # noqa: BLE001 - reported in the bench line
                res["status"] = f"failed: {type(ex).__name__}: {str(ex)[:120]}"
        th = threading.Thread(target=native, daemon=True)
        th.start()
        th.join(args.native_check_timeout)
        if th.is_alive():
            res["status"] = "timeout"
            hard_exit = True
        multi["native_route"] = res.get("status")
        multi["native_rccl_ranks"] = res.get("ranks")
        if line is not None:
            line["multi_gpu"] = multi
    if rank == 0:
        line["c2"], line["c5s"], line["c4"], line["c3f"] = c2, c5, c4, c3f
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B.gpt_sd, B.xtts_sd, dims, B.cond, B.spk, B.text_ids, args.cpu_tokens, args.cpu_c1,
                                                activation="gelu" if B.gelu_erf else "gelu_new")
        else:
            line["cpu_baseline"] = None
        full_path = args.out
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(line, f, indent=1)
        except OSError as ex:
            _log(f"could not write {full_path}: {ex}")
            full_path = None
        print("[bench full] " + json.dumps(line), file=sys.stderr, flush=True)
        comp = compact(line, os.path.relpath(full_path, ROOT) if full_path and os.path.isabs(full_path) else full_path)
        txt = json.dumps(comp, separators=(",", ":"))
        _log(f"compact line: {len(txt)} bytes")
        print(txt, file=json_out, flush=True)
    if hard_exit:   # a rank is stuck inside the native collective: nothing below would return
        sys.stderr.flush()
        os._exit(0)
    eng.close()
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
