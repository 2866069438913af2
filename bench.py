#!/usr/bin/env python3
"""bench.py — realtime-factor / audio-samples-per-second of the generate_speech() hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: BASELINE.json configs[2]
(64 concurrent 200-char utterances, shared speaker latent, temperature 0.75 / top_p 0.85 / top_k 50 /
repetition penalty 5.0) per GPU: 70 text tokens -> prefill of 103 rows -> 280 mel tokens (fixed-length
mode) -> latent stash -> HiFi-GAN -> 312 064 samples per utterance, waveforms copied to host.
Weights (seeded synthetic, true shapes) and speaker conditioning are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Multi-GPU: utterances are independent, so each rank runs its own 64-way batch (weak scaling); the only
collective is one RCCL broadcast of the speaker conditioning (133 120 B) from rank 0 before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # exact-f32 MFMA / vector peak


def _log(msg: str):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(gpt_sd, xtts_sd, dims, cond, spk, text_ids, n_tokens: int, budget_s: float = 30.0):
    """Time the CPU oracle (kind 'port') on a bounded sample of the same workload: ONE utterance, greedy,
    up to n_tokens mel tokens (prefill + decode + literal second pass + vocoder).  The decode loop stops early
    when `budget_s` is used up, so the leg is bounded whatever the host looks like."""
    from oracle import xtts_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(cores, 64)))
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    w = O.vocoder_effective_weights(xtts_sd)
    c = gpt.build_cond(cond, text_ids)
    t0 = time.perf_counter()
    out = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=8, ignore_stop=True))
    t_probe = time.perf_counter() - t0
    # scale the sample so that prefill + decode + second pass + vocoder stay near the budget
    per_tok = max(1e-3, t_probe / 8.0)
    n = int(max(8, min(n_tokens, budget_s / (3.0 * per_tok))))
    _log(f"cpu_baseline: probe 8 tokens in {t_probe:.2f}s on {torch.get_num_threads()} threads -> sample {n} tokens")
    t0 = time.perf_counter()
    out = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=n, ignore_stop=True))
    t1 = time.perf_counter()
    lat = gpt.second_pass_latents(c, out["tokens"])
    t2 = time.perf_counter()
    wav = O.hifi_decoder_forward(w, lat, spk)
    t3 = time.perf_counter()
    ns = wav.numel()
    return {
        "value": ns / (t3 - t0), "unit": "audio-samples/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"1 utterance, 70 text tokens, {n} mel tokens greedy -> {ns} samples: prefill+decode "
                  f"{t1 - t0:.2f}s, second pass {t2 - t1:.2f}s, vocoder {t3 - t2:.2f}s (torch CPU fp32 oracle)",
        "rtf": (t3 - t0) / (ns / 24000.0),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--tokens", type=int, default=280, help="mel tokens per utterance (fixed-length mode)")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--vocoder", choices=["fp32", "fp16"], default="fp16",
                    help="MFMA input type of the HiFi-GAN convs (fp32 accumulate either way; GPT is fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="queue all steps at once so the vocoder of batch k overlaps the GPT of batch k+1 (measured "
                         "neutral on MI355X in fp32: both stages want the same CUs)")
    ap.add_argument("--cpu-tokens", type=int, default=64)
    args = ap.parse_args()

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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_text_ids,
                                        make_synthetic_xtts)
    from auralis_amd.config import XTTSDims
    from auralis_amd.parallel import broadcast_conditioning
    from auralis_amd.weights import pack_all

    dims = XTTSDims()
    _log("building synthetic checkpoint")
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=args.layers)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    eng = NativeEngine(n_layer=args.layers, max_seqs=args.batch, device=local_rank, profile=True,
                       vocoder_fp16=(args.vocoder == "fp16"), return_latents=False)   # audio + tokens, as TTSOutput
    eng.load_weights(pack_all(gpt_sd, xtts_sd))
    _log("weights resident")

    # speaker conditioning: computed on rank 0, RCCL-broadcast over xGMI, registered from the device buffer
    cond, spk = make_synthetic_conditioning(dims)
    SPK = 1
    if use_dist:
        broadcast_conditioning(eng, SPK, cond if rank == 0 else None, spk if rank == 0 else None, src=0,
                               device=torch.device("cuda", local_rank))
    else:
        eng.set_conditioning(SPK, cond.numpy(), spk.numpy())
    text_ids = make_synthetic_text_ids(dims, n_text=70, seed=11)

    def run_steps(first: int, n: int):
        """n steps = n batches of `batch` utterances, one after the other.  With --pipeline all n batches are queued at
        once: the continuous batcher admits batch k+1 into the slots batch k frees when its tokens are done, so the
        HiFi-GAN of batch k (own stream) overlaps the GPT prefill/decode of batch k+1."""
        total = 0
        groups = [range(first, first + n)] if args.pipeline else [[k] for k in range(first, first + n)]
        for grp in groups:
            for k in grp:
                for b in range(args.batch):
                    eng.submit(text_ids, SPK, temperature=0.75, top_p=0.85, top_k=50, repetition_penalty=5.0,
                               max_tokens=args.tokens, seed=(rank * 100003 + (k + 7) * 1009 + b), ignore_stop=True)
            outs = eng.run_until_done(max_steps=len(grp) * (args.tokens + 16) + 64)
            assert len(outs) == args.batch * len(grp)
            total += sum(len(o["wav"]) for o in outs)
        return total

    def fence():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        eng.sync()

    if args.warmup:
        run_steps(-args.warmup, args.warmup)
        _log(f"{args.warmup} warmup step(s) done")
    eng.reset_stats()
    fence()
    t0 = time.perf_counter()
    samples = run_steps(0, args.steps)
    _log(f"{args.steps} timed step(s) done at +{time.perf_counter() - t0:.3f}s")
    fence()
    dt = time.perf_counter() - t0
    st = eng.stats()

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(s, op=torch.distributed.ReduceOp.SUM)
        samples = float(s.item())

    if rank == 0:
        audio_s = samples / 24000.0
        conv_s = st["conv_ms"] * 1e-3
        n_conv = max(1, st["conv_launches"])
        # PMC-measured HBM bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, profiles/hbm_traffic.json; the
        # counters cannot be read from inside this process)
        traffic = gemm_traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"conv_{args.vocoder}_bytes_per_launch")
                if args.batch == 64 and args.layers == 30:
                    gemm_traffic = tj.get("decode_gemm", {}).get("bytes_per_launch")
            except Exception:
                traffic = gemm_traffic = None
        conv_kernel = "conv1d_mfma_f16_kernel" if args.vocoder == "fp16" else "conv1d_mfma_kernel"
        conv_gbps = st["conv_bytes"] / conv_s / 1e9 if conv_s > 0 else 0.0
        conv_tflops = st["conv_flops"] / conv_s / 1e12 if conv_s > 0 else 0.0
        roof_conv = {
            "kernel": f"{conv_kernel} (HiFi-GAN convs, all instantiations)",
            "bound": "hbm", "achieved": conv_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": conv_gbps / HBM_PEAK_GBPS,
            "traffic": traffic, "avg_launch_ms": st["conv_ms"] / n_conv, "launches": st["conv_launches"],
            "algorithmic_bytes_per_launch": st["conv_bytes"] / n_conv, "total_ms_in_timed_region": st["conv_ms"],
            "mfma": {"achieved": conv_tflops, "unit": "TFLOP/s",
                     "peak": FP32_MFMA_PEAK_TFLOPS if args.vocoder == "fp32" else 2500.0,
                     "frac": conv_tflops / (FP32_MFMA_PEAK_TFLOPS if args.vocoder == "fp32" else 2500.0)},
        }
        # decode GEMMs: HIP-event pairs around every launch of each 16th decode step (sampled)
        # `achieved` uses the RAW event intervals (kernel + dispatch latency exposed by the event records: conservative);
        # the empty-event-pair overhead and the overhead-corrected average are reported next to it, and the rocprofv3
        # per-kernel average (pure execution time) is in profiles/.
        gemm_s = st["gemm_ms_raw"] * 1e-3
        n_gemm = max(1, st["gemm_launches"])
        gemm_gbps = st["gemm_bytes"] / gemm_s / 1e9 if gemm_s > 0 else 0.0
        gemm_tflops = st["gemm_flops"] / gemm_s / 1e12 if gemm_s > 0 else 0.0
        est_gemm_total_ms = (st["gemm_ms"] / n_gemm) * 121.0 * max(0, st["steps"] - args.steps)   # overhead-corrected
        roof_gemm = {
            "kernel": "gemm_splitk_kernel<false> (decode QKV / proj / FC / proj2 / mel-head, M = live sequences)",
            "bound": "hbm", "achieved": gemm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gemm_gbps / HBM_PEAK_GBPS,
            "traffic": gemm_traffic, "avg_launch_ms": st["gemm_ms_raw"] / n_gemm, "avg_launch_ms_minus_event_overhead": st["gemm_ms"] / n_gemm,
            "event_pair_overhead_ms": st["event_pair_overhead_ms"], "launches_sampled": st["gemm_launches"],
            "algorithmic_bytes_per_launch": st["gemm_bytes"] / n_gemm, "total_ms_in_timed_region_est": est_gemm_total_ms,
            "mfma": {"achieved": gemm_tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / FP32_MFMA_PEAK_TFLOPS},
            "note": "weights stream once per step (HBM) but at M = 64 exact-f32 MFMA time is of the same order; per-launch "
                    "latency dominates (one 32x64x256 tile per workgroup, <= 2 workgroups per CU); traffic = weights once + "
                    "the activation matrix once per XCD L2 (PMC, profiles/hbm_traffic.json)",
        }
        dominant, other = (roof_gemm, roof_conv) if est_gemm_total_ms > st["conv_ms"] else (roof_conv, roof_gemm)
        line = {
            "metric": "audio_samples_per_s (64-way batch; rtf = wall_s / audio_s alongside)",
            "value": samples / dt, "unit": "audio-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.vocoder == "fp32" else "f32 (GPT) + f16-in/f32-acc MFMA (vocoder convs)", "data": "synthetic",
            "rtf": dt / audio_s, "audio_sec_per_wall_sec": audio_s / dt,
            "config": {"workload": f"{args.batch} concurrent 200-char utterances per GPU (70 text tokens -> "
                                   f"{args.tokens} mel tokens fixed-length -> {dims.voc.samples_for_latents(args.tokens)} "
                                   f"samples each), T=0.75 top_p=0.85 top_k=50 rep_pen=5.0, shared speaker latent, "
                                   f"continuous batching; BASELINE.json configs[2]"
                                   + ("; consecutive steps pipelined (vocoder of batch k overlaps GPT of batch k+1)" if args.pipeline else ""),
                       "utterances_per_gpu": args.batch, "mel_tokens": args.tokens, "gpt_layers": args.layers,
                       "vocoder_mfma_inputs": args.vocoder,
                       "parallelism": f"dp{world} (independent utterances, 1 RCCL broadcast of conditioning)"},
            "roofline": dominant,
            "roofline_second_kernel": other,
            "breakdown_ms_per_step": {"gpt": st["gpt_ms"] / args.steps, "vocoder": st["vocoder_ms"] / args.steps,
                                      "vocoder_convs": st["conv_ms"] / args.steps,
                                      "gpt_ms_per_decode_step": st["gpt_ms"] / max(1, st["steps"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(gpt_sd, xtts_sd, dims, cond, spk, text_ids, args.cpu_tokens)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    eng.close()
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
