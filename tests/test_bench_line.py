"""bench.py's driver-facing line: the compact record must stay parseable (< 4 KB, one JSON object) and carry the contract fields,
`roofline` and `cpu_baseline` (a 20 KB line defeated the driver's parser).  CPU only: the engine statistics are faked."""
import argparse
import json

import bench


def _stats():
    L = 30
    st = {k: 0 for k in ("steps", "prefill_rows", "decode_rows", "tokens_generated", "samples_generated", "vocoder_batches", "conv_launches",
                         "gemm_launches", "attn_launches", "decode_steps", "kv_blocks_total", "kv_blocks_free")}
    st.update({k: 0.0 for k in ("conv_ms", "conv_flops", "conv_bytes", "gemm_ms", "gemm_ms_raw", "event_pair_overhead_ms", "gemm_flops", "gemm_bytes",
                                "vocoder_ms", "gpt_ms", "attn_ms", "attn_bytes", "decode_ms", "prefill_ms", "decode_weight_bytes", "decode_kv_bytes")})
    n_b = 35   # profiled steps
    st["gemm_kind_launches"] = [n_b * L] * 5
    st["gemm_kind_ms"] = [n_b * L * us * 1e-3 for us in (8.4, 5.5, 9.2, 12.1, 4.9)]
    shp = [(64, 3072, 1024, 1.0), (64, 1024, 1024, 2.0), (64, 4096, 1024, 1.0), (64, 1024, 4096, 2.0), (64, 1088, 1024, 1.0)]
    st["gemm_kind_bytes"] = [n_b * L * 4.0 * (k * n + m * k + m * n * o) for m, n, k, o in shp]
    st["gemm_kind_flops"] = [n_b * L * 2.0 * m * n * k for m, n, k, _ in shp]
    st["attn_launches"], st["attn_ms"], st["attn_bytes"] = n_b * L, n_b * L * 22.9e-3, n_b * L * 127.7e6
    st["decode_steps"], st["decode_ms"], st["decode_rows"] = 279, 279 * 1.83, 279 * 64
    st["decode_weight_bytes"], st["decode_kv_bytes"] = 279 * 1.514e9, 279 * 3.83e9
    st["prefill_ms"], st["prefill_rows"], st["vocoder_batches"] = 31.6, 4544, 1
    st["conv_launches"], st["conv_ms"], st["conv_flops"], st["conv_bytes"] = 59, 66.2, 48.4e12, 64 * 312064 * 6700.0
    st["conv_class_launches"] = [18, 18, 9, 9, 5]
    st["conv_class_ms"] = [10.9, 25.1, 15.5, 9.6, 4.0]
    st["conv_class_bytes"] = [1e10, 4e10, 4e10, 3e10, 1e10]
    st["conv_class_flops"] = [1e13, 2e13, 1e13, 0.5e13, 0.3e13]
    st["vocoder_ms"], st["gpt_ms"], st["event_pair_overhead_ms"] = 67.8, 540.0, 0.0046
    return st


def test_compact_line_is_small_and_complete():
    args = argparse.Namespace(gemm="bf16x3", vocoder="fp16", layers=30, batch=64, tokens=280, kv="fp32")

    class D:   # dims stand-in (kernel_rooflines does not touch it)
        pass
    st = _stats()
    k = bench.kernel_rooflines(args, st, D(), 64 * 312064)
    fam = k["gemm_family"]
    assert abs(fam["four_gemms_per_layer_us"] - 35.2) < 1e-6
    # the family fraction is reproducible by hand: algorithmic bytes of the four block GEMMs per step / their time per step / 8 TB/s
    # (the mel head's L2-warm replay is reported per kind only)
    by = sum(st["gemm_kind_bytes"][i] / st["gemm_kind_launches"][i] * 30 for i in range(4))
    us = 30 * 35.2
    assert abs(fam["frac"] - by / (us * 1e-6) / 8e12) < 1e-9
    ds = bench.decode_step_roofline(st)
    line = {"metric": "audio_samples_per_s (64-way batch; rtf = wall_s / audio_s alongside)", "value": 32345678.9, "unit": "audio-samples/s",
            "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 617.123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (GPT; GEMMs as exact bf16x3 splits, f32 acc) + f16-in/f32-acc vocoder", "data": "synthetic", "rtf": 0.000741234,
            "config": {"workload": "BASELINE configs[2]: 64 concurrent 200-char utterances per GPU (70 text tokens -> 280 mel tokens fixed-length "
                                   "-> 312064 samples each), T=0.75 top_p=0.85 top_k=50 rep_pen=5.0, shared speaker latent, continuous batching",
                       "utterances_per_gpu": 64, "mel_tokens": 280, "gpt_layers": 30, "vocoder_mfma_inputs": "fp16", "kv_cache": "fp32",
                       "gemm_arithmetic": "bf16x3", "engine_profile_mode_in_timed_region": False,
                       "parallelism": "dp8 (independent utterances, 1 RCCL broadcast of conditioning)"},
            "kernels": k, "dominant_share": 0.449, "decode_step": ds,
            "breakdown_ms_per_step": {"gpt": 540.3, "gpt_prefill": 31.8, "gpt_decode": 508.5, "vocoder": 67.8, "gpt_ms_per_decode_step": 1.8231},
            "c2": {"time_to_audio_ms": 312.3, "rtf": 0.024, "decode_step": {"ms": 0.9, "frac_as_stored": 0.21}, "prefill_ms": 3.1, "vocoder_ms": 4.2},
            "c5s": {"chars": 20123, "chunks": 101, "samples_per_s": 2.1e7, "rtf": 0.0011, "slot_occupancy": 0.71, "first_chunk_s": 0.41, "in_order": True},
            "c4": {"workload": "x", "units": 512, "units_returned_in_order": True, "wall_s": 0.7, "samples": 1.6e8, "samples_per_s": 2.3e8, "rtf": 0.0001,
                   "scaling": "strong", "ids_digest": "0123456789abcdef"},
            "cpu_baseline": {"value": 13350.1, "unit": "audio-samples/s", "cores": 16, "host_cores": 256, "kind": "port", "rtf": 1.7977,
                             "sample": "C2 in full: 1 utterance, 70 text ids, 280 greedy mel tokens -> 312064 samples in 23.4 s (AR 20.5, second pass 0.3, "
                                       "HiFi-GAN 2.6); torch CPU fp32 oracle, 16 threads", "c2": {}, "thread_sweep_s_per_token": {}},
            "multi_gpu": {"rccl_ranks": 8, "rccl_rank_of_rank0": 0, "native_route_failed": False, "bcast_route": "native",
                          "conditioning_hash_equal_across_ranks": True, "output_hash_equal_across_ranks": True,
                          "per_rank_ms_per_step": [617.12] * 8}}
    txt = json.dumps(bench.compact(line, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(txt) < 4096, len(txt)
    assert "\n" not in txt
    back = json.loads(txt)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in back["roofline"], key
    assert back["roofline"]["bound"] == "hbm" and abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / 8000.0) < 2e-3
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert "workload" in back["config"] and "model" not in back["config"]
