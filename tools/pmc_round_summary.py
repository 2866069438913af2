"""Fold the rocprofv3 PMC passes of the default bench command into profiles/hbm_traffic.json["<round>_decode"] / ["<round>_conv"]
(round tag from PMC_ROUND, default r04; bench.py reads the newest keys it finds).

  PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' tools/pmc.sh r03dec
  PMC_PASSES='fetch write' PMC_KERNELS='conv1d_|resblock_round' tools/pmc.sh r03conv
  python tools/pmc_round_summary.py gpurun_out/pmc_r03dec 40 gpurun_out/pmc_r03conv

Units and corrections (MI355X_MICROARCH.md §HBM): Counter_Value is KiB per dispatch; FETCH_SIZE reports 1/2 of the bytes of a
wide coalesced streaming read (16 B per lane), other access widths are uncalibrated, WRITE_SIZE is taken as reported.
* decode kernels: every load is a 16-B-per-lane float4 stream -> fetch = 2 x raw.
* conv kernels: two input families, each calibrated on launches whose compulsory read traffic is known exactly
    fp16 interleaved inputs (ResBlock convs, fused rounds, transposed convs: LDS-DMA copies or 16-B loads + 8-B loads): the 9 fused
      round launches of the 32-channel stage, 11 fp16 tensors of 64 x 32 x 312 064 halves
    fp32 inputs (conv_pre, the four polyphase transposed convs: 4-B loads): the transposed convs' input tensors
  every launch's FETCH_SIZE is divided by its family's factor (raw / known)."""
import collections
import csv
import json
import os
import re
import sys

KIB = 1024.0
DECODE = [("attention", r"paged_attention_kernel<", None),
          ("gemm_qkv", r"gemm_rows_kernel<\d+, 1, true, 3,", 4.0 * (1024 * 3072 + 64 * 1024 + 64 * 3072)),
          ("gemm_proj", r"gemm_rows_kernel<\d+, 1, false, 2,", 4.0 * (1024 * 1024 + 64 * 1024 + 2 * 64 * 1024)),
          ("gemm_fc", r"gemm_rows_kernel<\d+, 1, true, 1,", 4.0 * (1024 * 4096 + 64 * 1024 + 64 * 4096)),
          ("gemm_proj2", r"gemm_rows_kernel<\d+, 4, false, 2,", 4.0 * (4096 * 1024 + 64 * 4096 + 2 * 64 * 1024)),
          ("gemm_head", r"gemm_rows_kernel<\d+, 1, false, 0,", 4.0 * (1024 * 1088 + 64 * 1024 + 64 * 1088))]


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def decode_summary(d, T):
    def load(p):
        g = collections.defaultdict(list)
        for r in rows(p):
            for key, pat, _ in DECODE:
                if re.search(pat, r["Kernel_Name"]):
                    g[key].append(float(r["Counter_Value"]) * KIB)
                    break
        return g
    f, w = load(os.path.join(d, "fetch", "pmc_counter_collection.csv")), load(os.path.join(d, "write", "pmc_counter_collection.csv"))
    out = {"tokens": T,
           "command": f"PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens {T}' "
                      f"tools/pmc.sh <tag>; python tools/pmc_round_summary.py ...",
           "fetch_correction": "x2 (16 B per lane streaming reads, MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported"}
    for key, _, alg in DECODE:
        if key not in f:
            continue
        fr, wr = f[key], w.get(key, [])
        fetch, write = 2.0 * sum(fr) / len(fr), (sum(wr) / len(wr) if wr else 0.0)
        if alg is None:   # attention: decode steps 1..T-1, 64 sequences, context = 103 prompt rows + step (+ the new token)
            ctx = sum(103 + s + 1 for s in range(1, T)) / max(1, T - 1)
            alg = 64.0 * ctx * 8192.0 + 2 * 64 * 1024 * 4.0
        out[key] = {"launches": len(fr), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch_in_that_run": alg,
                    "ratio_to_algorithmic": (fetch + write) / alg}
    return out


def conv_summary(d):
    def load(p, counter):
        g = collections.defaultdict(list)
        for r in rows(p):
            if r["Counter_Name"] == counter:
                g[r["Kernel_Name"].replace("void aur::", "").replace("(aur::ConvArgs)", "")].append(float(r["Counter_Value"]) * KIB)
        return g
    f = load(os.path.join(d, "fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    w = load(os.path.join(d, "write", "pmc_counter_collection.csv"), "WRITE_SIZE")

    def family(name):   # "h": fp16 interleaved inputs (16-B loads / LDS-DMA copies, 8-B residual loads); "f": fp32 inputs (4-B loads)
        if name.startswith("conv1d_dma_f16_kernel<") or name.startswith("resblock_round_f16_kernel<"):
            return "h"
        m = re.search(r"conv1d_mfma_f16_kernel<(\d+), (\d+), (\d+), (true|false)", name)
        return None if not m else ("h" if m.group(4) == "true" else "f")
    half = 64 * 32 * 312064 * 2.0   # one fp16 tensor of the 32-channel stage, 64 utterances at full length: 1.278 GB
    # known compulsory reads (one co-tile per launch, so nothing is read twice):
    #  h: the 9 fused ResBlock-round launches of the 32-channel stage: each reads the stream once (1 tensor), the last rounds of
    #     ResBlocks 1 and 2 also the running sum (1 each) = 11 tensors
    #  f: the four transposed convs read their fp32 input once per 64-row co-tile group; compulsory = the input tensors
    known_h = 11 * half
    known_f = 64 * 4.0 * (512 * 1219 + 256 * 9752 + 128 * 78016 + 64 * 156032)
    raw_h = sum(sum(v) for k, v in f.items() if re.match(r"resblock_round_f16_kernel<\d+, \d+, 32", k))
    raw_f = sum(sum(v) for k, v in f.items() if re.match(r"conv1d_mfma_f16_kernel<2, 1, \d+, false", k))
    # (since the stage inputs are halves too, only conv_pre reads fp32: no calibration launches, FETCH_SIZE taken as reported)
    factor = {"h": raw_h / known_h if raw_h else 0.5, "f": raw_f / known_f if raw_f else 1.0}
    cal = {"fp16_inputs": {"launches": "resblock_round_f16_kernel<*, *, 32, *> (the 32-channel stage)", "known_read_bytes": known_h,
                           "fetch_raw": raw_h, "fetch_raw_over_known": factor["h"]},
           "fp32_inputs": {"launches": "conv1d_mfma_f16_kernel<2, 1, *, false> (the four transposed convs)", "known_read_bytes": known_f,
                           "fetch_raw": raw_f, "fetch_raw_over_known": factor["f"]}}
    n = tot_f = tot_w = 0.0
    by_kernel = {}
    for k, v in f.items():
        c = family(k)
        if c is None:
            continue
        fb = sum(v) / factor[c]
        wb = sum(w.get(k, []))
        n += len(v)
        tot_f += fb
        tot_w += wb
        by_kernel[k] = {"launches": len(v), "fetch_bytes_per_launch": fb / len(v), "write_bytes_per_launch": wb / len(v)}
    return {"command": "PMC_PASSES='fetch write' PMC_KERNELS='conv1d_|resblock_round' tools/pmc.sh r03conv; python tools/pmc_r03_summary.py ...",
            "conv_fp16": {"launches": int(n), "fetch_bytes_per_launch": tot_f / max(1, n), "write_bytes_per_launch": tot_w / max(1, n),
                          "bytes_per_launch": (tot_f + tot_w) / max(1, n)},
            "calibration": cal, "by_kernel": by_kernel,
            "note": "FETCH_SIZE divided by the factor measured on launches of known compulsory traffic, per input family (module "
                    "docstring); WRITE_SIZE as reported"}


RND = os.environ.get("PMC_ROUND", "r04")


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    j = json.load(open(path))
    if len(sys.argv) > 2 and os.path.isdir(sys.argv[1]):
        j[RND + "_decode"] = decode_summary(sys.argv[1], int(sys.argv[2]))
        print(json.dumps({k: (v if not isinstance(v, dict) else {"ratio": v.get("ratio_to_algorithmic")}) for k, v in j[RND + "_decode"].items()}, indent=0))
    if len(sys.argv) > 3 and os.path.isdir(sys.argv[3]):
        j[RND + "_conv"] = conv_summary(sys.argv[3])
        print(json.dumps({k: v for k, v in j[RND + "_conv"].items() if k != "note"}, indent=0))
    json.dump(j, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
