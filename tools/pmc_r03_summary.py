"""Fold the rocprofv3 PMC passes of the default bench command into profiles/hbm_traffic.json["r03_decode"] / ["r03_conv"].

  PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' tools/pmc.sh r03dec
  PMC_PASSES='fetch write' PMC_KERNELS='conv1d_mfma' tools/pmc.sh r03conv
  python tools/pmc_r03_summary.py gpurun_out/pmc_r03dec 40 gpurun_out/pmc_r03conv

Units and corrections (MI355X_MICROARCH.md §HBM): Counter_Value is KiB per dispatch; FETCH_SIZE reports 1/2 of the bytes of a
wide coalesced streaming read (16 B per lane), other access widths are uncalibrated, WRITE_SIZE is taken as reported.
* decode kernels: every load is a 16-B-per-lane float4 stream -> fetch = 2 x raw.
* conv kernels: three access mixes, each calibrated on the stage-4 (32-channel, one co-tile) launch of its class whose compulsory
  traffic is known exactly (64 utterances x 32 channels x 312 064 samples: fp32 tensor 2.556 GB, fp16 tensor 1.278 GB):
    class A  first convs of ResBlock rounds 1, 2 (<k, d > 1, .., XH = true>): read one fp16 tensor with 16-B loads
    class B  second convs (<k, 1, .., XH = true>): fp16 tensor with 16-B loads + fp32 residual with 4-B loads
    class C  fp32-input convs (XH = false: conv_pre, transposed convs, first conv of round 0): 4-B loads
  every launch's FETCH_SIZE is divided by its class factor (raw / known of the calibration launch)."""
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
    out = {"command": f"PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens {T}' "
                      f"tools/pmc.sh r03dec; python tools/pmc_r03_summary.py ...",
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

    def cls(name):
        m = re.search(r"conv1d_mfma_f16_kernel<(\d+), (\d+), (\d+), (true|false)", name)
        if not m:
            return None
        xh, dil = m.group(4) == "true", int(m.group(2))
        return "A" if (xh and dil > 1) else ("B" if xh else "C")
    t32 = 64 * 32 * 312064 * 4.0
    known = {"A": ("conv1d_mfma_f16_kernel<3, 3, 32, true", t32 / 2), "B": ("conv1d_mfma_f16_kernel<3, 1, 32, true", t32 / 2 + t32),
             "C": ("conv1d_mfma_f16_kernel<3, 1, 32, false", t32)}
    factor, cal = {}, {}
    for c, (prefix, kb) in known.items():
        ks = [k for k in f if k.startswith(prefix)]
        if not ks:
            continue
        v = f[ks[0]]
        # class B's calibration kernel runs 9 times per batch, 3 of them with the MRF accumulator in the epilogue: take the
        # smallest launches (plain residual) for the known-traffic comparison
        raw = sorted(v)[: max(1, len(v) // 3)] if c == "B" else v
        raw_mean = sum(raw) / len(raw)
        factor[c] = raw_mean / kb
        cal[c] = {"kernel": ks[0], "known_read_bytes": kb, "fetch_raw_mean": raw_mean, "fetch_raw_over_known": factor[c]}
    n = tot_f = tot_w = 0.0
    by_class = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, v in f.items():
        c = cls(k)
        if c is None:
            continue
        corr = factor.get(c, 0.5)
        fb = sum(v) / corr
        wb = sum(w.get(k, []))
        n += len(v)
        tot_f += fb
        tot_w += wb
        by_class[c][0] += len(v)
        by_class[c][1] += fb
        by_class[c][2] += wb
    return {"command": "PMC_PASSES='fetch write' PMC_KERNELS='conv1d_mfma' tools/pmc.sh r03conv; python tools/pmc_r03_summary.py ...",
            "conv_fp16": {"launches": int(n), "fetch_bytes_per_launch": tot_f / max(1, n), "write_bytes_per_launch": tot_w / max(1, n),
                          "bytes_per_launch": (tot_f + tot_w) / max(1, n)},
            "calibration": cal,
            "by_class": {c: {"launches": v[0], "fetch_bytes_per_launch": v[1] / max(1, v[0]), "write_bytes_per_launch": v[2] / max(1, v[0])}
                         for c, v in by_class.items()},
            "note": "FETCH_SIZE divided by the class factor measured on a launch of known compulsory traffic (module docstring); "
                    "WRITE_SIZE as reported (exact for 4-B/lane fp32 stores, over-counts 2-B/lane fp16 stores by ~1.19x in round 1)"}


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    j = json.load(open(path))
    if len(sys.argv) > 2 and os.path.isdir(sys.argv[1]):
        j["r03_decode"] = decode_summary(sys.argv[1], int(sys.argv[2]))
        print(json.dumps({k: (v if not isinstance(v, dict) else {"ratio": v.get("ratio_to_algorithmic")}) for k, v in j["r03_decode"].items()}, indent=0))
    if len(sys.argv) > 3 and os.path.isdir(sys.argv[3]):
        j["r03_conv"] = conv_summary(sys.argv[3])
        print(json.dumps({k: v for k, v in j["r03_conv"].items() if k != "note"}, indent=0))
    json.dump(j, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
