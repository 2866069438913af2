"""Fold the FETCH_SIZE / WRITE_SIZE passes of the decode kernels (paged attention + the five gemm_rows GEMMs) into
profiles/hbm_traffic.json["r02_decode"].

  PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens 40' tools/pmc.sh dec
  python tools/pmc_decode_summary.py gpurun_out/pmc_dec 40

FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide coalesced streaming read (16 B per lane; MI355X_MICROARCH.md §HBM):
every load of these kernels is a 16-B-per-lane float4 stream, so fetch = 2 x raw.  WRITE_SIZE is taken as reported.
Counter_Value is in KiB.  The attention launch's bytes grow with the context, so its ALGORITHMIC bytes are computed for
the same launches (64 sequences, prompt 103 rows, decode steps 1 .. T-1 of the PMC run) and the ratio is what transfers to
other context lengths."""
import collections
import csv
import json
import os
import re
import sys

KERNELS = [("attention", r"paged_attention_kernel<false(>|, )", None),
           ("gemm_qkv", r"gemm_rows_kernel<\d, 1, true, 3", 4.0 * (1024 * 3072 + 64 * 1024 + 64 * 3072)),
           ("gemm_proj", r"gemm_rows_kernel<\d, 1, false, 2", 4.0 * (1024 * 1024 + 64 * 1024 + 2 * 64 * 1024)),
           ("gemm_fc", r"gemm_rows_kernel<\d, 1, true, 1", 4.0 * (1024 * 4096 + 64 * 1024 + 64 * 4096)),
           ("gemm_proj2", r"gemm_rows_kernel<\d, 4, false, 2", 4.0 * (4096 * 1024 + 64 * 4096 + 2 * 64 * 1024)),
           ("gemm_head", r"gemm_rows_kernel<\d, 1, false, 0", 4.0 * (1024 * 1088 + 64 * 1024 + 64 * 1088))]


def load(path):
    g = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            for key, pat, _ in KERNELS:
                if re.search(pat, r["Kernel_Name"]):
                    g[key].append(float(r["Counter_Value"]) * 1024.0)
                    break
    return g


def main():
    d, T = sys.argv[1], int(sys.argv[2])
    f = load(os.path.join(d, "fetch", "pmc_counter_collection.csv"))
    w = load(os.path.join(d, "write", "pmc_counter_collection.csv"))
    out = {"command": f"PMC_PASSES='fetch write' PMC_KERNELS='paged_attention_kernel|gemm_rows_kernel' PMC_BENCH_ARGS='--tokens {T}' "
                      f"tools/pmc.sh dec; python tools/pmc_decode_summary.py gpurun_out/pmc_dec {T}",
           "fetch_correction": "x2 (16 B per lane streaming reads, MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported"}
    for key, _, alg in KERNELS:
        if key not in f:
            continue
        fr, wr = f[key], w.get(key, [])
        fetch, write = 2.0 * sum(fr) / len(fr), (sum(wr) / len(wr) if wr else 0.0)
        if alg is None:   # attention: decode steps 1..T-1, 64 sequences, context = 103 prompt rows + step (+ the new token)
            n_steps = max(1, T - 1)
            ctx = sum(103 + s + 1 for s in range(1, T)) / n_steps
            alg = 64.0 * ctx * 8192.0 + 2 * 64 * 1024 * 4.0
        out[key] = {"launches": len(fr), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch_in_that_run": alg,
                    "ratio_to_algorithmic": (fetch + write) / alg}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    j = json.load(open(path))
    j["r02_decode"] = out
    json.dump(j, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
