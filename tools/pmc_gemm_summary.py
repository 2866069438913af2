"""Fold the FETCH_SIZE / WRITE_SIZE passes of the decode GEMM (tools/pmc.sh with PMC_KERNELS='gemm_splitk_kernel<false',
PMC_PASSES='fetch write') into profiles/hbm_traffic.json["decode_gemm"].

The weight stream is the calibrated case of the microarch guide (16 B/lane, FETCH_SIZE reports 1/2): fetch = 2 x raw.
usage: python tools/pmc_gemm_summary.py gpurun_out/pmc_gemm"""
import collections
import csv
import json
import os
import sys

SHAPES = {98304: ("qkv  [1024x3072]", 1024 * 3072 * 4, 64 * 1024 * 4), 32768: ("proj [1024x1024]", 1024 * 1024 * 4, 64 * 1024 * 4),
          131072: ("fc [1024x4096] and proj2 [4096x1024] (same grid)", 4096 * 1024 * 4, (64 * 1024 * 4 + 64 * 4096 * 4) // 2),
          34816: ("mel head [1024x1088]", 1024 * 1088 * 4, 64 * 1024 * 4)}


def load(path):
    g = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            g[int(r["Grid_Size"])].append(float(r["Counter_Value"]) * 1024.0)
    return g


def main():
    d = sys.argv[1]
    f = load(os.path.join(d, "fetch", "pmc_counter_collection.csv"))
    w = load(os.path.join(d, "write", "pmc_counter_collection.csv"))
    by = {}
    tf = tw = n = 0.0
    for gs, (name, wbytes, abytes) in SHAPES.items():
        if gs not in f:
            continue
        fr, wr = f[gs], w.get(gs, [])
        by[name] = {"launches": len(fr), "fetch_bytes (2 x raw)": 2 * sum(fr) / len(fr), "write_bytes": sum(wr) / max(1, len(wr)),
                    "weight_bytes": wbytes, "activation_bytes": abytes,
                    "fetch_minus_weights_over_activations": (2 * sum(fr) / len(fr) - wbytes) / abytes}
        tf += 2 * sum(fr)
        tw += sum(wr)
        n += len(fr)
    out = {"command": "PMC_PASSES='fetch write' PMC_KERNELS='gemm_splitk_kernel<false' tools/pmc.sh gemm (M = 64 decode launches of one "
                      "bench step), summarised by tools/pmc_gemm_summary.py",
           "launches": int(n), "fetch_bytes_per_launch": tf / n, "write_bytes_per_launch": tw / n,
           "bytes_per_launch": (tf + tw) / n,
           "reading": "weights leave HBM exactly once per launch (the two M-tile workgroups of a weight tile share an XCD's L2); the "
                      "excess over the weight bytes equals ~8 x the activation matrix: each of the 8 XCD L2s fetches its own copy "
                      "of the rows the previous kernel wrote.  The mel head (17 column tiles, not a multiple of 8) falls back to the "
                      "plain tile order and reads its weights twice.",
           "by_shape": by}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    j = json.load(open(path))
    j["decode_gemm"] = out
    json.dump(j, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
