"""Summarise the rocprofv3 PMC passes of tools/pmc.sh into profiles/hbm_traffic.json.

usage: python tools/pmc_summary.py gpurun_out/pmc_<tag> <round-tag> [fp16|fp32]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  The microarch guide warns that FETCH_SIZE is only
calibrated for 16-B/lane streams (where it reads 1/2); other access patterns must be calibrated on a known byte count.
This file does that on two instantiations whose compulsory traffic is known exactly (stage-4 ResBlock convs, one
co-tile, so no input re-reads): the first conv of a pair (fp32 input, 4 B/lane loads) and the second conv (fp16
input with 2 B/lane loads + fp32 residual in the epilogue)."""
import csv
import json
import os
import sys
from collections import defaultdict

KIB = 1024.0


def load(path):
    per = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [values per dispatch]
    order = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("void aur::", "").replace("(aur::ConvArgs)", "")
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            order.append((int(r["Dispatch_Id"]), k))
    return per, order


def main():
    d, tag = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "fp16"
    fetch, _ = load(os.path.join(d, "fetch", "pmc_counter_collection.csv"))
    write, _ = load(os.path.join(d, "write", "pmc_counter_collection.csv"))
    util, _ = load(os.path.join(d, "util", "pmc_counter_collection.csv"))
    n = sum(len(v["FETCH_SIZE"]) for v in fetch.values())
    fb = sum(sum(v["FETCH_SIZE"]) for v in fetch.values()) * KIB
    wb = sum(sum(v["WRITE_SIZE"]) for v in write.values()) * KIB
    per_kernel = {}
    for k in sorted(fetch):
        per_kernel[k] = {"launches": len(fetch[k]["FETCH_SIZE"]),
                         "fetch_bytes": [x * KIB for x in fetch[k]["FETCH_SIZE"]],
                         "write_bytes": [x * KIB for x in write.get(k, {}).get("WRITE_SIZE", [])]}
    # known-traffic calibration (64 utterances x 32 channels x 312064 samples; fp32 tensor = 2.556 GB, fp16 = 1.278 GB)
    t32 = 64 * 32 * 312064 * 4.0
    cal = {}
    for name, known_r, known_w in (("conv1d_mfma_f16_kernel<3, 3, 32, false>", t32, t32 / 2),
                                   ("conv1d_mfma_f16_kernel<3, 1, 32, true>", t32 / 2 + t32, t32)):
        if name in per_kernel:
            fr = per_kernel[name]["fetch_bytes"]
            wr = per_kernel[name]["write_bytes"]
            cal[name] = {"known_read_bytes": known_r, "known_write_bytes": known_w,
                         "fetch_raw_mean": sum(fr) / len(fr), "write_raw_mean": (sum(wr) / len(wr)) if wr else None,
                         "fetch_raw_over_known": sum(fr) / len(fr) / known_r,
                         "write_raw_over_known": (sum(wr) / len(wr) / known_w) if wr else None}
    # FETCH_SIZE correction: calibrated on the conv whose compulsory read is known exactly and whose loads are the
    # dominant pattern of the family (fp32 rows, 4 B/lane, 1 KiB per wave instruction); WRITE_SIZE is exact for the
    # 4-B/lane fp32 stores (ratio 1.000 below) and is used raw.
    c1 = cal.get("conv1d_mfma_f16_kernel<3, 3, 32, false>")
    fcorr = c1["fetch_raw_over_known"] if c1 else 1.0
    mf = sum(sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) for v in util.values())
    gui = sum(sum(v.get("GRBM_GUI_ACTIVE", [])) for v in util.values())
    wc = sum(sum(v.get("SQ_WAVE_CYCLES", [])) for v in util.values())
    wa = sum(sum(v.get("SQ_WAIT_ANY", [])) for v in util.values())
    out = {
        "round": tag,
        "command": "tools/pmc.sh (rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_* GRBM_GUI_ACTIVE> --kernel-include-regex "
                   "conv1d_mfma -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline; one counter group per run), "
                   "summarised by tools/pmc_summary.py",
        "launches": n,
        "fetch_bytes_raw": fb,
        "write_bytes_raw": wb,
        "fetch_correction": fcorr,
        "fetch_bytes_corrected": fb / fcorr,
        f"conv_{mode}_bytes_per_launch": (fb / fcorr + wb) / max(1, n),
        "calibration": cal,
        "calibration_note": "the microarch guide calibrates FETCH_SIZE only for 16-B/lane streams (reads 1/2 there) and asks "
                            "for a calibration on a known byte count otherwise.  Here: the first conv of a stage-4 ResBlock "
                            "pair reads exactly one 2.556 GB fp32 tensor with 4-B/lane loads and FETCH_SIZE reports "
                            "fetch_raw_over_known of it; every FETCH_SIZE is divided by that factor.  WRITE_SIZE reproduces "
                            "the 4-B/lane fp32 stores exactly (second calibration kernel) and over-counts the 2-B/lane fp16 "
                            "stores by ~1.19x (partial lines); it is used raw.  The second kernel's corrected fetch stays "
                            "below its compulsory reads (its fp16 input was written by the launch before it and is partly "
                            "served on-die), so `traffic` is a lower-bound style estimate, not an exact byte count.",
        "mfma_busy_frac": (mf / (gui / 8.0 * 1024.0)) if gui else None,
        "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
        "wave_wait_frac": (wa / wc) if wc else None,
        "per_kernel": {k: {"launches": v["launches"], "fetch_bytes_mean": sum(v["fetch_bytes"]) / v["launches"],
                           "write_bytes_mean": (sum(v["write_bytes"]) / len(v["write_bytes"])) if v["write_bytes"] else None}
                       for k, v in per_kernel.items()},
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    prev = {}
    if os.path.isfile(path):
        try:
            prev = json.load(open(path))
        except Exception:
            prev = {}
    keep = {k: prev[k] for k in ("conv_fp32_bytes_per_launch",) if k in prev and mode != "fp32"}
    if keep:
        out["conv_fp32_bytes_per_launch"] = keep["conv_fp32_bytes_per_launch"]
        out["conv_fp32_note"] = "fp32-mode figure measured in r01 before the epilogue rewrite (same algorithmic bytes)"
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k != "per_kernel"}, indent=1))


if __name__ == "__main__":
    main()
