"""Load / wait / MFMA / barrier skeleton of the gfx950 kernels of one translation unit (no GPU needed).

usage: python tools/isa_skeleton.py auralis_amd/csrc/gpt_kernels.hip [kernel-name-substring] [max chars per kernel]

Compiles the file to assembly (hipcc -S --cuda-device-only) and prints, per kernel whose mangled name contains the substring,
VGPRs, LDS bytes and the order of the instructions that decide how memory latency is covered:

    L  global / buffer load      D  LDS-DMA copy (global_load_lds)      S  global store / atomic
    w  ds_write                  M  MFMA                                 B  s_barrier
    [Wn]  s_waitcnt vmcnt(n)     |  basic-block boundary

Runs are counted (M24 = 24 MFMAs in a row).  What to look for: a `[W0]` (or a count smaller than the loads issued since)
between a group of prefetch loads and the MFMAs it was meant to overlap.  hipcc's wait insertion takes the stricter of the two
paths at every join, so a branch around a prefetch load (`if (k + 2 < n) load(...)`) makes "nothing newer in flight" one of the
paths and the loop ends up waiting for the loads it has just issued.  Round 4 found exactly that in gemm_tile_split_kernel
(`[W4]..[W0]` in front of every step's LDS stores) -- 27.5 -> 25.7 ms per prefill once the loads were
unconditional."""
import os
import re
import subprocess
import sys
import tempfile


def skeleton(asm: str, pat: str, width: int) -> None:
    for m in re.finditer(r"^(_ZN3aur\w*" + re.escape(pat) + r"\w*):.*?^\s*\.end_amdhsa_kernel", asm, re.S | re.M):
        body, name = m.group(0), m.group(1)
        seq = []
        for line in body.splitlines():
            t = line.strip()
            if t.startswith("v_mfma"):
                seq.append("M")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("W" + t.split("vmcnt(")[1].split(")")[0])
            elif t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t):
                seq.append("D")
            elif t.startswith(("global_load", "buffer_load")):
                seq.append("L")
            elif t.startswith(("global_store", "global_atomic", "buffer_store")):
                seq.append("S")
            elif t.startswith("s_barrier"):
                seq.append("B")
            elif t.startswith("ds_write"):
                seq.append("w")
            elif t.startswith(".LBB"):
                seq.append("|")
        out = "".join(x if len(x) == 1 else "[" + x + "]" for x in seq)
        for c in "MwLDS|":
            out = re.sub(re.escape(c) + r"{2,}", lambda mm: "%s%d " % (c, len(mm.group(0))), out)
        vg = [l.split()[-1] for l in body.splitlines() if "next_free_vgpr" in l]
        lds = [l.split()[-1] for l in body.splitlines() if "group_segment_fixed_size" in l]
        try:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            demangled = name
        print(f"{demangled[:140]}\n  vgpr {vg[0] if vg else '?'}  lds {lds[0] if lds else '?'} B\n  {out[:width]}")


def main() -> None:
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                            "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-3000:])
        with open(out) as f:
            skeleton(f.read(), pat, width)


if __name__ == "__main__":
    main()
