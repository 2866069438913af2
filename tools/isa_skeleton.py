"""Load / wait / MFMA / barrier skeleton of the gfx950 kernels of one translation unit (no GPU needed).

usage: python tools/isa_skeleton.py auralis_amd/csrc/gpt_kernels.hip [kernel-name-substring] [max chars per kernel]

Compiles the file to assembly with the build's flags (hipcc -S --cuda-device-only) and prints, per kernel whose mangled name
contains the substring, VGPRs, LDS bytes, the preloaded kernel-argument dwords and the order of the instructions that decide how
memory latency is covered:

    k  s_load (scalar: kernel arguments, uniform tables)      [Kn]  s_waitcnt lgkmcnt(n)
    L  global / buffer load      F  flat load                  D  LDS-DMA copy (global_load_lds)      S  global store / atomic
    w  ds_write                  M  MFMA                       B  s_barrier
    [Wn]  s_waitcnt vmcnt(n)     |  basic-block boundary       >  the jump over the kernarg-preload compatibility prologue

Runs are counted (M24 = 24 MFMAs in a row).  What to look for:
  * a `[W0]` (or a count smaller than the loads issued since) between a group of prefetch loads and the MFMAs it was meant to
    overlap.  hipcc's wait insertion takes the stricter of the two paths at every join, so a branch around a prefetch load makes
    "nothing newer in flight" one of the paths (round 4: gemm_tile_split_kernel, 27.5 -> 25.7 ms per prefill);
  * `k [K0] k [K0] ...` in front of the first `L`: every pair is one DEPENDENT scalar round trip before the first byte of data is
    requested.  hipcc sinks each kernel-argument fetch to its first use; round 4's decode kernels had three to six of them
    (VERDICT r04).  Everything behind `>` runs with the leading arguments already in SGPRs (-amdgpu-kernarg-preload-count);
  * an `F`: a pointer that went through an asm output operand has lost its address space; after a FLAT load every wait is vmcnt(0);
  * an `L` inside a `|...|` block that follows a `[Wn]` of the same iteration: hipcc sinks a load into the only branch that uses
    its result (round 5: the V load of the decode attention, one dependent round trip per 64 tokens).

`prologue(body)` returns what tests/test_isa_prologue.py asserts on: the scalar round trips (waits with a scalar load outstanding), scalar waits, vector waits, barriers and flat loads between
the kernel's real entry (behind the preload prologue) and its first wide data load."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_flags():
    """The build's own flag list (auralis_amd/build.py), minus what only matters when linking."""
    sys.path.insert(0, ROOT)
    from auralis_amd.build import FLAGS
    return [f for f in FLAGS if f != "-fPIC"]


def compile_to_asm(src: str) -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *build_flags(), "-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        with open(out) as f:
            return f.read()


def kernels(asm: str, pat: str = ""):
    """[(mangled name, body)] of every aur:: kernel whose mangled name contains `pat`."""
    return [(m.group(1), m.group(0))
            for m in re.finditer(r"^(_ZN3aur\w*" + re.escape(pat) + r"\w*):.*?^\s*\.end_amdhsa_kernel", asm, re.S | re.M)]


def tokens(body: str):
    seq = []
    for line in body.splitlines():
        t = line.strip()
        if t.startswith("v_mfma"):
            seq.append("M")
        elif t.startswith("s_waitcnt"):
            if "vmcnt" in t:
                seq.append("W" + t.split("vmcnt(")[1].split(")")[0])
            if "lgkmcnt" in t:
                seq.append("K" + t.split("lgkmcnt(")[1].split(")")[0])
        elif t.startswith("s_load") or t.startswith("s_buffer_load"):
            seq.append("k")
        elif t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t):
            seq.append("D")
        elif t.startswith("flat_load"):
            seq.append("F")
        elif t.startswith(("global_load", "buffer_load")):
            seq.append("L4" if "dwordx4" in t else "L")
        elif t.startswith(("global_store", "global_atomic", "buffer_store", "flat_store", "flat_atomic")):
            seq.append("S")
        elif t.startswith("s_barrier"):
            seq.append("B")
        elif t.startswith("ds_write"):
            seq.append("w")
        elif t.startswith("s_branch"):
            seq.append(">")
        elif t.startswith(".LBB"):
            seq.append("|")
    return seq


def prologue(body: str, skip: int = 0) -> dict:
    """What stands between the kernel's entry and its first 16-byte data load (its (skip + 1)-th: the decode attention requests its
    q row first, the K/V rows are what the launch is about).  With kernarg preloading the function starts with a
    compatibility prologue (s_load of the preloaded range, wait, s_branch over the padding) that firmware with preload support
    skips: counting starts behind its s_branch."""
    seq = tokens(body)
    preload = [l.split()[-1] for l in body.splitlines() if "kernarg_preload_length" in l]
    start = 0
    if preload and int(preload[0]) > 0 and ">" in seq:
        start = seq.index(">") + 1
    wide = [i for i in range(start, len(seq)) if seq[i] == "L4"]
    first = wide[skip] if len(wide) > skip else len(seq)
    pre = seq[start:first]
    # a scalar wait costs a round trip only when a scalar load is outstanding beyond what it lets pass (a wait at a loop header that
    # the first iteration reaches with nothing in flight is free)
    pending = trips = 0
    for x in pre:
        if x == "k":
            pending += 1
        elif x.startswith("K"):
            n = int(x[1:])
            if pending > n:
                trips += 1
                pending = n
    return {
        "preload_dwords": int(preload[0]) if preload else 0,
        "scalar_round_trips": trips,
        "scalar_waits": sum(1 for x in pre if x.startswith("K")),
        "scalar_loads": sum(1 for x in pre if x == "k"),
        "vector_waits": sum(1 for x in pre if x.startswith("W")),
        "barriers": sum(1 for x in pre if x == "B"),
        "flat_loads": sum(1 for x in seq if x == "F"),
        "found_data_load": first < len(seq),
    }


def skeleton(asm: str, pat: str, width: int) -> None:
    for name, body in kernels(asm, pat):
        out = "".join(x if len(x) == 1 else ("L" if x == "L4" else "[" + x + "]") for x in tokens(body))
        for c in "MwLDSk|":
            out = re.sub(re.escape(c) + r"{2,}", lambda mm: "%s%d " % (c, len(mm.group(0))), out)
        vg = [l.split()[-1] for l in body.splitlines() if "next_free_vgpr" in l]
        lds = [l.split()[-1] for l in body.splitlines() if "group_segment_fixed_size" in l]
        p = prologue(body)
        try:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            demangled = name
        print(f"{demangled[:140]}\n  vgpr {vg[0] if vg else '?'}  lds {lds[0] if lds else '?'} B  preloaded kernarg dwords {p['preload_dwords']}  "
              f"before the first 16-byte load: {p['scalar_waits']} scalar waits, {p['vector_waits']} vector waits, {p['barriers']} barriers\n  {out[:width]}")


def main() -> None:
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    skeleton(compile_to_asm(src), pat, width)


if __name__ == "__main__":
    main()
