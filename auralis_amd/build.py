"""Build the gfx950 shared library in-tree: auralis_amd/_C/libauralis_amd.so (hipcc, no cmake)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_C")
LIB = os.path.join(OUT_DIR, "libauralis_amd.so")
SOURCES = ["vocoder_kernels.hip", "gpt_kernels.hip", "cond_kernels.hip", "engine.hip"]
HEADERS = ["common.h", "vocoder_kernels.h", "gpt_kernels.h", "gemm_rows_kernel.inc", "cond_kernels.h", "cond_net.h", os.path.join("..", "..", "include", "auralis_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count=16: the leading scalar kernel arguments (up to 14 dwords next to the kernarg pointer) arrive in SGPRs
# with the wave instead of through dependent s_load round trips; a kernel whose firmware does not preload runs its compatibility
# prologue (one s_load burst).  The decode kernels order their arguments for it (gemm_rows_kernel.inc, paged_attention_kernel).
# -Rpass-analysis=kernel-resource-usage: the compiler reports every kernel's registers, scratch and spills while it compiles; the
# build keeps them next to the object (_C/<unit>.resources.json) and tests/test_isa_prologue.py holds the kernels to them without
# compiling the four-minute vocoder unit a second time.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-Rpass-analysis=kernel-resource-usage"]


HASH_FILE = LIB + ".srchash"


def source_hash() -> str:
    """sha256 over every translation unit, every header (csrc/ and include/) and the compiler flags: what the binary is keyed on
    (file times say nothing once a tree has been copied to another machine)."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for rel in sorted(SOURCES + HEADERS):
        path = os.path.normpath(os.path.join(CSRC, rel))
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def resources_path(src: str) -> str:
    return os.path.join(OUT_DIR, src.replace(".hip", ".resources.json"))


def parse_resource_remarks(stderr: str) -> dict:
    """hipcc's kernel-resource-usage remarks -> {mangled kernel name: {"vgprs", "agprs", "sgprs", "scratch", "sgpr_spill", "vgpr_spill",
    "lds", "occupancy"}}"""
    import re
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def kernel_resources(src: str) -> dict:
    """The resource report of one translation unit of the CURRENT build (builds first when the tree is stale)."""
    build()
    with open(resources_path(src)) as f:
        return json.load(f)


def _stale(want: str) -> bool:
    if not os.path.isfile(LIB) or not os.path.isfile(HASH_FILE):
        return True
    with open(HASH_FILE) as f:
        return f.read().strip() != want


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link the C-ABI library, unless the library on disk was built from
    exactly these sources.  Prints which of the two happened.  Returns the library's path."""
    want = source_hash()
    if not force and not _stale(want):
        print(f"[auralis_amd.build] reused {os.path.relpath(LIB, os.path.dirname(HERE))} (source hash {want[:16]})", file=sys.stderr, flush=True)
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)

    def unit_hash(src: str) -> str:
        """what one object file is keyed on: its translation unit, every header (any of them may be included) and the flags"""
        import hashlib
        h = hashlib.sha256()
        h.update(" ".join(FLAGS).encode())
        for rel in [src] + sorted(HEADERS):
            with open(os.path.normpath(os.path.join(CSRC, rel)), "rb") as f:
                h.update(rel.encode())
                h.update(f.read())
        return h.hexdigest()

    compiled = []

    def cc(src: str) -> str:
        obj = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        tag, want_u = obj + ".srchash", unit_hash(src)
        if not force and os.path.isfile(obj) and os.path.isfile(tag) and open(tag).read().strip() == want_u:
            return obj   # this unit and the headers are unchanged (the vocoder unit alone takes four minutes)
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        with open(resources_path(src), "w") as f:
            json.dump(parse_resource_remarks(r.stderr), f, indent=0, sort_keys=True)
        with open(tag, "w") as f:
            f.write(want_u + "\n")
        compiled.append(src)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    with open(HASH_FILE, "w") as f:
        f.write(want + "\n")
    print(f"[auralis_amd.build] compiled {len(compiled)} of {len(SOURCES)} translation units with hipcc for gfx950 ({', '.join(compiled) or 'objects up to date'}) -> "
          f"{os.path.relpath(LIB, os.path.dirname(HERE))} (source hash {want[:16]})", file=sys.stderr, flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
