"""Generate tests/golden/ fixtures by running the REFERENCE's own HifiDecoder (build container only).

TEST INFRASTRUCTURE ONLY.  Usage: python -m oracle.make_golden
The reference class is imported unmodified from /root/reference (oracle/ref_import.py); weights are the
seeded synthetic checkpoint (auralis_amd.checkpoint.make_synthetic_xtts, seed 1234) so only inputs and
reference outputs are stored.  These vectors pin oracle/xtts_oracle.py (CPU test) and the HIP vocoder
(GPU test) to the reference's arithmetic.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_xtts  # noqa: E402
from auralis_amd.config import XTTSDims  # noqa: E402
from oracle.ref_import import build_reference_decoder  # noqa: E402


def main():
    dims = XTTSDims()
    sd = make_synthetic_xtts(dims, seed=1234)
    dec = build_reference_decoder(sd)
    _, spk = make_synthetic_conditioning(dims)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for T, seed in ((5, 101), (23, 102)):
        g = torch.Generator().manual_seed(seed)
        lat = torch.randn(1, T, 1024, generator=g)
        with torch.no_grad():
            wav = dec(lat, g=spk)
        np.savez_compressed(os.path.join(out_dir, f"vocoder_T{T}.npz"),
                            latents=lat.numpy(), speaker=spk.numpy(), wav=wav.numpy().reshape(-1),
                            weights_seed=np.int64(1234), latents_seed=np.int64(seed))
        print(f"T={T}: wav {tuple(wav.shape)} rms {wav.pow(2).mean().sqrt().item():.5f}")


if __name__ == "__main__":
    main()
