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
from auralis_amd.checkpoint import make_loud_vocoder, make_synthetic_conditioning, make_synthetic_xtts  # noqa: E402
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
    # BASELINE utterance length (280 latent frames -> 312 064 samples): the fp16 vocoder is pinned to the reference class at the
    # size the bench runs, not only through the restatement.  The latents are regenerated from the seed (their float64 sum is
    # stored, so a change of torch's generator is noticed), only the reference waveform is kept (1.2 MB).
    g = torch.Generator().manual_seed(103)
    lat = torch.randn(1, 280, 1024, generator=g)
    with torch.no_grad():
        wav = dec(lat, g=spk)
    np.savez_compressed(os.path.join(out_dir, "vocoder_ref_T280.npz"), wav=wav.numpy().reshape(-1), speaker=spk.numpy(),
                        weights_seed=np.int64(1234), latents_seed=np.int64(103), latents_sum=np.float64(lat.double().sum().item()),
                        latents_abs_sum=np.float64(lat.double().abs().sum().item()))
    print(f"T=280: wav {tuple(wav.shape)} rms {wav.pow(2).mean().sqrt().item():.5f}")
    # speech-amplitude variant of the synthetic vocoder (make_loud_vocoder): output RMS 0.12, so the north-star 1e-3 absolute bar
    # is the one that binds, and the fp16-stored tensors carry correspondingly larger values
    dec_l = build_reference_decoder(make_loud_vocoder(sd))
    g = torch.Generator().manual_seed(104)
    lat = torch.randn(1, 47, 1024, generator=g)
    with torch.no_grad():
        wav = dec_l(lat, g=spk)
    np.savez_compressed(os.path.join(out_dir, "vocoder_loud_T47.npz"), latents=lat.numpy(), speaker=spk.numpy(), wav=wav.numpy().reshape(-1),
                        weights_seed=np.int64(1234), latents_seed=np.int64(104), up_gain=np.float64(3.0), post_gain=np.float64(1.5))
    print(f"loud T=47: wav {tuple(wav.shape)} rms {wav.pow(2).mean().sqrt().item():.5f} peak {wav.abs().max().item():.3f}")


if __name__ == "__main__":
    main()
