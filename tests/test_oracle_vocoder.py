"""CPU: pin oracle/xtts_oracle.py's vocoder to the reference's own HifiDecoder (golden fixtures produced by
oracle/make_golden.py from the unmodified reference class; live comparison when /root/reference exists)."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import xtts_oracle as O
from oracle.ref_import import build_reference_decoder, reference_available

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vocoder_T*.npz")))


def test_golden_fixtures_exist():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN)
def test_oracle_matches_reference_golden(path, xtts_sd):
    g = np.load(path)
    w = O.vocoder_effective_weights(xtts_sd)
    wav = O.hifi_decoder_forward(w, torch.from_numpy(g["latents"]), torch.from_numpy(g["speaker"]))
    ref = torch.from_numpy(g["wav"])
    assert wav.numel() == ref.numel()
    err = (wav.reshape(-1) - ref).abs().max().item()
    assert err < 1e-6, err


@pytest.mark.parametrize("L,scale", [(7, 4.0), (280, 4.0), (1120, 24000 / 22050), (20, 24000 / 22050), (1, 4.0)])
def test_interp_closed_form_matches_torch(L, scale):
    torch.manual_seed(L)
    x = torch.randn(1, 5, L)
    ref = F.interpolate(x, scale_factor=scale, mode="linear", align_corners=False)
    got = O.interp_linear(x, scale)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 5e-5   # fp32 source-index rounding (SURVEY A7'(i): 9e-6 typical)


def test_frame_count_formula(dims):
    # 280 latents -> 1120 -> 1219 frames -> 312064 samples
    assert dims.voc.frames_for_latents(280) == 1219
    assert dims.voc.samples_for_latents(280) == 312064
    z = O.interp_linear(O.interp_linear(torch.zeros(1, 2, 280), 4.0), 24000 / 22050)
    assert z.shape[-1] == 1219


def test_weight_norm_fold_dims():
    # Conv1d: norm over (Cin,k) per Cout; ConvTranspose1d: dim 0 is Cin (SURVEY A7'(iii))
    v = torch.randn(6, 4, 3)
    g = torch.rand(6, 1, 1) + 0.5
    w = O.fold_weight_norm(g, v)
    assert torch.allclose(w.flatten(1).norm(dim=1), g.flatten(), atol=1e-6)


@pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")
def test_oracle_matches_live_reference(xtts_sd, conditioning):
    dec = build_reference_decoder(xtts_sd)
    torch.manual_seed(3)
    lat = torch.randn(1, 9, 1024)
    with torch.no_grad():
        ref = dec(lat, g=conditioning[1])
    w = O.vocoder_effective_weights(xtts_sd)
    got = O.hifi_decoder_forward(w, lat, conditioning[1])
    assert (got - ref).abs().max().item() < 1e-6


def golden_T280_latents():
    """The 280-frame latents of tests/golden/vocoder_ref_T280.npz, regenerated from the stored seed and checked against the stored sums."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocoder_ref_T280.npz"))
    lat = torch.randn(1, 280, 1024, generator=torch.Generator().manual_seed(int(g["latents_seed"])))
    assert abs(lat.double().sum().item() - float(g["latents_sum"])) < 1e-6 and abs(lat.double().abs().sum().item() - float(g["latents_abs_sum"])) < 1e-6, \
        "torch's CPU generator changed: regenerate the fixture (python -m oracle.make_golden)"
    return lat, g


def test_oracle_matches_reference_golden_at_baseline_length(xtts_sd):
    """280 latent frames -> 312 064 samples: the restatement against the reference class at the size the bench runs."""
    lat, g = golden_T280_latents()
    wav = O.hifi_decoder_forward(O.vocoder_effective_weights(xtts_sd), lat, torch.from_numpy(g["speaker"])).reshape(-1)
    assert wav.numel() == 312064 == g["wav"].size
    assert (wav - torch.from_numpy(g["wav"])).abs().max().item() < 1e-6


def test_oracle_matches_reference_golden_loud(xtts_sd):
    """speech-amplitude synthetic vocoder (checkpoint.make_loud_vocoder): output RMS 0.12"""
    from auralis_amd.checkpoint import make_loud_vocoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocoder_loud_T47.npz"))
    sd = make_loud_vocoder(xtts_sd, float(g["up_gain"]), float(g["post_gain"]))
    wav = O.hifi_decoder_forward(O.vocoder_effective_weights(sd), torch.from_numpy(g["latents"]), torch.from_numpy(g["speaker"])).reshape(-1)
    ref = torch.from_numpy(g["wav"])
    assert ref.pow(2).mean().sqrt().item() > 0.1
    assert (wav - ref).abs().max().item() < 5e-6
