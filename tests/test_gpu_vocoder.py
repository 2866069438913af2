"""GPU: HIP vocoder vs (a) golden vectors produced by the REFERENCE HifiDecoder, (b) the oracle at other sizes,
(c) size-independent properties at the BASELINE utterance length."""
import glob
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import SPK_KEY, make_engine, rms

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vocoder_T*.npz")))


@pytest.fixture(scope="module")
def ctx():
    e, gpt_sd, xtts_sd, cond, spk = make_engine(1, max_seqs=2)
    yield e, xtts_sd, spk
    e.close()


def _check(got, ref):
    assert got.shape == ref.shape
    err = rms(got - ref)
    sig = rms(ref)
    # north_star: <= 1e-3 RMS on the waveform; synthetic weights give a quiet signal, so also bound the
    # error relative to the signal
    assert err <= 1e-3 and err <= 1e-2 * sig, (err, sig)
    return err, sig


@pytest.mark.parametrize("path", GOLDEN)
def test_vocoder_matches_reference_golden(ctx, path):
    e, _, _ = ctx
    g = np.load(path)
    wav = e.vocode(g["latents"], None, SPK_KEY)[0]
    err, sig = _check(wav, g["wav"])
    assert err < 1e-5, err          # fp32 MFMA is an exact-f32 fma chain: expect ~1e-7


def test_vocoder_ragged_batch_matches_oracle(ctx):
    from oracle import xtts_oracle as O
    e, xtts_sd, spk = ctx
    w = O.vocoder_effective_weights(xtts_sd)
    gen = torch.Generator().manual_seed(9)
    lens = [31, 7, 18]
    lat = torch.zeros(3, 31, 1024)
    for b, n in enumerate(lens):
        lat[b, :n] = torch.randn(n, 1024, generator=gen)
    wavs = e.vocode(lat.numpy(), lens, SPK_KEY)
    for b, n in enumerate(lens):
        ref = O.hifi_decoder_forward(w, lat[b:b + 1, :n], spk).reshape(-1).numpy()
        _check(wavs[b], ref)


def test_vocoder_full_length_properties(ctx, dims):
    """280 latent frames (the 200-char utterance of BASELINE configs): length, range, determinism, batch invariance."""
    e, _, _ = ctx
    gen = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 280, 1024, generator=gen).numpy()
    w1 = e.vocode(lat, None, SPK_KEY)[0]
    assert w1.shape == (dims.voc.samples_for_latents(280),) == (312064,)
    assert np.isfinite(w1).all() and np.abs(w1).max() <= 1.0
    w2 = e.vocode(lat, None, SPK_KEY)[0]
    assert np.array_equal(w1, w2)
    both = e.vocode(np.concatenate([lat, lat[:, ::-1].copy()], axis=0), None, SPK_KEY)
    assert np.array_equal(both[0], w1)
    # locality: the first samples do not depend on frames far away (receptive field << 140 frames)
    lat3 = lat.copy()
    lat3[0, 200:] = 0.0
    w3 = e.vocode(lat3, None, SPK_KEY)[0]
    assert np.array_equal(w3[:100000], w1[:100000])


# ---- fp16-input MFMA mode (fp32 accumulate, fp32 activations in HBM) ---------------------------------------------
@pytest.fixture(scope="module")
def ctx16():
    e, gpt_sd, xtts_sd, cond, spk = make_engine(1, max_seqs=2, vocoder_fp16=True)
    yield e, xtts_sd, spk
    e.close()


@pytest.mark.parametrize("path", GOLDEN)
def test_vocoder_fp16_within_north_star_tolerance(ctx16, path):
    """Reference golden waveform: <= 1e-3 RMS (north_star) and <= 1 % of the signal RMS with fp16 MFMA inputs."""
    e, _, _ = ctx16
    g = np.load(path)
    wav = e.vocode(g["latents"], None, SPK_KEY)[0]
    err, sig = _check(wav, g["wav"])
    print(f"fp16 vocoder: rms err {err:.3e} signal rms {sig:.3e} ratio {err / sig:.3e}")


def test_vocoder_fp16_full_length(ctx16, ctx, dims):
    e16, xtts_sd, spk = ctx16
    e32, _, _ = ctx
    gen = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 280, 1024, generator=gen).numpy()
    w16 = e16.vocode(lat, [280, 133], SPK_KEY)
    w32 = e32.vocode(lat, [280, 133], SPK_KEY)
    assert w16[0].shape == (312064,) and w16[1].shape == w32[1].shape
    for a, b in zip(w16, w32):
        assert np.isfinite(a).all() and np.abs(a).max() <= 1.0
        err, sig = rms(a - b), rms(b)
        assert err <= 1e-3 and err <= 1e-2 * sig, (err, sig)
    assert np.array_equal(w16[0], e16.vocode(lat[:1], None, SPK_KEY)[0])       # deterministic, batch invariant


def test_vocoder_fp16_storage_of_resblock_tensors(ctx16, monkeypatch):
    """fp16 vocoder: the ResBlock c1 -> c2 intermediate and the residual stream between rounds live in HBM as halves (the
    reference's GPU path runs these blocks under fp16 autocast, hifigan_decoder.py:242); AUR_XT_F16=0 keeps both fp32 with
    the same fp16 MFMA inputs.  The two must agree far inside the north_star tolerance."""
    e16, _, _ = ctx16
    monkeypatch.setenv("AUR_XT_F16", "0")
    e_ref, *_ = make_engine(1, max_seqs=2, vocoder_fp16=True)
    try:
        gen = torch.Generator().manual_seed(4)
        lat = torch.randn(2, 57, 1024, generator=gen).numpy()
        a = e16.vocode(lat, [57, 20], SPK_KEY)
        b = e_ref.vocode(lat, [57, 20], SPK_KEY)
        for x, y in zip(a, b):
            err, sig = rms(x - y), rms(y)
            print(f"fp16 residual stream vs fp32 storage: rms err {err:.3e} signal rms {sig:.3e}")
            assert x.shape == y.shape and err <= 1e-4 and err <= 2e-3 * sig, (err, sig)
    finally:
        e_ref.close()


def test_vocoder_lds_dma_staging_equals_register_staging(ctx16, monkeypatch):
    """The ResBlock and transposed convs of the fp16 vocoder run on the LDS-DMA staged kernel (conv1d_dma_f16_kernel);
    AUR_CONV_DMA=0 selects the register-staged one.  Same chunks, same tap order, same MFMAs: the waveforms are equal bit for bit."""
    e16, _, _ = ctx16
    monkeypatch.setenv("AUR_CONV_DMA", "0")
    e_reg, *_ = make_engine(1, max_seqs=2, vocoder_fp16=True)
    try:
        gen = torch.Generator().manual_seed(6)
        lat = torch.randn(2, 61, 1024, generator=gen).numpy()
        a = e16.vocode(lat, [61, 17], SPK_KEY)
        b = e_reg.vocode(lat, [61, 17], SPK_KEY)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    finally:
        e_reg.close()


# ---- the reference class at the BASELINE utterance length, and at speech amplitude --------------------------------------
def test_vocoder_reference_golden_at_baseline_length(ctx, ctx16):
    """tests/golden/vocoder_ref_T280.npz: the waveform the reference's own HifiDecoder produces for 280 latent frames (312 064
    samples).  Exact-f32 mode ~1e-8; the default fp16 mode (fp16 MFMA inputs, ResBlock tensors stored as halves, fused rounds)
    inside the north-star bar — the fp16 path pinned to the reference itself at the size the bench runs."""
    from tests.test_oracle_vocoder import golden_T280_latents
    lat, g = golden_T280_latents()
    e32, _, _ = ctx
    e16, _, _ = ctx16
    w32 = e32.vocode(lat.numpy(), None, SPK_KEY)[0]
    err32, _ = _check(w32, g["wav"])
    assert err32 < 1e-5, err32
    w16 = e16.vocode(lat.numpy(), None, SPK_KEY)[0]
    err16, sig = _check(w16, g["wav"])
    print(f"fp16 vocoder vs reference class at T=280: rms err {err16:.3e} signal rms {sig:.3e} ratio {err16 / sig:.3e}")


def test_vocoder_at_speech_amplitude(monkeypatch):
    """checkpoint.make_loud_vocoder: output RMS 0.12 (peaks 0.5), where the north-star 1e-3 ABSOLUTE bar is the binding one.
    Golden from the reference's HifiDecoder on the same weights.  The default fp16 mode stores the residual stream and the MRF
    sums as halves (the reference keeps the MRF sum fp32, hifigan_decoder.py:253-255); AUR_XT_F16=0 keeps them fp32 — both are
    measured against the reference and both must pass."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_loud_vocoder
    from auralis_amd.weights import pack_all
    from tests.gpu_util import packed_weights
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocoder_loud_T47.npz"))
    _, gpt_sd, xtts_sd = packed_weights(1)
    packed = pack_all(gpt_sd, make_loud_vocoder(xtts_sd, float(g["up_gain"]), float(g["post_gain"])))
    errs = {}
    for name, env, f16 in (("fp32", None, False), ("fp16, halves in HBM", None, True), ("fp16 inputs, fp32 in HBM", "0", True)):
        if env is not None:
            monkeypatch.setenv("AUR_XT_F16", env)
        e = NativeEngine(n_layer=1, max_seqs=2, vocoder_fp16=f16)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, np.zeros((32, 1024), np.float32), g["speaker"].reshape(512))
            wav = e.vocode(g["latents"], None, SPK_KEY)[0]
        finally:
            e.close()
            monkeypatch.delenv("AUR_XT_F16", raising=False)
        err, sig = rms(wav - g["wav"]), rms(g["wav"])
        errs[name] = err
        assert sig > 0.1
        assert err <= 1e-3 and err <= 1e-2 * sig, (name, err, sig)
    print("speech-amplitude vocoder, RMS error against the reference class: " + ", ".join(f"{k}: {v:.3e}" for k, v in errs.items()))
    assert errs["fp32"] < 1e-5
