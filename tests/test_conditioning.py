"""CPU: the speaker-conditioning restatement (auralis_amd/conditioning.py) against the reference's own module classes
(imported unmodified when /root/reference exists) plus front-end property tests."""
import math

import numpy as np
import pytest
import torch

from oracle import conditioning_oracle as Cn
from auralis_amd.checkpoint import conditioning_param_shapes, make_synthetic_conditioning_weights
from oracle.ref_import import reference_available

needs_ref = pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")


@pytest.fixture(scope="module")
def cond_sd(dims):
    return make_synthetic_conditioning_weights(dims, seed=99)


@needs_ref
def test_param_names_and_shapes_match_reference_modules(dims):
    from oracle.ref_import import load_reference_hifigan, load_reference_xtts_layer
    le, pe, hg = load_reference_xtts_layer("latent_encoder"), load_reference_xtts_layer("perceiver_encoder"), load_reference_hifigan()
    ref = {}
    ref.update({"conditioning_encoder." + k: tuple(v.shape) for k, v in le.ConditioningEncoder(80, 1024, num_attn_heads=16).state_dict().items()})
    ref.update({"conditioning_perceiver." + k: tuple(v.shape) for k, v in pe.PerceiverResampler(
        dim=1024, depth=2, dim_context=1024, num_latents=32, dim_head=64, heads=8, ff_mult=4).state_dict().items()})
    se = hg.HifiDecoder().speaker_encoder
    ref.update({"hifigan_decoder.speaker_encoder." + k: tuple(v.shape) for k, v in se.state_dict().items()})
    assert conditioning_param_shapes(dims) == ref


@needs_ref
def test_conditioning_encoder_and_perceiver_match_reference(cond_sd):
    from oracle.ref_import import load_reference_xtts_layer
    le, pe = load_reference_xtts_layer("latent_encoder"), load_reference_xtts_layer("perceiver_encoder")
    enc = le.ConditioningEncoder(80, 1024, num_attn_heads=16).eval()
    enc.load_state_dict({k[len("conditioning_encoder."):]: v for k, v in cond_sd.items() if k.startswith("conditioning_encoder.")})
    per = pe.PerceiverResampler(dim=1024, depth=2, dim_context=1024, num_latents=32, dim_head=64, heads=8, ff_mult=4).eval()
    per.load_state_dict({k[len("conditioning_perceiver."):]: v for k, v in cond_sd.items() if k.startswith("conditioning_perceiver.")})
    mel = torch.randn(1, 80, 57, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref_h = enc(mel)
        ref_l = per(ref_h.permute(0, 2, 1))
        got_h = Cn.conditioning_encoder(cond_sd, mel)
        got_l = Cn.perceiver_resampler(cond_sd, got_h.permute(0, 2, 1))
    assert (got_h - ref_h).abs().max().item() < 1e-4
    assert got_l.shape == (1, 32, 1024) and (got_l - ref_l).abs().max().item() < 1e-4


@needs_ref
def test_speaker_encoder_body_matches_reference(cond_sd):
    from oracle.ref_import import load_reference_hifigan
    se = load_reference_hifigan().HifiDecoder().speaker_encoder.eval()
    se.load_state_dict({k[len("hifigan_decoder.speaker_encoder."):]: v for k, v in cond_sd.items()
                        if k.startswith("hifigan_decoder.speaker_encoder.")})
    se.use_torch_spec = False            # the torchaudio mel front-end is stubbed offline: feed the mel directly
    mel = torch.rand(2, 64, 120, generator=torch.Generator().manual_seed(2)) * 3.0
    with torch.no_grad():
        ref = se(mel.clone(), l2_norm=True)
        got = Cn.speaker_encoder_from_mel(cond_sd, mel.clone())
    assert got.shape == (2, 512) and (got - ref).abs().max().item() < 1e-5


def test_resample_keeps_a_tone_and_length():
    sr, new = 44100, 22050
    t = torch.arange(sr, dtype=torch.float32) / sr
    x = (0.5 * torch.sin(2 * math.pi * 1000.0 * t))[None]
    y = Cn.resample(x, sr, new)
    assert y.shape[-1] == new
    ref = 0.5 * torch.sin(2 * math.pi * 1000.0 * torch.arange(new, dtype=torch.float32) / new)
    assert (y[0, 200:-200] - ref[200:-200]).abs().max().item() < 2e-3
    up = Cn.resample(x[:, :4410], 22050, 24000)
    assert up.shape[-1] == math.ceil(4410 * 24000 / 22050)


def test_mel_filterbank_properties():
    fb = Cn.mel_filterbank(1025, 0.0, 8000.0, 80, 22050, slaney_norm=True)
    assert fb.shape == (1025, 80) and (fb >= 0).all()
    peaks = fb.argmax(dim=0)
    assert (peaks[1:] >= peaks[:-1]).all()                 # centre frequencies increase
    assert fb[int(8000 / (22050 / 2) * 1024) + 3:, :].abs().max() == 0    # nothing above f_max
    fb2 = Cn.mel_filterbank(257, 0.0, 8000.0, 64, 16000, slaney_norm=False)
    assert abs(fb2.max().item() - 1.0) < 0.05              # un-normalised triangles peak at ~1


def test_wav_roundtrip_and_end_to_end_shapes(tmp_path, cond_sd):
    from auralis_amd import TTSOutput
    sr = 44100
    t = np.arange(int(2.5 * sr)) / sr
    wav = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 1760 * t)).astype(np.float32)
    p = tmp_path / "voice.wav"
    TTSOutput(array=wav, sample_rate=sr).save(p)
    a, got_sr = Cn.read_wav(str(p))
    assert got_sr == sr and a.shape == (1, len(wav)) and np.abs(a.numpy()[0] - wav).max() < 1e-4
    sd = dict(cond_sd)
    sd["mel_stats"] = torch.ones(80)
    cond, spk = Cn.get_conditioning_latents(sd, [str(p), p.read_bytes()], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=2)
    assert cond.shape == (1, 32, 1024) and spk.shape == (1, 512, 1)
    assert torch.isfinite(cond).all() and abs(spk.norm().item() - 1.0) < 1e-4
    with pytest.raises(ValueError):
        Cn.read_wav(b"not a wav file at all, definitely not....................")


# ------------------------------------------------------------------------------------------------ real-speech golden + front-end
def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cond_female_6s.npz"))


def _wav_bytes(pcm16: np.ndarray, sr: int = 22050) -> bytes:
    import io
    import wave
    b = io.BytesIO()
    with wave.open(b, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm16.astype(np.int16).tobytes())
    return b.getvalue()


def _wav_bytes_f32(x: np.ndarray, sr: int = 22050) -> bytes:
    """IEEE-float WAV: the decoder hands back exactly these float32 samples."""
    import struct
    data = np.asarray(x, dtype="<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, sr, sr * 4, 4, 32)
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + \
        b"data" + struct.pack("<I", len(data)) + data


def _golden_sd(dims):
    sd = make_synthetic_conditioning_weights(dims, seed=99)
    sd["mel_stats"] = torch.ones(80)
    return sd


def test_conditioning_of_real_speech_matches_reference_classes(dims):
    """6 s of the reference's female.wav (fixture made by oracle/make_golden_cond.py): conditioning.py end to end against the
    outputs the reference's ConditioningEncoder / PerceiverResampler / ResNetSpeakerEncoder classes gave on the same clip."""
    g = _golden()
    cond, spk = Cn.get_conditioning_latents(_golden_sd(dims), [_wav_bytes(g["pcm16"])], max_ref_length=30, gpt_cond_len=6,
                                            gpt_cond_chunk_len=6)
    assert cond.shape == (1, 32, 1024) and spk.shape == (1, 512, 1)
    assert (cond - torch.from_numpy(g["gpt_cond_latent"])).abs().max().item() < 2e-4
    assert (spk - torch.from_numpy(g["speaker_embedding"])).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_conditioning_on_gpu_matches_reference_classes(dims):
    """The same clip through conditioning.py on cuda:0 (PyTorch-ROCm: rocFFT STFT, MIOpen convs, rocBLAS attention)."""
    g = _golden()
    cond, spk = Cn.get_conditioning_latents(_golden_sd(dims), [_wav_bytes(g["pcm16"])], max_ref_length=30, gpt_cond_len=6,
                                            gpt_cond_chunk_len=6, device="cuda")
    assert cond.is_cuda and spk.is_cuda
    assert (cond.cpu() - torch.from_numpy(g["gpt_cond_latent"])).abs().max().item() < 2e-3
    assert (spk.cpu() - torch.from_numpy(g["speaker_embedding"])).abs().max().item() < 1e-4


def test_mel_front_end_against_an_independent_restatement():
    """torchaudio is not installed, so the mel front-end cannot be pinned to the reference's MelSpectrogram objects
    (common/utilities.py:9-71, hifigan_decoder.py:537-548).  Independent cross-check on real speech: scipy's STFT (zero phase
    reference implementation, own framing) + a filterbank coded from the HTK-mel / Slaney-normalisation definitions."""
    import scipy.signal as ss
    g = _golden()
    x = g["pcm16"].astype(np.float64) / 32767.0

    def mel_ref(x, sr, n_fft, win_len, hop, n_mels, fmin, fmax, window, slaney):
        win = ss.get_window(window, win_len, fftbins=True)
        pad = (n_fft - win_len) // 2
        win = np.pad(win, (pad, n_fft - win_len - pad))
        xp = np.pad(x, n_fft // 2, mode="reflect")
        frames = 1 + (len(xp) - n_fft) // hop
        idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
        spec = np.fft.rfft(xp[idx] * win[None, :], axis=1)
        power = (spec.real ** 2 + spec.imag ** 2).T                                  # [freq, frames]
        hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
        mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
        pts = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
        freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
        fb = np.zeros((n_mels, len(freqs)))
        for m in range(n_mels):
            lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
            up = (freqs - lo) / (ce - lo)
            down = (hi - freqs) / (hi - ce)
            fb[m] = np.maximum(0.0, np.minimum(up, down))
            if slaney:
                fb[m] *= 2.0 / (hi - lo)
        return fb @ power

    ours = Cn.mel_spectrogram(torch.from_numpy(x).float()[None], 22050, 2048, 1024, 256, 80, 0.0, 8000.0, "hann", slaney_norm=True)[0]
    ref = mel_ref(x, 22050, 2048, 1024, 256, 80, 0.0, 8000.0, "hann", True)
    assert ours.shape == ref.shape
    assert np.abs(ours.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    # the fixture's own log-mel (what the reference classes were fed) is this front-end's output
    logm = torch.log(torch.clamp(ours, min=1e-5)).numpy()
    assert np.abs(logm - g["mel_gpt"][0].astype(np.float32)).max() < 2e-2           # (fixture stores fp16)
    a16 = Cn.resample(torch.from_numpy(x).float()[None], 22050, 16000)[0].numpy().astype(np.float64)
    pre = np.concatenate([[a16[1]], a16])                                             # reflect pad of one sample
    pre = pre[1:] - 0.97 * pre[:-1]
    ours16 = Cn.mel_spectrogram(torch.from_numpy(pre).float()[None], 16000, 512, 400, 160, 64, window="hamming")[0]
    ref16 = mel_ref(pre, 16000, 512, 400, 160, 64, 0.0, 8000.0, "hamming", False)
    assert np.abs(ours16.numpy() - ref16).max() <= 2e-4 * np.abs(ref16).max()


def test_resampler_against_scipy_polyphase():
    """22 050 -> 16 000 Hz on real speech: the windowed-sinc resampler (torchaudio's default kernel restated) against
    scipy.signal.resample_poly (Kaiser-windowed polyphase FIR): different filters, so only band-limited agreement is asked."""
    import scipy.signal as ss
    g = _golden()
    x = g["pcm16"].astype(np.float64) / 32767.0
    ours = Cn.resample(torch.from_numpy(x).float()[None], 22050, 16000)[0].numpy()
    ref = ss.resample_poly(x, 320, 441)
    n = min(len(ours), len(ref))
    assert abs(len(ours) - len(ref)) <= 1
    err = ours[200:n - 200] - ref[200:n - 200]
    assert np.sqrt(np.mean(err ** 2)) < 0.05 * np.sqrt(np.mean(ref[200:n - 200] ** 2))


# ------------------------------------------------------------------------------------------------ HIP path (aur_compute_conditioning)
def _hip_engine(dims, sd):
    from auralis_amd._lib import NativeEngine
    from auralis_amd.weights import pack_conditioning
    eng = NativeEngine(n_layer=1, max_seqs=1)
    eng.load_weights(pack_conditioning(sd))
    return eng


@pytest.mark.gpu
def test_hip_conditioning_matches_reference_classes_on_real_speech(dims):
    """aur_compute_conditioning (hand-written HIP: DFT-as-GEMM mel front-ends, ConditioningEncoder, PerceiverResampler,
    ResNet-SE speaker encoder on exact-f32 MFMA GEMMs) on the golden clip, against the outputs of the reference's own module
    classes (tests/golden/cond_female_6s.npz, oracle/make_golden_cond.py)."""
    g = _golden()
    eng = _hip_engine(dims, _golden_sd(dims))
    try:
        pcm = g["pcm16"].astype(np.float32) / 32767.0
        cond, spk = eng.compute_conditioning([pcm], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
        assert cond.shape == (1, 32, 1024) and spk.shape == (1, 512, 1)
        e_c = float(np.abs(cond - g["gpt_cond_latent"]).max())
        e_s = float(np.abs(spk - g["speaker_embedding"]).max())
        print(f"HIP conditioning vs reference classes: latent max |err| {e_c:.3e} (rows are O(1)), embedding {e_s:.3e} (unit norm)")
        assert e_c < 2e-4 and e_s < 1e-5
        again = eng.compute_conditioning([pcm], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
        assert np.array_equal(again[0], cond) and np.array_equal(again[1], spk)          # deterministic, workspaces reused
        import time
        t0 = time.perf_counter()
        for _ in range(5):
            eng.compute_conditioning([pcm], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
        t_hip = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        Cn.get_conditioning_latents(_golden_sd(dims), [_wav_bytes(g["pcm16"])], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
        t_cpu = time.perf_counter() - t0
        print(f"6 s reference clip: HIP path {1e3 * t_hip:.1f} ms per call (host copies included), PyTorch CPU fp32 {1e3 * t_cpu:.0f} ms")
    finally:
        eng.close()


@pytest.mark.gpu
def test_hip_conditioning_chunks_references_and_normalisation_equal_the_torch_path(dims):
    """Several references, several chunks (the last one shorter), clipping and sound_norm_refs: HIP path vs conditioning.py
    (PyTorch, CPU fp32) on the same inputs (XTTSv2.py:409-468 semantics: embeddings averaged over references, latents over
    the chunks of the concatenated audio)."""
    g = _golden()
    sd = _golden_sd(dims)
    eng = _hip_engine(dims, sd)
    try:
        x = g["pcm16"].astype(np.float32) / 32767.0
        refs = [x[: 22050 * 4 + 1234], 0.5 * x[22050 * 2:]]
        for kw in (dict(max_ref_length=3, gpt_cond_len=5, gpt_cond_chunk_len=2, sound_norm_refs=False),
                   dict(max_ref_length=30, gpt_cond_len=30, gpt_cond_chunk_len=3, sound_norm_refs=True)):
            cond, spk = eng.compute_conditioning(refs, **kw)
            rc, rs = Cn.get_conditioning_latents(sd, [_wav_bytes_f32(r) for r in refs], device="cpu", **kw)
            e_c, e_s = float(np.abs(cond - rc.numpy()).max()), float(np.abs(spk - rs.numpy()).max())
            print(kw, f"latent {e_c:.3e} embedding {e_s:.3e}")
            assert e_c < 2e-4 and e_s < 1e-5
        with pytest.raises(Exception, match="0.33"):
            eng.compute_conditioning([x[:5000]], max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6)
    finally:
        eng.close()
