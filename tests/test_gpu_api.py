"""GPU: the reference-shaped Python surface (TTS().from_pretrained(dir).generate_speech(request)) on the real HIP
engine, loading a checkpoint directory written in the reference's on-disk format."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tts_facade_end_to_end(tmp_path, dims):
    from auralis_amd import TTS, TTSOutput, TTSRequest
    from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts,
                                        save_checkpoint)
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    sd = {k: v.clone() for k, v in gpt_sd.items()}
    sd["mel_head.bias"][1025] = 3.0      # let sequences stop naturally after a handful of tokens
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=sd)
    save_checkpoint(str(tmp_path), sd, xtts_sd, dims, synthetic_tokenizer=True)
    cond, spk = make_synthetic_conditioning(dims)
    voice = {"gpt_cond_latent": cond.numpy(), "speaker_embedding": spk.numpy()}
    text = ("The quick brown fox jumps over the lazy dog near the river bank. It was a bright cold day in April, and "
            "the clocks were striking thirteen! Who would have thought that such a thing could happen? Nobody, really; "
            "yet here we are, walking slowly along the old road, counting the stones and the years that went by. ") * 3
    tts = TTS(scheduler_max_concurrency=4).from_pretrained(str(tmp_path))
    try:
        req = TTSRequest(text=text, speaker_files=[voice], language="en", temperature=0.0, seed=5)
        out = tts.generate_speech(req)
        assert isinstance(out, TTSOutput) and out.sample_rate == 24000
        assert len(out.array) > 0 and np.isfinite(out.array).all() and np.abs(out.array).max() <= 1.0
        chunks = list(tts.generate_speech(TTSRequest(text=text, speaker_files=[voice], language="en", temperature=0.0,
                                                      seed=5, stream=True)))
        assert len(chunks) >= 3
        assert np.array_equal(np.concatenate([c.array for c in chunks]), out.array)   # greedy: stream == non-stream
        assert len(out.to_bytes("wav")) > 44
    finally:
        tts.close()


def test_voice_cloning_from_reference_audio_end_to_end(tmp_path, dims):
    """speaker_files = reference AUDIO (FLAC and WAV bytes of the same real-speech clip): the facade decodes it on the host,
    the HIP conditioning path (aur_compute_conditioning) turns it into latents + embedding once, and synthesis runs with them."""
    import os

    from auralis_amd import TTS, TTSRequest
    from auralis_amd.api import codecs, flac
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cond_female_6s.npz"))
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 3.0
    save_checkpoint(str(tmp_path), gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True)
    as_flac = flac.encode(g["pcm16"], 22050, 16)
    as_wav = codecs.wav_bytes(g["pcm16"].astype(np.float32) / 32767.0, 22050)
    tts = TTS(scheduler_max_concurrency=2).from_pretrained(str(tmp_path))
    try:
        eng = tts.tts_engine
        assert eng.conditioning_weights is not None and hasattr(eng.native, "compute_conditioning")
        outs = []
        for ref in (as_flac, as_wav):
            req = TTSRequest(text="Hello there, this voice was cloned from six seconds of speech.", speaker_files=[ref],
                             language="en", temperature=0.0, seed=1)
            outs.append(tts.generate_speech(req))
        assert len(outs[0].array) > 0 and np.isfinite(outs[0].array).all()
        # 16-bit FLAC and 16-bit WAV decode to the same samples up to the int16 scale convention (32767 vs 32768)
        assert len(eng._cond_cache) == 2
        (g1, s1), (g2, s2) = eng._cond_cache.values()
        assert g1.shape == (1, 32, 1024) and np.abs(g1 - g2).max() < 1e-3 and np.abs(s1 - s2).max() < 1e-4
        direct = eng.native.compute_conditioning([codecs.decode(as_flac)[0]], max_ref_length=60, gpt_cond_len=30, gpt_cond_chunk_len=4)
        assert np.array_equal(direct[0], g1) and np.array_equal(direct[1], s1)
    finally:
        tts.close()
