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
