"""GPU, 30 layers: the FACADE against the oracle -- text in, waveform out.

`TTS.from_pretrained(dir).generate_speech(TTSRequest(text=...))` (the surface a user of the reference calls, core/tts.py:310-355)
is held to the CPU restatement of the reference's per-request recipe (oracle/make_golden_facade.py -> tests/golden/facade_L30.npz:
language detection -> split_sentence -> tokenizer -> per-chunk generation -> literal second pass -> HiFi-GAN -> combine_outputs):

  * BASELINE configs[0]: the 50-character English sentence, greedy;
  * three ~500-character paragraphs in English, French and German, language="auto", sampled with the request defaults under a fixed
    seed, non-streaming and streaming.

Per request: detected language, chunk texts and chunk ids as stored (the facade chunked and encoded like the oracle leg), the mel ids
of EVERY chunk bit-exact, chunks in order, waveform within 1e-3 RMS of the fp32 CPU path (north_star tolerance); stream == non-stream
bit for bit.  Checkpoint: the fixture's seeded synthetic directory written in the reference's on-disk format (gpt_max_audio_tokens in
its config, stop id reachable)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "facade_L30.npz")


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, np.float64)))))


@pytest.fixture(scope="module")
def facade(tmp_path_factory, dims):
    from auralis_amd import TTS
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    g = np.load(GOLD)
    root = str(tmp_path_factory.mktemp("facade_ckpt"))
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=int(g["n_layer"]))
    gpt_sd["mel_head.bias"][1025] = float(g["stop_bias"])
    save_checkpoint(root, gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True,
                    gpt_max_audio_tokens=int(g["max_tokens"]))
    cond, spk = make_synthetic_conditioning(dims)
    voice = {"gpt_cond_latent": cond.numpy(), "speaker_embedding": spk.numpy()}
    tts = TTS(scheduler_max_concurrency=8).from_pretrained(root)
    # what the plugin hands to the engine, per request: (chunk ids, the chunk's result future)
    eng = tts.tts_engine
    log = []
    orig_ctx, orig_split = eng.get_generation_context, eng.tokenizer.batch_encode_with_split

    def spy_split(text, lang):
        out = orig_split(text, lang)
        log.append({"language": lang, "ids": [list(map(int, c)) for c in out], "handles": None})
        return out

    async def spy_ctx(request, **kw):
        out = await orig_ctx(request, **kw)
        log[-1]["handles"] = list(out[0])
        return out
    eng.tokenizer.batch_encode_with_split = spy_split
    eng.get_generation_context = spy_ctx
    yield tts, voice, g, log
    tts.close()


def _request(g, name, voice, stream):
    from auralis_amd import TTSRequest
    kw = dict(text=str(g[f"{name}_text"]), speaker_files=[voice], language="auto", seed=int(g[f"{name}_seed"]), stream=stream)
    if float(g[f"{name}_temperature"]) == 0.0:
        kw["temperature"] = 0.0
    return TTSRequest(**kw)


def _check(g, name, rec, wav, chunks=None):
    n = int(g[f"{name}_n_chunks"])
    assert rec["language"] == str(g[f"{name}_language"])
    assert len(rec["ids"]) == n, (name, len(rec["ids"]), n)
    off = 0
    for i in range(n):
        assert rec["ids"][i] == g[f"{name}_ids_{i}"].tolist(), (name, i, "the facade encoded this chunk differently from the oracle leg")
        got = rec["handles"][i].future.result()["tokens"].tolist()
        want = g[f"{name}_tokens_{i}"].tolist()
        d = next((k for k, (a, b) in enumerate(zip(got, want)) if a != b), None if len(got) == len(want) else min(len(got), len(want)))
        assert d is None, f"{name} chunk {i}: mel ids differ at step {d} (got {len(got)} ids, oracle {len(want)})"
        if chunks is not None:   # streamed: chunk i arrives i-th and is chunk i's audio
            assert chunks[i].token_length == len(want)
            off += len(chunks[i].array)
    ref = g[f"{name}_wav"]
    assert wav.shape == ref.shape, (name, wav.shape, ref.shape)
    err = rms(wav - ref)
    assert err <= 1e-3 and err <= 0.01 * max(rms(ref), 1e-9) + 1e-6, (name, err, rms(ref))
    return err


def test_c1_fifty_char_sentence_through_the_facade(facade):
    tts, voice, g, log = facade
    from auralis_amd import TTSOutput
    assert len(str(g["c1_text"])) == 50 and int(g["c1_n_chunks"]) == 1 and str(g["c1_language"]) == "en"
    out = tts.generate_speech(_request(g, "c1", voice, stream=False))
    assert isinstance(out, TTSOutput) and out.sample_rate == 24000
    _check(g, "c1", log[-1], out.array)


@pytest.mark.parametrize("name,lang", [("en", "en"), ("fr", "fr"), ("de", "de")])
def test_auto_language_paragraph_through_the_facade_stream_and_not(facade, name, lang):
    tts, voice, g, log = facade
    assert str(g[f"{name}_language"]) == lang and int(g[f"{name}_n_chunks"]) >= 2
    whole = tts.generate_speech(_request(g, name, voice, stream=False))
    _check(g, name, log[-1], whole.array)
    chunks = list(tts.generate_speech(_request(g, name, voice, stream=True)))
    assert len(chunks) == int(g[f"{name}_n_chunks"])
    streamed = np.concatenate([c.array for c in chunks])
    _check(g, name, log[-1], streamed, chunks)
    assert np.array_equal(streamed, whole.array)   # same seed: the stream is the non-stream output, chunk by chunk


def test_c1_runs_to_its_natural_stop_at_the_full_token_budget(tmp_path_factory, dims):
    """VERDICT r05, thin spot: the other facade goldens cap generation at 96 tokens.  Here BASELINE configs[0] -- the 50-character
    sentence, greedy -- runs the way a user runs it: gpt_max_audio_tokens = 605 (the reference's own default) in the checkpoint's
    config and nothing but the stop id to end it.  The CPU oracle generates 237 ids and the stop id
    (oracle/make_golden_facade.py --c1-natural -> tests/golden/facade_c1_natural_L30.npz); the facade on the HIP engine must emit the
    same ids, stop at the same step, and deliver the waveform within 1e-3 RMS."""
    from auralis_amd import TTS, TTSRequest
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    g = np.load(os.path.join(os.path.dirname(GOLD), "facade_c1_natural_L30.npz"))
    assert int(g["max_tokens"]) == 605 and 96 < len(g["tokens"]) < 605 and int(g["tokens"][-1]) == 1025
    root = str(tmp_path_factory.mktemp("c1_natural_ckpt"))
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=int(g["n_layer"]))
    gpt_sd["mel_head.bias"][1025] = float(g["stop_bias"])
    save_checkpoint(root, gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True, gpt_max_audio_tokens=605)
    cond, spk = make_synthetic_conditioning(dims)
    voice = {"gpt_cond_latent": cond.numpy(), "speaker_embedding": spk.numpy()}
    tts = TTS(scheduler_max_concurrency=2).from_pretrained(root)
    try:
        assert tts.tts_engine.gpt_max_audio_tokens == 605
        seen = []
        orig = tts.tts_engine.tokenizer.batch_encode_with_split
        tts.tts_engine.tokenizer.batch_encode_with_split = lambda text, lang: (seen.append(orig(text, lang)), seen[-1])[1]
        handles = []
        orig_ctx = tts.tts_engine.get_generation_context

        async def spy_ctx(request, **kw):
            res = await orig_ctx(request, **kw)
            handles.extend(res[0])
            return res
        tts.tts_engine.get_generation_context = spy_ctx
        out = tts.generate_speech(TTSRequest(text=str(g["text"]), speaker_files=[voice], language="auto", temperature=0.0, seed=int(g["seed"])))
        assert [list(map(int, c)) for c in seen[-1]] == [g["ids"].tolist()]
        got, want = handles[0].future.result()["tokens"].tolist(), g["tokens"].tolist()
        d = next((k for k, (a, b) in enumerate(zip(got, want)) if a != b), None if len(got) == len(want) else min(len(got), len(want)))
        assert d is None, f"mel ids differ at step {d} (got {len(got)} ids, oracle {len(want)})"
        assert out.token_length == len(g["tokens"]), (out.token_length, len(g["tokens"]))       # stopped at the oracle's step
        assert out.array.shape == g["wav"].shape
        err = rms(out.array - g["wav"])
        assert err <= 1e-3 and err <= 0.01 * max(rms(g["wav"]), 1e-9) + 1e-6, (err, rms(g["wav"]))
    finally:
        tts.close()
