"""GPU: the OpenAI-compatible HTTP route on the REAL engine (reference: src/auralis/entrypoints/oai_server.py:65-93).

FastAPI TestClient -> POST /v1/audio/speech -> TTS.generate_speech_async -> XTTSv2Engine plugin -> EngineDriver -> C ABI -> HIP
kernels; the bytes that come back are decoded per `response_format` and must be the int16 quantisation of what
`tts.generate_speech(TTSRequest(...))` returns for the same request (same text, voice, sampling parameters and seed): the HTTP layer,
the base64 voice transport and the codecs add nothing and lose nothing.  wav / pcm / flac are the built-in (lossless) formats; one
multi-chunk request checks that the route's combine_outputs path delivers the chunks in order."""
import base64
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
fastapi_testclient = pytest.importorskip("fastapi.testclient")

TEXT = "The harbour lights came on one by one while the last ferry crossed the bay."
LONG = ("It was a bright cold day in April, and the clocks were striking thirteen. Nobody in the street seemed to notice, and the wind "
        "kept pushing the dust along the old road as if nothing had happened at all. Who would have thought that such a thing could "
        "happen? Nobody, really; yet here we are, walking slowly along the old road, counting the stones and the years that went by.")


@pytest.fixture(scope="module")
def served(tmp_path_factory, dims):
    from auralis_amd import TTS
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    from auralis_amd.entrypoints.oai_server import create_app
    root = str(tmp_path_factory.mktemp("server_ckpt"))
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 3.0      # natural stop after a handful of tokens
    save_checkpoint(root, gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True)
    cond, spk = make_synthetic_conditioning(dims)
    buf = io.BytesIO()
    np.savez(buf, gpt_cond_latent=cond.numpy(), speaker_embedding=spk.numpy())
    voice_bytes = buf.getvalue()
    tts = TTS(scheduler_max_concurrency=4).from_pretrained(root)
    client = fastapi_testclient.TestClient(create_app(tts))
    yield tts, client, voice_bytes
    tts.close()


def _int16(x):
    return (np.clip(np.asarray(x, np.float32), -1.0, 1.0) * 32767.0).astype(np.int16)


def _decode(fmt, data):
    from auralis_amd.api import flac
    if fmt == "pcm":
        return np.frombuffer(data, dtype="<i2")
    if fmt == "wav":
        import wave
        with wave.open(io.BytesIO(data), "rb") as w:
            assert (w.getframerate(), w.getnchannels(), w.getsampwidth()) == (24000, 1, 2)
            return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    x, sr, bps = flac.decode(data)
    assert (sr, bps) == (24000, 16)
    return x.reshape(-1).astype(np.int16)


@pytest.mark.parametrize("fmt,media", [("wav", "audio/wav"), ("pcm", "audio/pcm"), ("flac", "audio/flac")])
def test_speech_route_returns_what_generate_speech_returns(served, fmt, media):
    from auralis_amd import TTSRequest
    tts, client, voice = served
    kw = dict(temperature=0.75, top_p=0.85, top_k=50, repetition_penalty=5.0, seed=11, language="en")
    r = client.post("/v1/audio/speech", json={"input": TEXT, "model": "xtts", "voice": [base64.b64encode(voice).decode()],
                                                "response_format": fmt, **kw})
    assert r.status_code == 200, r.text
    assert r.headers["content-type"] == media
    got = _decode(fmt, r.content)
    direct = tts.generate_speech(TTSRequest(text=TEXT, speaker_files=[voice], **kw))
    want = _int16(direct.array)
    assert len(want) > 0 and got.shape == want.shape, (fmt, got.shape, want.shape)
    assert np.array_equal(got, want), f"{fmt}: {int(np.count_nonzero(got != want))} of {len(want)} samples differ from generate_speech()"
    assert np.abs(want).max() > 0   # not silence


def test_speech_route_multi_chunk_request_and_streaming_consumer(served):
    """A text above the 250-character limit becomes several chunks; the route answers their concatenation in order (TTSOutput.
    combine_outputs), and a streaming generate_speech() of the same request yields those chunks one by one."""
    from auralis_amd import TTSRequest
    tts, client, voice = served
    kw = dict(temperature=0.75, top_p=0.85, top_k=50, repetition_penalty=5.0, seed=23, language="en")
    r = client.post("/v1/audio/speech", json={"input": LONG, "model": "xtts", "voice": [base64.b64encode(voice).decode()],
                                                "response_format": "pcm", **kw})
    assert r.status_code == 200, r.text
    got = np.frombuffer(r.content, dtype="<i2")
    chunks = list(tts.generate_speech(TTSRequest(text=LONG, speaker_files=[voice], stream=True, **kw)))
    assert len(chunks) >= 2
    want = _int16(np.concatenate([c.array for c in chunks]))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_speech_route_error_envelope_on_the_real_engine(served):
    """An unusable voice answers the reference's 500 envelope (oai_server.py:92-93) and the engine serves the next request."""
    tts, client, voice = served
    r = client.post("/v1/audio/speech", json={"input": TEXT, "model": "xtts", "voice": [base64.b64encode(b"not audio at all").decode()],
                                                "response_format": "wav", "language": "en"})
    assert r.status_code == 500 and "Error generating audio" in r.json()["error"]
    r = client.post("/v1/audio/speech", json={"input": "Still here.", "model": "xtts", "voice": [base64.b64encode(voice).decode()],
                                                "response_format": "wav", "language": "en", "seed": 1})
    assert r.status_code == 200 and len(r.content) > 44
