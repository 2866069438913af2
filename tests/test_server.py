"""CPU: OpenAI-compatible /v1/audio/speech shim over the facade, with a fake engine."""
import base64
import io

import numpy as np
import pytest

from auralis_amd import TTS, TTSOutput
from auralis_amd.api.text import XTTSTokenizer
from auralis_amd.api.xtts_engine import XTTSv2Engine
from auralis_amd.entrypoints.oai_server import create_app
from tests.fakes import FakeNativeEngine

fastapi_testclient = pytest.importorskip("fastapi.testclient")


def _npz_voice():
    buf = io.BytesIO()
    np.savez(buf, gpt_cond_latent=np.zeros((1, 32, 1024), np.float32), speaker_embedding=np.ones((1, 512, 1), np.float32))
    return base64.b64encode(buf.getvalue()).decode()


def test_speech_endpoint_roundtrip(tmp_path):
    tts = TTS(scheduler_max_concurrency=2).with_engine(XTTSv2Engine(FakeNativeEngine(max_seqs=2), XTTSTokenizer(None, synthetic=True)))
    try:
        client = fastapi_testclient.TestClient(create_app(tts))
        r = client.post("/v1/audio/speech", json={"input": "Hello from the server side of things.", "model": "xtts",
                                                    "voice": [_npz_voice()], "response_format": "wav", "language": "en"})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
        p = tmp_path / "o.wav"
        p.write_bytes(r.content)
        out = TTSOutput.from_file(p)
        assert out.sample_rate == 24000 and len(out.array) > 0
        r = client.post("/v1/audio/speech", json={"input": "x", "model": "xtts", "voice": ["@@not-base64@@"]})
        assert r.status_code == 422
        r = client.post("/v1/audio/speech", json={"input": "hi", "model": "xtts", "voice": [_npz_voice()],
                                                    "response_format": "mp3", "language": "en"})
        assert r.status_code == 500 and "error" in r.json()
        assert client.post("/v1/chat/completions", json={}).status_code == 501
        assert client.get("/health").json()["status"] == "ok"
    finally:
        tts.close()


def test_speech_endpoint_without_engine():
    client = fastapi_testclient.TestClient(create_app(None))
    r = client.post("/v1/audio/speech", json={"input": "hi", "model": "xtts", "voice": [_npz_voice()]})
    assert r.status_code == 500
