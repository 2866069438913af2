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
                                                    "response_format": "flac", "language": "en"})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/flac" and r.content[:4] == b"fLaC"
        from auralis_amd.api import codecs
        r = client.post("/v1/audio/speech", json={"input": "hi", "model": "xtts", "voice": [_npz_voice()],
                                                    "response_format": "mp3", "language": "en"})
        if codecs.external_backend() is None:   # lossy codecs need the reference's own back-end (torchaudio / ffmpeg)
            assert r.status_code == 500 and "ffmpeg" in r.json()["error"]
        else:
            assert r.status_code == 200
        assert client.get("/health").json()["status"] == "ok"
    finally:
        tts.close()


def test_speech_endpoint_without_engine():
    client = fastapi_testclient.TestClient(create_app(None))
    r = client.post("/v1/audio/speech", json={"input": "hi", "model": "xtts", "voice": [_npz_voice()]})
    assert r.status_code == 500


def _upstream_app(deltas):
    """A stand-in for the upstream OpenAI-compatible endpoint: checks the forwarded request, streams `deltas` as SSE."""
    from aiohttp import web
    seen = {}

    async def handler(request):
        seen["auth"] = request.headers.get("Authorization")
        seen["body"] = await request.json()
        resp = web.StreamResponse(status=200, headers={"Content-Type": "text/event-stream"})
        await resp.prepare(request)
        await resp.write(b": keep-alive comment\n\n")
        await resp.write(b'data: {"choices": [{"delta": {"role": "assistant"}}]}\n\n')
        for d in deltas:
            import json
            await resp.write(("data: " + json.dumps({"choices": [{"delta": {"content": d}, "index": 0}]}) + "\n\n").encode())
        await resp.write(b"data: not-json\n\n")
        await resp.write(b"data: [DONE]\n\n")
        await resp.write_eof()
        return resp

    async def refuse(request):
        return web.Response(status=401, text="bad key")

    app = web.Application()
    app.router.add_post("/v1/chat/completions", handler)
    app.router.add_post("/refuse", refuse)
    return app, seen


def test_chat_completions_proxy_streams_text_and_audio():
    """The voice-chat proxy (reference oai_server.py:95-222): text deltas passed through, audio every n words + remainder,
    stop chunk and [DONE]; Bearer key and the non-speech fields forwarded upstream; speech-side fields kept out of it."""
    import asyncio
    import json
    import threading

    from aiohttp import web
    deltas = ["Hello there, ", "this is the first ", "part of the answer. ", "And here ", "comes the rest."]
    up, seen = _upstream_app(deltas)
    loop = asyncio.new_event_loop()
    runner = web.AppRunner(up)
    loop.run_until_complete(runner.setup())
    site = web.TCPSite(runner, "127.0.0.1", 0)
    loop.run_until_complete(site.start())
    port = runner.addresses[0][1]
    th = threading.Thread(target=loop.run_forever, daemon=True)
    th.start()
    fake = FakeNativeEngine(max_seqs=2)
    tts = TTS(scheduler_max_concurrency=2).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    try:
        client = fastapi_testclient.TestClient(create_app(tts))
        body = {"model": "some-llm", "messages": [{"role": "user", "content": "hi"}], "speaker_files": [_npz_voice()],
                "openai_api_url": f"http://127.0.0.1:{port}/v1/chat/completions", "vocalize_at_every_n_words": 6,
                "language": "en", "max_tokens": 64}
        assert client.post("/v1/chat/completions", json=body).status_code == 400            # no Bearer token
        hdr = {"Authorization": "Bearer sk-test"}
        assert client.post("/v1/chat/completions", json={**body, "modalities": ["video"]}, headers=hdr).status_code == 400
        assert client.post("/v1/chat/completions", json={**body, "stream": False}, headers=hdr).status_code == 422
        no_url = {k: v for k, v in body.items() if k != "openai_api_url"}
        assert client.post("/v1/chat/completions", json=no_url, headers=hdr).status_code == 422

        r = client.post("/v1/chat/completions", json=body, headers=hdr)
        assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
        events = [ln[5:].strip() for ln in r.text.splitlines() if ln.startswith("data:")]
        assert events[-1] == "[DONE]"
        objs = [json.loads(e) for e in events[:-1]]
        text = "".join(o["choices"][0]["delta"].get("content", "") for o in objs if "choices" in o)
        assert text == "".join(deltas)
        audio = [o for o in objs if o.get("object") == "audio.chunk"]
        # 6-word threshold: after delta 2 (6 words), after delta 4 (7 words since), and the remainder -> three audio events
        assert len(audio) == 3 and len(fake.submitted) == 3
        for o in audio:
            wav = base64.b64decode(o["data"])
            assert wav[:4] == b"RIFF" and len(wav) > 44
        assert objs[-1]["choices"][0]["finish_reason"] == "stop"
        # an audio event follows the text delta that crossed the threshold
        kinds = ["a" if o.get("object") == "audio.chunk" else "t" for o in objs]
        assert kinds.index("a") > 0 and kinds[kinds.index("a") - 1] == "t"
        # what went upstream
        assert seen["auth"] == "Bearer sk-test" and seen["body"]["stream"] is True and seen["body"]["max_tokens"] == 64
        assert seen["body"]["model"] == "some-llm" and seen["body"]["messages"][0]["content"] == "hi"
        for k in ("speaker_files", "openai_api_url", "vocalize_at_every_n_words", "modalities", "temperature", "language"):
            assert k not in seen["body"]

        r = client.post("/v1/chat/completions", json={**body, "modalities": ["audio"]}, headers=hdr)
        objs = [json.loads(ln[5:]) for ln in r.text.splitlines() if ln.startswith("data:") and "[DONE]" not in ln]
        assert objs and all(o.get("object") == "audio.chunk" for o in objs)                  # audio only: no text, no stop chunk

        r = client.post("/v1/chat/completions", json={**body, "openai_api_url": f"http://127.0.0.1:{port}/refuse"}, headers=hdr)
        assert r.status_code == 200 and "bad key" in r.text and "[DONE]" not in r.text       # upstream error inside the stream
    finally:
        tts.close()
        loop.call_soon_threadsafe(loop.stop)
        th.join(timeout=5)


def test_speech_request_of_a_client_that_disconnected_is_cancelled_in_the_engine():
    """POST /v1/audio/speech is a plain response: the ASGI server does not cancel its handler when the client goes away.  The route
    polls the connection beside the work (`run_unless_disconnected`); here a long request (a dozen chunks on a one-slot engine that
    takes 20 ms per step) whose client reports a disconnect after the second poll: the handler gives up, the chunks still queued or
    decoding are cancelled in the engine, and the next request is served."""
    import asyncio
    import time

    from auralis_amd import TTSRequest
    from auralis_amd.entrypoints.oai_server import ClientDisconnected, run_unless_disconnected
    fake = FakeNativeEngine(max_seqs=1, step_delay=0.02)
    tts = TTS(scheduler_max_concurrency=2).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    voice = {"gpt_cond_latent": np.zeros((1, 32, 1024), np.float32), "speaker_embedding": np.ones((1, 512, 1), np.float32)}
    text = " ".join(["It was a bright cold day in April, and the clocks were striking thirteen. Nobody in the street seemed to notice, "
                     "and the wind kept pushing the dust along the old road as if nothing had happened at all."] * 12)

    class Gone:
        polls = 0

        async def is_disconnected(self):
            Gone.polls += 1
            return Gone.polls >= 2

    async def handler():
        fut = asyncio.run_coroutine_threadsafe(tts.generate_speech_async(TTSRequest(text=text, speaker_files=[voice], language="en")), tts._loop)
        return await run_unless_disconnected(Gone(), asyncio.wrap_future(fut), poll_s=0.05)
    try:
        with pytest.raises(ClientDisconnected):
            asyncio.run(handler())
        n = len(fake.submitted)
        assert n >= 10
        t0 = time.time()
        while (fake.finished_total < n or tts.tts_engine.driver._pending) and time.time() - t0 < 10:
            time.sleep(0.01)
        assert len(fake.cancelled) >= n - 5 and not fake.waiting and not fake.running, (n, fake.cancelled)

        async def served():   # a client that stays: the same helper returns the result
            class Here:
                async def is_disconnected(self):
                    return False
            fut = asyncio.run_coroutine_threadsafe(tts.generate_speech_async(TTSRequest(text="Still here.", speaker_files=[voice], language="en")), tts._loop)
            return await run_unless_disconnected(Here(), asyncio.wrap_future(fut), poll_s=0.01)
        assert len(asyncio.run(served()).array) > 0
    finally:
        tts.close()
