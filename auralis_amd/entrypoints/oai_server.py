"""OpenAI-compatible HTTP shim over TTS.generate_speech_async — same route, request fields and CLI flags as the
reference server (src/auralis/entrypoints/oai_server.py:65-93, 224-247; request model
src/auralis/common/definitions/openai.py:111-164).  No arithmetic lives here.

    python -m auralis_amd.entrypoints.oai_server --model /path/to/checkpoint_dir --port 8000

`voice` carries base64 files exactly like the reference: reference audio (RIFF/WAVE or FLAC natively, other containers
through torchaudio / ffmpeg when present), or an .npz with precomputed conditioning.  `response_format`: wav, pcm, flac are
built in; mp3 / opus / aac need torchaudio or an ffmpeg executable (api/codecs.py) and otherwise answer the reference's
500 envelope naming the missing back-end.  The default here is wav (the reference's is mp3).  `/v1/chat/completions` is the reference's voice-chat proxy (oai_server.py:95-222, request model
common/definitions/openai.py:16-108): the request is forwarded to an upstream OpenAI-compatible `openai_api_url` with the caller's
Bearer key, the streamed text deltas are passed through (modality "text") and vocalised every `vocalize_at_every_n_words`
words (modality "audio": `audio.chunk` events carrying base64 WAV), then the remainder, a stop chunk and `[DONE]`."""
from __future__ import annotations

import argparse
import asyncio
import base64
import json
import uuid
from typing import Any, Dict, List, Literal, Optional

from fastapi import FastAPI, Header, HTTPException, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse
from pydantic import BaseModel, Field, field_validator

from ..api.requests import TTSRequest
from ..api.tts import TTS

_MEDIA = {"wav": "audio/wav", "pcm": "audio/pcm", "mp3": "audio/mpeg", "opus": "audio/opus", "aac": "audio/aac",
          "flac": "audio/flac"}


class AudioSpeechGenerationRequest(BaseModel):
    input: str = Field(..., description="The textual input to convert")
    model: str = Field(..., description="The model to use for conversion")
    voice: List[str] = Field(..., description="List of base64-encoded reference files")
    response_format: Literal["mp3", "opus", "aac", "flac", "wav", "pcm"] = "wav"
    speed: float = 1.0
    # TTSRequest parameters (defaults of requests.py:164-190)
    enhance_speech: bool = False
    language: str = "auto"
    max_ref_length: int = 60
    gpt_cond_len: int = 30
    gpt_cond_chunk_len: int = 4
    temperature: float = 0.75
    top_p: float = 0.85
    top_k: int = 50
    repetition_penalty: float = 5.0
    length_penalty: float = 1.0
    do_sample: bool = True
    seed: Optional[int] = None

    @field_validator("voice")
    @classmethod
    def _voices_are_base64(cls, v):
        if not v:
            raise ValueError("At least one voice file is required")
        for f in v:
            try:
                base64.b64decode(f, validate=True)
            except Exception:
                raise ValueError("Invalid base64 encoding in voice file")
        return v

    def to_tts_request(self) -> TTSRequest:
        return TTSRequest(text=self.input, stream=False, speaker_files=[base64.b64decode(f) for f in self.voice],
                          enhance_speech=self.enhance_speech, language=self.language,
                          max_ref_length=self.max_ref_length, gpt_cond_len=self.gpt_cond_len,
                          gpt_cond_chunk_len=self.gpt_cond_chunk_len, temperature=self.temperature, top_p=self.top_p,
                          top_k=self.top_k, repetition_penalty=self.repetition_penalty,
                          length_penalty=self.length_penalty, do_sample=self.do_sample, seed=self.seed)


class ChatCompletionMessage(BaseModel):
    role: Literal["system", "user", "assistant"]
    content: str


_TTS_FIELDS = ("enhance_speech", "language", "max_ref_length", "gpt_cond_len", "gpt_cond_chunk_len", "temperature", "top_p",
               "top_k", "repetition_penalty", "length_penalty", "do_sample")


class VoiceChatCompletionRequest(BaseModel):
    """Field for field the reference's model (openai.py:16-108); extra OpenAI fields (max_tokens, ...) are forwarded."""
    model_config = {"extra": "allow"}
    model: str
    messages: List[ChatCompletionMessage]
    speaker_files: List[str] = Field(..., description="List of base64-encoded audio files")
    modalities: List[str] = Field(default=["text", "audio"])   # validated in the route: the reference answers 400, not 422
    openai_api_url: Optional[str] = Field(default=None, validate_default=True)
    vocalize_at_every_n_words: int = Field(default=100, ge=1)
    stream: bool = True
    enhance_speech: bool = False
    language: str = "auto"
    max_ref_length: int = 60
    gpt_cond_len: int = 30
    gpt_cond_chunk_len: int = 4
    temperature: float = 0.75
    top_p: float = 0.85
    top_k: int = 50
    repetition_penalty: float = 5.0
    length_penalty: float = 1.0
    do_sample: bool = True

    @field_validator("openai_api_url")
    @classmethod
    def _url_required(cls, v):
        if v is None:
            raise ValueError("You should always give a url for the text generation")
        return v

    @field_validator("stream")
    @classmethod
    def _stream_only(cls, v):
        if not v:
            raise ValueError("Streaming should be enabled! For non-streaming conversion use the audio endpoint")
        return v

    @field_validator("speaker_files")
    @classmethod
    def _speakers_are_base64(cls, v):
        if not v:
            raise ValueError("At least one speaker file is required")
        for f in v:
            try:
                base64.b64decode(f, validate=True)
            except Exception:
                raise ValueError("Invalid base64 encoding in speaker file")
        return v

    def to_tts_request(self, text: str = "") -> TTSRequest:
        return TTSRequest(text=text, stream=False, speaker_files=[base64.b64decode(f) for f in self.speaker_files],
                          **{k: getattr(self, k) for k in _TTS_FIELDS})

    def to_openai_request(self) -> Dict[str, Any]:
        """What goes upstream: everything except the speech-side fields, always streamed (openai.py:99-108)."""
        drop = {"speaker_files", "openai_api_url", "vocalize_at_every_n_words", "modalities", *_TTS_FIELDS}
        d = {k: v for k, v in self.model_dump().items() if k not in drop}
        d["stream"] = True
        return d


class ClientDisconnected(Exception):
    pass


async def run_unless_disconnected(http_request: Any, awaitable, poll_s: float = 0.1):
    """Await `awaitable` while the HTTP client is still there.  A plain (non-streaming) ASGI handler is not cancelled when its client
    goes away -- the server only queues an `http.disconnect` message -- so the request would be synthesised to the end for nobody
    (the reference does exactly that).  Here the connection is polled (`Request.is_disconnected`) beside the work; when the client has
    gone the work is cancelled, which reaches the engine as aur_cancel for every chunk that is still queued or decoding
    (api/scheduler.py, api/driver.py).  `http_request` = None: no watcher."""
    task = asyncio.ensure_future(awaitable)
    if http_request is None:
        return await task
    try:
        while True:
            done, _ = await asyncio.wait({task}, timeout=poll_s)
            if done:
                return task.result()
            if await http_request.is_disconnected():
                raise ClientDisconnected()
    finally:
        if not task.done():
            task.cancel()


def create_app(tts: Optional[TTS]) -> FastAPI:
    app = FastAPI(title="auralis_amd TTS server")
    app.state.tts = tts

    @app.post("/v1/audio/speech")
    async def generate_audio(request: AudioSpeechGenerationRequest, http_request: Request):
        engine: Optional[TTS] = app.state.tts
        if engine is None or engine.tts_engine is None:
            raise HTTPException(status_code=500, detail="TTS engine not initialized")
        try:
            # the facade owns its own event loop thread: hop over to it and await the result here
            fut = asyncio.run_coroutine_threadsafe(engine.generate_speech_async(request.to_tts_request()), engine._loop)
            try:
                output = await run_unless_disconnected(http_request, asyncio.wrap_future(fut))
            except ClientDisconnected:
                return Response(status_code=499)   # (nobody reads it; nginx's "client closed request")
            if request.speed != 1.0:
                output = output.change_speed(request.speed)
            data = output.to_bytes(request.response_format)
            return Response(content=data, media_type=_MEDIA[request.response_format])
        except Exception as e:  # same envelope as the reference (oai_server.py:92-93)
            return JSONResponse(status_code=500, content={"error": f"Error generating audio: {e}"})

    async def _speak(engine: TTS, req: TTSRequest):
        fut = asyncio.run_coroutine_threadsafe(engine.generate_speech_async(req), engine._loop)
        return await asyncio.wrap_future(fut)

    @app.post("/v1/chat/completions")
    async def chat_completions(request: VoiceChatCompletionRequest, authorization: Optional[str] = Header(None)):
        engine: Optional[TTS] = app.state.tts
        if engine is None or engine.tts_engine is None:
            raise HTTPException(status_code=500, detail="TTS engine not initialized")
        if not authorization or not authorization.startswith("Bearer "):
            return JSONResponse(status_code=400, content={"error": "Authorization header with Bearer token is required"})
        valid = ["text", "audio"]
        if not all(m in valid for m in request.modalities):
            return JSONResponse(status_code=400, content={"error": f"Invalid modalities. Must be one or more of {valid}"})
        try:
            import aiohttp
            modalities, every_n = request.modalities, request.vocalize_at_every_n_words
            upstream_body = request.to_openai_request()
            headers = {"Content-Type": "application/json", "Authorization": authorization}
            request_id = uuid.uuid4().hex

            async def audio_event(text: str) -> str:
                req = request.to_tts_request(text=text)   # (language "auto" is resolved from this piece of text)
                out = await _speak(engine, req)
                b64 = base64.b64encode(out.to_bytes()).decode("utf-8")
                return f"data: {json.dumps({'id': request_id, 'object': 'audio.chunk', 'data': b64})}\n\n"

            async def stream_generator():
                pending = ""
                try:
                    async with aiohttp.ClientSession() as session:
                        async with session.post(request.openai_api_url, json=upstream_body, headers=headers) as resp:
                            if resp.status != 200:
                                raise HTTPException(status_code=resp.status, detail=await resp.text())
                            async for raw in resp.content:
                                line = raw.decode("utf-8").strip() if raw else ""
                                if not line.startswith("data:"):
                                    continue
                                payload = line[5:].strip()
                                if payload == "[DONE]":
                                    break
                                try:
                                    data = json.loads(payload)
                                except json.JSONDecodeError:
                                    continue
                                content = (data.get("choices") or [{}])[0].get("delta", {}).get("content", "")
                                if content:
                                    pending += content
                                    if "text" in modalities:
                                        yield f"data: {json.dumps(data)}\n\n"
                                    if len(pending.split()) >= every_n:
                                        if "audio" in modalities:
                                            yield await audio_event(pending)
                                        pending = ""
                                elif "text" in modalities:
                                    yield f"data: {json.dumps(data)}\n\n"
                    if pending and "audio" in modalities:
                        yield await audio_event(pending)
                    if "text" in modalities:
                        stop = {"id": request_id, "object": "chat.completion.chunk",
                                "choices": [{"delta": {}, "index": 0, "finish_reason": "stop"}]}
                        yield f"data: {json.dumps(stop)}\n\n"
                    yield "data: [DONE]\n\n"
                except Exception as e:   # same envelope as the reference: the error travels inside the event stream
                    detail = getattr(e, "detail", None) or str(e)
                    yield f"data: {json.dumps({'error': str(detail)})}\n\n"

            return StreamingResponse(stream_generator(), media_type="text/event-stream")
        except Exception as e:
            return JSONResponse(status_code=500, content={"error": f"Error in chat completions: {e}"})

    @app.get("/health")
    async def health():
        return {"status": "ok" if app.state.tts is not None else "no-engine"}

    return app


def main():
    import uvicorn
    p = argparse.ArgumentParser(description="auralis_amd TTS FastAPI server")
    p.add_argument("--host", type=str, default="127.0.0.1")
    p.add_argument("--port", type=int, default=8000)
    p.add_argument("--model", type=str, required=True, help="checkpoint directory in the reference's on-disk format")
    p.add_argument("--max_concurrency", type=int, default=8)
    p.add_argument("--vllm_logging_level", type=str, default="warn", help="accepted for CLI compatibility; unused")
    a = p.parse_args()
    tts = TTS(scheduler_max_concurrency=a.max_concurrency).from_pretrained(a.model)
    uvicorn.run(create_app(tts), host=a.host, port=a.port)


if __name__ == "__main__":
    main()
