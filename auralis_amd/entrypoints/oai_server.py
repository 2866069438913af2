"""OpenAI-compatible HTTP shim over TTS.generate_speech_async — same route, request fields and CLI flags as the
reference server (src/auralis/entrypoints/oai_server.py:65-93, 224-247; request model
src/auralis/common/definitions/openai.py:111-164).  No arithmetic lives here.

    python -m auralis_amd.entrypoints.oai_server --model /path/to/checkpoint_dir --port 8000

`voice` carries base64 files exactly like the reference: RIFF/WAVE reference audio, or an .npz with precomputed
conditioning.  `/v1/chat/completions` (a proxy that TTS-es an upstream LLM stream, oai_server.py:95-222) is outside the
synthesis path and answers 501."""
from __future__ import annotations

import argparse
import base64
from typing import List, Literal, Optional

from fastapi import FastAPI, HTTPException
from fastapi.responses import JSONResponse, Response
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


def create_app(tts: Optional[TTS]) -> FastAPI:
    app = FastAPI(title="auralis_amd TTS server")
    app.state.tts = tts

    @app.post("/v1/audio/speech")
    async def generate_audio(request: AudioSpeechGenerationRequest):
        engine: Optional[TTS] = app.state.tts
        if engine is None or engine.tts_engine is None:
            raise HTTPException(status_code=500, detail="TTS engine not initialized")
        try:
            import asyncio
            # the facade owns its own event loop thread: hop over to it and await the result here
            fut = asyncio.run_coroutine_threadsafe(engine.generate_speech_async(request.to_tts_request()), engine._loop)
            output = await asyncio.wrap_future(fut)
            if request.speed != 1.0:
                output = output.change_speed(request.speed)
            data = output.to_bytes(request.response_format)
            return Response(content=data, media_type=_MEDIA[request.response_format])
        except Exception as e:  # same envelope as the reference (oai_server.py:92-93)
            return JSONResponse(status_code=500, content={"error": f"Error generating audio: {e}"})

    @app.post("/v1/chat/completions")
    async def chat_completions():
        return JSONResponse(status_code=501, content={"error": "chat-completions proxy is outside the synthesis path"})

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
