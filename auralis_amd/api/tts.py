"""TTS facade — same public surface as the reference (src/auralis/core/tts.py:20-362): TTS(scheduler_max_concurrency),
from_pretrained, generate_speech (sync; generator when request.stream), generate_speech_async, split_requests,
prepare_for_streaming_generation, shutdown.  Engines are picked from MODEL_REGISTRY by config.json["model_type"]
(core/tts.py:85) and driven through the two plugin calls get_generation_context / process_tokens_to_speech."""
from __future__ import annotations

import asyncio
import json
import os
import threading
import time
import uuid
from functools import partial
from typing import AsyncGenerator, Dict, Generator, List, Optional, Union

from .engine_base import MODEL_REGISTRY, BaseAsyncTTSEngine
from .output import TTSOutput
from .requests import TTSRequest
from .scheduler import TwoPhaseScheduler


class TTS:
    def __init__(self, scheduler_max_concurrency: int = 10, vllm_logging_level=None):
        self.scheduler_max_concurrency = scheduler_max_concurrency
        # the same bound the reference passes (core/tts.py:20-51: TwoPhaseScheduler(scheduler_max_concurrency)); the engine gets the
        # same number as its slot count (the reference hands it to vLLM's max_num_seqs, XTTSv2.py:235-243).  Phase 1 submits every
        # chunk of a request to the engine at once, so what the engine cannot seat waits in ITS queue -- which is where grouped
        # admission (aur_config.admit_min_batch) takes its groups from -- whatever the width of this gate.
        self.scheduler: Optional[TwoPhaseScheduler] = TwoPhaseScheduler(max(1, int(scheduler_max_concurrency)))
        self.tts_engine: Optional[BaseAsyncTTSEngine] = None
        self.concurrency = scheduler_max_concurrency
        self.max_vllm_memory = None
        self._loop = asyncio.new_event_loop()
        self._thread = threading.Thread(target=self._loop.run_forever, name="auralis-amd-loop", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ model loading
    def from_pretrained(self, model_name_or_path: str, **kwargs) -> "TTS":
        cfg_path = os.path.join(model_name_or_path, "config.json")
        if not os.path.isfile(cfg_path):
            cfg_path = os.path.join(model_name_or_path, "core_xttsv2", "config.json")
        if not os.path.isfile(cfg_path):
            raise FileNotFoundError(f"no config.json under {model_name_or_path} (no network: local checkpoints only)")
        with open(cfg_path) as f:
            model_type = json.load(f)["model_type"]
        from . import xtts_engine  # noqa: F401  (registers "xtts")
        if model_type not in MODEL_REGISTRY:
            raise ValueError(f"Could not load model '{model_type}': not in MODEL_REGISTRY {sorted(MODEL_REGISTRY)}")
        kwargs.setdefault("max_concurrency", self.scheduler_max_concurrency)
        self.tts_engine = MODEL_REGISTRY[model_type].from_pretrained(model_name_or_path, **kwargs)
        return self

    def with_engine(self, engine: BaseAsyncTTSEngine) -> "TTS":
        """Attach an already constructed engine plugin (tests, custom engines)."""
        self.tts_engine = engine
        return self

    # ------------------------------------------------------------------ the two scheduler callbacks
    async def prepare_for_streaming_generation(self, request: TTSRequest):
        conditioning = await self.tts_engine.get_audio_conditioning(request.speaker_files)
        return partial(self.tts_engine.get_generation_context, gpt_cond_latent=conditioning[0],
                       speaker_embeddings=conditioning[1])

    async def _prepare_generation_context(self, input_request: TTSRequest) -> Dict:
        input_request.start_time = time.time()
        fn = input_request.context_partial_function or self.tts_engine.get_generation_context
        gens, ids, spk, cond = await fn(input_request)
        return {"parallel_inputs": [{"generator": g, "speaker_embedding": spk, "multimodal_data": cond,
                                     "request": input_request} for g in gens],
                "request": input_request}

    async def _second_phase_fn(self, gen_input: Dict) -> AsyncGenerator[TTSOutput, None]:
        async for chunk in self.tts_engine.process_tokens_to_speech(
                generator=gen_input["generator"], speaker_embeddings=gen_input["speaker_embedding"],
                multimodal_data=gen_input["multimodal_data"], request=gen_input["request"]):
            yield chunk

    # ------------------------------------------------------------------ generation
    @staticmethod
    def split_requests(request: TTSRequest, max_length: int = 100000) -> List[TTSRequest]:
        """Texts longer than max_length become several requests with fresh ids (core/tts.py:236-255)."""
        if not isinstance(request.text, str) or len(request.text) <= max_length:
            return [request]
        out = []
        for i in range(0, len(request.text), max_length):
            r = request.copy()
            r.text = request.text[i:i + max_length]
            r.request_id = uuid.uuid4().hex
            out.append(r)
        return out

    async def _chunks(self, request: TTSRequest) -> AsyncGenerator[TTSOutput, None]:
        for sub in self.split_requests(request):
            async for chunk in self.scheduler.run(inputs=sub, request_id=sub.request_id,
                                                  first_phase_fn=self._prepare_generation_context,
                                                  second_phase_fn=self._second_phase_fn):
                yield chunk

    async def generate_speech_async(self, request: TTSRequest) -> Union[AsyncGenerator[TTSOutput, None], TTSOutput]:
        if self.tts_engine is None:
            raise RuntimeError("call from_pretrained() first")
        if request.stream:
            return self._chunks(request)
        subs = self.split_requests(request)

        async def one(sub: TTSRequest) -> List[TTSOutput]:
            return [c async for c in self.scheduler.run(inputs=sub, request_id=sub.request_id,
                                                        first_phase_fn=self._prepare_generation_context,
                                                        second_phase_fn=self._second_phase_fn)]
        parts = [await one(subs[0])] if len(subs) == 1 else await asyncio.gather(*[one(s) for s in subs])   # (no task for the usual case)
        return TTSOutput.combine_outputs([c for p in parts for c in p])

    def _submit(self, coro):
        return asyncio.run_coroutine_threadsafe(coro, self._loop)

    def generate_speech(self, request: TTSRequest) -> Union[Generator[TTSOutput, None, None], TTSOutput]:
        if not request.stream:
            return self._submit(self.generate_speech_async(request)).result()

        def streaming_wrapper():
            agen = self._chunks(request)
            try:
                while True:
                    try:
                        yield self._submit(agen.__anext__()).result()
                    except StopAsyncIteration:
                        return
            finally:
                self._submit(agen.aclose()).result()
        return streaming_wrapper()

    async def shutdown(self, keep_engine: bool = False):
        if self.scheduler:
            await self.scheduler.shutdown()
        if self.tts_engine:
            if keep_engine:   # stop the plugin's driver thread, leave the native engine to its owner (with_engine callers)
                drv = getattr(self.tts_engine, "driver", None)
                if drv is not None:
                    drv.shutdown()
            else:
                await self.tts_engine.shutdown()

    def close(self, keep_engine: bool = False):
        try:
            self._submit(self.shutdown(keep_engine)).result(timeout=10)
        finally:
            self._loop.call_soon_threadsafe(self._loop.stop)
            self._thread.join(timeout=5)
