"""Long-form synthesis (BASELINE config 5: book-length text, mixed languages, chunk-ordered streaming).

The reference handles a book as ONE request: `TTS.split_requests` cuts it at 100 000 characters, `split_sentence`
cuts those at the per-language character limit, and chunk outputs are re-emitted in order (core/tts.py:236-355,
config/tokenizer.py:119-236, two_phase_scheduler.py:308-388).  A request carries one language tag, so mixed-language
material is fed paragraph by paragraph with `language="auto"`.  This helper does exactly that on top of the facade:
every paragraph becomes a TTSRequest (language detected per paragraph), up to `window` paragraphs are in flight at a
time so the engine's continuous batcher stays full, and audio is yielded strictly in (paragraph, chunk) order.
On several GPUs each rank takes the paragraphs `shard_units` deals to it (auralis_amd/parallel.py)."""
from __future__ import annotations

import asyncio
import re
from typing import AsyncGenerator, Iterable, List, Optional, Sequence, Tuple

from .api.output import TTSOutput
from .api.requests import TTSRequest


def split_paragraphs(text: str) -> List[str]:
    return [p.strip() for p in re.split(r"\n\s*\n", text) if p.strip()]


def build_requests(paragraphs: Sequence[str], speaker_files, seed: Optional[int] = None, **gen) -> List[TTSRequest]:
    reqs = []
    for i, p in enumerate(paragraphs):
        reqs.append(TTSRequest(text=p, speaker_files=speaker_files, language="auto", stream=True,
                               seed=None if seed is None else seed + 1000 * i, **gen))
    return reqs


async def stream_longform_async(tts, requests: Sequence[TTSRequest], window: int = 8
                                ) -> AsyncGenerator[Tuple[int, TTSOutput], None]:
    """Yield (paragraph index, chunk) in order; at most `window` paragraphs are being synthesised ahead."""
    queues = [asyncio.Queue() for _ in requests]
    sem = asyncio.Semaphore(max(1, window))
    _END = object()

    async def run(i: int, req: TTSRequest):
        try:
            async with sem:
                agen = await tts.generate_speech_async(req)
                async for chunk in agen:
                    await queues[i].put(chunk)
        except BaseException as e:
            await queues[i].put(e)
        finally:
            await queues[i].put(_END)

    tasks = [asyncio.ensure_future(run(i, r)) for i, r in enumerate(requests)]
    try:
        for i, q in enumerate(queues):
            while True:
                item = await q.get()
                if item is _END:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield i, item
    finally:
        for t in tasks:
            if not t.done():
                t.cancel()


def stream_longform(tts, requests: Sequence[TTSRequest], window: int = 8) -> Iterable[Tuple[int, TTSOutput]]:
    """Synchronous wrapper running on the facade's own event loop."""
    agen = stream_longform_async(tts, requests, window)
    try:
        while True:
            try:
                yield asyncio.run_coroutine_threadsafe(agen.__anext__(), tts._loop).result()
            except StopAsyncIteration:
                return
    finally:
        asyncio.run_coroutine_threadsafe(agen.aclose(), tts._loop).result()
