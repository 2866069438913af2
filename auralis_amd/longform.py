"""Long-form synthesis (BASELINE config 5: book-length text, mixed languages, chunk-ordered streaming).

The reference handles a book as ONE request: `TTS.split_requests` cuts it at 100 000 characters, `split_sentence`
cuts those at the per-language character limit, and chunk outputs are re-emitted in order (core/tts.py:236-355,
config/tokenizer.py:119-236, two_phase_scheduler.py:308-388).  A request carries one language tag, so mixed-language
material is fed paragraph by paragraph with `language="auto"`.  This helper does exactly that on top of the facade:
every paragraph becomes a TTSRequest (language detected per paragraph), up to `window` paragraphs are in flight at a
time so the engine's continuous batcher stays full, and audio is yielded strictly in (paragraph, chunk) order.
On several GPUs each rank takes the paragraphs `shard_units` deals to it (`synthesize_sharded`, auralis_amd/parallel.py)."""
from __future__ import annotations

import asyncio
import re
from typing import AsyncGenerator, Iterable, List, Optional, Sequence, Tuple

import numpy as np

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


def synthesize_sharded(tts, requests: Sequence[TTSRequest], window: int = 8, paragraphs_per_block: int = 8,
                       dst: int = 0) -> Optional[TTSOutput]:
    """Book on several GPUs (one process per GPU, torch.distributed initialised; BASELINE config 5 at 8 x MI355X): rank r
    synthesises the paragraphs `shard_units` deals to it in blocks of `paragraphs_per_block`, no collective on the data
    path; the per-paragraph audio is gathered once at the end and rank `dst` returns the book in order (other ranks None).
    Without a process group it is the single-GPU path."""
    import torch.distributed as dist

    from .parallel import merge_ordered, shard_units
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return TTSOutput.combine_outputs([c for _, c in stream_longform(tts, requests, window)])
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_units(len(requests), world, rank, per_gpu_batch=paragraphs_per_block)
    parts = {}
    for local_i, chunk in stream_longform(tts, [requests[i] for i in mine], window):
        parts.setdefault(mine[local_i], []).append(chunk.array)
    payload = [(i, np.concatenate(a)) for i, a in parts.items()]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if rank != dst:
        return None
    ordered = merge_ordered(gathered)
    assert [i for i, _ in ordered] == list(range(len(requests))), "a paragraph is missing from the gather"
    return TTSOutput(array=np.concatenate([a for _, a in ordered]), sample_rate=24000)
