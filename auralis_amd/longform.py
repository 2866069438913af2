"""Long-form synthesis (BASELINE config 5: book-length text, mixed languages, chunk-ordered streaming).

The reference handles a book as ONE request: `TTS.split_requests` cuts it at 100 000 characters, `split_sentence`
cuts those at the per-language character limit, and chunk outputs are re-emitted in order (core/tts.py:236-355,
config/tokenizer.py:119-236, two_phase_scheduler.py:308-388).  A request carries one language tag, so mixed-language
material is fed paragraph by paragraph with `language="auto"`.  This helper does exactly that on top of the facade:
every paragraph becomes a TTSRequest (language detected per paragraph), up to `window` paragraphs are in flight at a
time so the engine's continuous batcher stays full, and audio is yielded strictly in (paragraph, chunk) order.
On several GPUs each rank takes the paragraphs `shard_units` deals to it and the chunks stream to one rank in order
(`stream_sharded` / `synthesize_sharded`, auralis_amd/parallel.py)."""
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


def stream_sharded(tts, requests: Sequence[TTSRequest], window: int = 8, paragraphs_per_block: int = 8, dst: int = 0
                   ) -> Iterable[Tuple[int, np.ndarray]]:
    """Book on several GPUs, STREAMED (one process per GPU, torch.distributed initialised; BASELINE config 5 at 8 x MI355X).

    Rank r synthesises the paragraphs `shard_units` deals to it in blocks of `paragraphs_per_block`; every finished chunk
    is sent to rank `dst` at once (one small header + the PCM, point to point, no collective), and `dst` yields
    (paragraph index, PCM) strictly in (paragraph, chunk) order as soon as the next one in order is available — the
    multi-process form of the reference's ordered re-emission (two_phase_scheduler.py:308-388), which hands chunks out as
    they complete instead of at the end of the request.  On the other ranks the iterator yields nothing and returns when
    their last chunk has been delivered (they must still drain it).  Without a process group it is the single-GPU stream."""
    import queue
    import threading

    import torch
    import torch.distributed as dist

    from .parallel import shard_units
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        for i, c in stream_longform(tts, requests, window):
            yield i, np.asarray(c.array, dtype=np.float32)
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    owner = {}
    for r in range(world):
        for i in shard_units(len(requests), world, r, per_gpu_batch=paragraphs_per_block):
            owner[i] = r
    mine = [i for i in range(len(requests)) if owner[i] == rank]
    # local synthesis runs ahead on the facade's loop (at most `window` paragraphs); this thread only moves finished chunks
    local_q: "queue.Queue" = queue.Queue()
    _END = object()

    def produce():
        try:
            last = None
            for local_i, chunk in stream_longform(tts, [requests[i] for i in mine], window):
                if last is not None and mine[local_i] != last:
                    local_q.put((last, None))                      # paragraph `last` is complete
                last = mine[local_i]
                local_q.put((last, np.ascontiguousarray(chunk.array, dtype=np.float32)))
            if last is not None:
                local_q.put((last, None))
            local_q.put(_END)
        except BaseException as e:                                 # surfaces in the consumer
            local_q.put(e)

    th = threading.Thread(target=produce, name="auralis-amd-longform", daemon=True)
    th.start()

    def next_local():
        item = local_q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    if rank != dst:
        # header = [paragraph index, samples in this chunk (0 = paragraph complete)]
        while True:
            item = next_local()
            if item is _END:
                break
            i, pcm = item
            n = 0 if pcm is None else int(pcm.shape[0])
            dist.send(torch.tensor([i, n], dtype=torch.int64, device=dev), dst=dst)
            if n:
                dist.send(torch.from_numpy(pcm).to(dev), dst=dst)
        th.join()
        return
    for i in range(len(requests)):
        src = owner[i]
        while True:
            if src == rank:
                item = next_local()
                assert item is not _END and item[0] == i, "local paragraphs arrive in index order"
                pcm = item[1]
            else:
                hdr = torch.empty(2, dtype=torch.int64, device=dev)
                dist.recv(hdr, src=src)
                pi, n = int(hdr[0]), int(hdr[1])
                assert pi == i, f"rank {src} sent paragraph {pi}, expected {i}"
                pcm = None
                if n:
                    buf = torch.empty(n, dtype=torch.float32, device=dev)
                    dist.recv(buf, src=src)
                    pcm = buf.cpu().numpy()
            if pcm is None:
                break
            yield i, pcm
    assert next_local() is _END
    th.join()


def synthesize_sharded(tts, requests: Sequence[TTSRequest], window: int = 8, paragraphs_per_block: int = 8,
                       dst: int = 0) -> Optional[TTSOutput]:
    """The whole book as one TTSOutput on rank `dst` (None elsewhere): stream_sharded, concatenated."""
    import torch.distributed as dist
    parts = [pcm for _, pcm in stream_sharded(tts, requests, window, paragraphs_per_block, dst)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() != dst:
        return None
    return TTSOutput(array=np.concatenate(parts) if parts else np.zeros(0, np.float32), sample_rate=24000)
