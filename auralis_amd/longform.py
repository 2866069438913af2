"""Long-form synthesis (BASELINE config 5: book-length text, mixed languages, chunk-ordered streaming).

The reference handles a book as ONE request: `TTS.split_requests` cuts it at 100 000 characters, `split_sentence`
cuts those at the per-language character limit, and chunk outputs are re-emitted in order (core/tts.py:236-355,
config/tokenizer.py:119-236, two_phase_scheduler.py:308-388).  A request carries one language tag, so mixed-language
material is fed paragraph by paragraph with `language="auto"`.  This helper does exactly that on top of the facade:
every paragraph becomes a TTSRequest (language detected per paragraph), up to `window` paragraphs (default: twice the facade's
concurrency) are in flight at a time so the engine's continuous batcher stays full, and audio is yielded strictly in (paragraph, chunk) order.
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
    """One streaming request per paragraph.  The FIRST one is marked latency-critical (TTSRequest.priority): the stream is consumed
    in order, so its first chunk is the time to first audio of the whole book; everything behind it only has to keep ahead of the
    playback."""
    reqs = []
    for i, p in enumerate(paragraphs):
        reqs.append(TTSRequest(text=p, speaker_files=speaker_files, language="auto", stream=True,
                               seed=None if seed is None else seed + 1000 * i, **({"priority": 1} if i == 0 else {}), **gen))
    return reqs


def default_window(tts) -> int:
    """Paragraphs in flight when the caller names none: twice the facade's chunk concurrency (`scheduler_max_concurrency`), so that its
    semaphore -- and behind it the engine's slots and admission queue -- stays full while finished paragraphs wait to be emitted in order."""
    return max(8, 2 * int(getattr(tts, "scheduler_max_concurrency", 4) or 4))


async def stream_longform_async(tts, requests: Sequence[TTSRequest], window: Optional[int] = None
                                ) -> AsyncGenerator[Tuple[int, TTSOutput], None]:
    """Yield (paragraph index, chunk) in order; at most `window` paragraphs (default_window) are being synthesised ahead."""
    queues = [asyncio.Queue() for _ in requests]
    sem = asyncio.Semaphore(max(1, window or default_window(tts)))
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


def stream_longform(tts, requests: Sequence[TTSRequest], window: Optional[int] = None) -> Iterable[Tuple[int, TTSOutput]]:
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


def stream_sharded(tts, requests: Sequence[TTSRequest], window: Optional[int] = None, paragraphs_per_block: int = 8, dst: int = 0
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

    stop = threading.Event()   # set when this rank stops consuming: the producer drops what it still has instead of synthesising on

    def produce():
        """(paragraph, pcm) per chunk and (paragraph, None) once per OWNED paragraph, in index order, also for a paragraph that
        produced no chunk at all; an exception travels through the queue"""
        try:
            done = 0                                               # paragraphs of `mine` already terminated
            for local_i, chunk in stream_longform(tts, [requests[i] for i in mine], window):
                if stop.is_set():
                    return
                while done < local_i:
                    local_q.put((mine[done], None))
                    done += 1
                local_q.put((mine[local_i], np.ascontiguousarray(chunk.array, dtype=np.float32)))
            while done < len(mine):
                local_q.put((mine[done], None))
                done += 1
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

    # Wire protocol, point to point, per message: header int64[3] = [paragraph, status, payload length] then the payload.
    #   status >= 0 : a chunk of `status` samples (0 samples is a legitimate, empty chunk), payload = float32 PCM
    #   _DONE       : the paragraph is complete, no payload
    #   _ERROR      : synthesis failed on the sending rank, payload = utf-8 text of the exception; the stream of that rank ends
    _DONE, _ERROR = -1, -2

    def send_hdr(i, status, length):
        dist.send(torch.tensor([i, status, length], dtype=torch.int64, device=dev), dst=dst)

    if rank != dst:
        try:
            while True:
                item = next_local()
                if item is _END:
                    break
                i, pcm = item
                if pcm is None:
                    send_hdr(i, _DONE, 0)
                else:
                    send_hdr(i, int(pcm.shape[0]), int(pcm.shape[0]))
                    if pcm.shape[0]:
                        dist.send(torch.from_numpy(pcm).to(dev), dst=dst)
        except BaseException as e:                                 # tell dst instead of leaving it in recv forever
            # ... unless the failure IS the channel (a send that raised: peer gone, communicator aborted): a second blocking send
            # on it would hang where the first one failed
            # (the torch.distributed error classes are looked up defensively: on a build without them an AttributeError here would
            # mask `e` and skip the _ERROR message, which is the hang this handler exists to prevent)
            channel_errors = tuple(getattr(dist, n) for n in ("DistBackendError", "DistNetworkError") if hasattr(dist, n)) + (ConnectionError, BrokenPipeError)
            if not isinstance(e, channel_errors):
                msg = f"rank {rank}: {type(e).__name__}: {e}".encode("utf-8", "replace")[:4096]
                send_hdr(-1, _ERROR, len(msg))
                dist.send(torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev), dst=dst)
            raise
        finally:
            stop.set()
            th.join(timeout=5.0)
        return

    ended = set()          # source ranks whose stream is over (error seen)

    def recv_msg(src):
        hdr = torch.empty(3, dtype=torch.int64, device=dev)
        dist.recv(hdr, src=src)
        pi, status, length = int(hdr[0]), int(hdr[1]), int(hdr[2])
        if status == _ERROR:
            buf = torch.empty(length, dtype=torch.uint8, device=dev)
            if length:
                dist.recv(buf, src=src)
            ended.add(src)
            raise RuntimeError("long-form synthesis failed on " + bytes(buf.cpu().numpy().tobytes()).decode("utf-8", "replace"))
        if status == _DONE:
            return pi, None
        buf = torch.empty(status, dtype=torch.float32, device=dev)
        if status:
            dist.recv(buf, src=src)
        return pi, buf.cpu().numpy()

    i = 0
    in_paragraph = False   # headers of paragraph i have been consumed but not its terminator
    try:
        for i in range(len(requests)):
            src = owner[i]
            in_paragraph = True
            while True:
                if src == rank:
                    item = next_local()
                    assert item is not _END and item[0] == i, "local paragraphs arrive in index order"
                    pcm = item[1]
                else:
                    pi, pcm = recv_msg(src)
                    assert pi == i, f"rank {src} sent paragraph {pi}, expected {i}"
                if pcm is None:
                    in_paragraph = False
                    break
                yield i, pcm
        i = len(requests)
        assert next_local() is _END
    finally:
        # The consumer stopped early (closed the iterator, raised, or a rank reported an error): the other ranks are still
        # sending and would block in dist.send forever.  Receive and discard what they have left, paragraph by paragraph.
        for j in range(i, len(requests)):
            src = owner[j]
            if src == rank or src in ended:
                continue
            try:
                while True:
                    _, pcm = recv_msg(src)
                    if pcm is None:
                        break
            except RuntimeError:
                pass                                               # that rank's stream has ended with its own error
        stop.set()                                                 # (the drain above blocks until the other ranks have sent what they
        th.join(timeout=5.0)                                       # had queued: they stop at their next chunk, not at once)


def synthesize_sharded(tts, requests: Sequence[TTSRequest], window: Optional[int] = None, paragraphs_per_block: int = 8,
                       dst: int = 0) -> Optional[TTSOutput]:
    """The whole book as one TTSOutput on rank `dst` (None elsewhere): stream_sharded, concatenated."""
    import torch.distributed as dist
    parts = [pcm for _, pcm in stream_sharded(tts, requests, window, paragraphs_per_block, dst)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() != dst:
        return None
    return TTSOutput(array=np.concatenate(parts) if parts else np.zeros(0, np.float32), sample_rate=24000)
