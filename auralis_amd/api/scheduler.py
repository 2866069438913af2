"""TwoPhaseScheduler — same call contract as the reference (src/auralis/common/scheduling/two_phase_scheduler.py:
run(inputs, first_phase_fn, second_phase_fn, request_id) is an async generator of outputs in chunk order), built on
asyncio primitives only: phase 1 runs once per request, phase 2 runs one task per chunk gated by
Semaphore(second_phase_concurrency), and ordered re-emission waits on per-chunk queues instead of the reference's
10 ms polling loop (two_phase_scheduler.py:308-388).  Continuous batching itself happens inside the HIP engine."""
from __future__ import annotations

import asyncio
from typing import Any, AsyncGenerator, Awaitable, Callable, Dict, Optional

_END = object()


class TwoPhaseScheduler:
    def __init__(self, second_phase_concurrency: int = 10, request_timeout: Optional[float] = None,
                 generator_timeout: Optional[float] = None):
        self.second_phase_concurrency = max(1, int(second_phase_concurrency))
        self.request_timeout = request_timeout
        self.generator_timeout = generator_timeout
        self._sem: Optional[asyncio.Semaphore] = None
        self._tasks: set = set()
        self.is_running = True

    def _semaphore(self) -> asyncio.Semaphore:
        if self._sem is None:
            self._sem = asyncio.Semaphore(self.second_phase_concurrency)
        return self._sem

    async def run(self, inputs: Any, first_phase_fn: Callable[[Any], Awaitable[Dict]],
                  second_phase_fn: Callable[[Dict], AsyncGenerator], request_id: Optional[str] = None
                  ) -> AsyncGenerator[Any, None]:
        ctx = await asyncio.wait_for(first_phase_fn(inputs), self.request_timeout)
        gens = ctx["parallel_inputs"]
        queues = [asyncio.Queue() for _ in gens]
        sem = self._semaphore()

        async def pump(i: int, gen_input: Dict):
            try:
                async with sem:
                    agen = second_phase_fn(gen_input)
                    while True:
                        try:
                            item = await asyncio.wait_for(agen.__anext__(), self.generator_timeout)
                        except StopAsyncIteration:
                            break
                        await queues[i].put(item)
            except BaseException as e:  # propagate to the consumer in order (first error wins)
                await queues[i].put(e)
            finally:
                await queues[i].put(_END)

        tasks = [asyncio.ensure_future(pump(i, g)) for i, g in enumerate(gens)]
        self._tasks.update(tasks)
        try:
            for q in queues:
                while True:
                    item = await q.get()
                    if item is _END:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    yield item
        finally:
            for t in tasks:
                if not t.done():
                    t.cancel()
                self._tasks.discard(t)
            # the consumer left early (stream closed, timeout, a chunk failed): the chunks nobody will read are told so.  A pump that was
            # still waiting for its turn never awaited its chunk, so cancelling the task alone would leave that chunk decoding
            for g in gens:
                h = g.get("generator") if isinstance(g, dict) else None
                if hasattr(h, "cancel"):
                    h.cancel()

    async def shutdown(self):
        self.is_running = False
        for t in list(self._tasks):
            t.cancel()
