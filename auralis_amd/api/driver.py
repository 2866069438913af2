"""Bridge between asyncio and the native continuous batcher.

The reference needs three mechanisms here: vLLM's background engine loop, a HiddenStatesCollector that hands the
second-pass hidden states from the vLLM worker thread to asyncio with a 3 s timeout
(components/vllm/hidden_state_collector.py:89-167) and a 10 ms polling loop that re-orders chunk outputs
(two_phase_scheduler.py:308-388).  Natively one driver thread per engine calls aur_step() while anything is live and
resolves one asyncio future per sequence with loop.call_soon_threadsafe — no polling, no second pass."""
from __future__ import annotations

import asyncio
import threading
from typing import Any, Dict, Optional, Tuple


class EngineDriver:
    def __init__(self, engine: Any):
        """`engine` needs submit(**kw)->id, step()->(live, finished), poll()->list[dict] (NativeEngine or a test double)."""
        self.engine = engine
        self._pending: Dict[int, Tuple[asyncio.AbstractEventLoop, asyncio.Future]] = {}
        self._lock = threading.Lock()
        self._wake = threading.Event()
        self._stop = False
        self._error: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._run, name="auralis-amd-driver", daemon=True)
        self._thread.start()

    def submit(self, loop: asyncio.AbstractEventLoop, reregister=None, **seq) -> "asyncio.Future":
        """`reregister` (optional, no arguments): registers the sequence's voice again.  The engine's speaker table is bounded and
        a voice is only pinned from aur_submit on, so between the caller's presence check and this call another request can
        have evicted it; the engine then answers "unknown speaker_key" (AUR_E_INVALID) and the submission is repeated once
        after `reregister()`."""
        if self._error is not None:
            raise RuntimeError("engine driver stopped") from self._error
        fut = loop.create_future()
        with self._lock:
            try:
                sid = self.engine.submit(**seq)
            except Exception as e:
                if reregister is None or "unknown speaker_key" not in str(e):
                    raise
                reregister()
                sid = self.engine.submit(**seq)
            self._pending[sid] = (loop, fut)
        self._wake.set()
        return fut

    def _resolve(self, item: dict):
        with self._lock:
            ent = self._pending.pop(item["seq_id"], None)
        if ent is None:
            return
        loop, fut = ent

        def done():
            if not fut.done():
                if item.get("error"):
                    fut.set_exception(RuntimeError(f"sequence failed with code {item['error']}"))
                else:
                    fut.set_result(item)
        loop.call_soon_threadsafe(done)

    def _fail_all(self, exc: BaseException):
        with self._lock:
            pend, self._pending = self._pending, {}
        for loop, fut in pend.values():
            loop.call_soon_threadsafe(lambda f=fut: (not f.done()) and f.set_exception(exc))

    def _run(self):
        while not self._stop:
            self._wake.wait(timeout=0.5)
            self._wake.clear()
            try:
                while not self._stop:
                    live, _ = self.engine.step()
                    for item in self.engine.poll():
                        self._resolve(item)
                    if live == 0:
                        break
            except BaseException as e:  # first error wins, as in the reference scheduler (two_phase_scheduler.py:279-291)
                self._error = e
                self._fail_all(e)
                return

    def shutdown(self):
        self._stop = True
        self._wake.set()
        self._thread.join(timeout=5)
