"""Bridge between asyncio and the native continuous batcher.

The reference needs three mechanisms here: vLLM's background engine loop, a HiddenStatesCollector that hands the
second-pass hidden states from the vLLM worker thread to asyncio with a 3 s timeout
(components/vllm/hidden_state_collector.py:89-167) and a 10 ms polling loop that re-orders chunk outputs
(two_phase_scheduler.py:308-388).  Natively one driver thread per engine calls aur_step() while anything is live and
resolves one asyncio future per sequence with loop.call_soon_threadsafe — no polling loop, no second pass.

What the driver thread does per iteration is aur_step and nothing else: aur_poll_finished is called only when the step's
finished counter moved (or nothing is live any more), and results are taken as VIEWS of the engine's pinned result blocks
(`poll(copy=False)`): no waveform is copied on the thread that issues the next decode step.  The view carries a lease
(auralis_amd._lib.ResultLease) that gives the block back (aur_release) when the last array referring to it is dropped, i.e.
with the TTSOutput the plugin builds from it.

A failed aur_step fails the sequences that were in flight (the engine reports them through aur_poll_finished with
aur_result.error set and stays usable, engine.hip: Engine::step / fail_in_flight); their futures get the exception, queued
sequences and later submissions go on.  Only `max_consecutive_failures` failed steps in a row (a lost device) stop the driver
for good: then every pending future fails and submit() raises."""
from __future__ import annotations

import asyncio
import threading
import time
from typing import Any, Dict, Optional, Tuple


class EngineDriver:
    def __init__(self, engine: Any, max_consecutive_failures: int = 3, burst_gap_s: float = 1e-3, burst_max_s: float = 25e-3):
        """`engine` needs submit(**kw)->id, step()->(live, finished_total), poll(copy=False)->list[dict] (NativeEngine or a
        test double).  `burst_gap_s` / `burst_max_s`: an IDLE engine does not start on the first submission of a burst -- N
        concurrent generate_speech calls reach it a fraction of a millisecond apart, the event loop tokenises them one after
        the other, and a prefill pass costs the same ~4.5 ms of launches for one prompt as for 64 -- but once no submission has
        arrived for burst_gap_s (at most burst_max_s after the first).  A lone request pays burst_gap_s once; a running engine
        never waits (its admission groups arrivals itself, aur_config.admit_min_batch), and a latency-critical submission
        (priority > 0) ends the wait at once."""
        self.engine = engine
        self.max_consecutive_failures = max(1, int(max_consecutive_failures))
        self.burst_gap_s, self.burst_max_s = float(burst_gap_s), float(burst_max_s)
        self._last_submit = 0.0
        self._urgent = False          # a latency-critical submission (priority > 0) is in the burst: the engine starts at once
        self._burst_loops: set = set()   # event loops that submitted since the engine last went idle
        self._pending: Dict[int, Tuple[asyncio.AbstractEventLoop, asyncio.Future]] = {}
        self._lock = threading.Lock()
        self._wake = threading.Event()
        self._stop = False
        self._error: Optional[BaseException] = None
        self.failed_steps = 0          # aur_step calls that raised (the driver went on after them unless they came in a row)
        self.cancelled = 0             # sequences stopped with aur_cancel because their consumer had gone
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
            # a consumer that walks away (a cancelled task, a closed stream) cancels the future: the engine stops the sequence
            # instead of decoding it to the end and vocoding it for nobody (aur_cancel)
            fut.add_done_callback(lambda f, sid=sid: self._on_done(f, sid))
            self._last_submit = time.perf_counter()
            self._burst_loops.add(loop)
            if seq.get("priority", 0) > 0:
                self._urgent = True
        self._wake.set()
        return fut

    def _on_done(self, fut: "asyncio.Future", sid: int):
        if not fut.cancelled():
            return
        with self._lock:
            known = self._pending.get(sid) is not None
        cancel = getattr(self.engine, "cancel", None)
        if known and cancel is not None:
            try:
                cancel(sid)          # the engine still reports the sequence through poll(); _resolve_all drops it there
                self.cancelled += 1
            except Exception:        # noqa: BLE001 - already finished / released: nothing to stop
                pass
            self._wake.set()

    def _full_house(self) -> bool:
        slots = getattr(self.engine, "max_seqs", None)
        return bool(slots) and len(self._pending) >= int(slots)

    @staticmethod
    def _loop_busy(loop: asyncio.AbstractEventLoop) -> bool:
        """Callbacks queued on `loop` right now (CPython's BaseEventLoop keeps them in `_ready`; a loop without it counts as idle).
        Read from the driver thread without a lock: len() of a deque is atomic, and a stale answer only moves the end of the burst wait."""
        ready = getattr(loop, "_ready", None)
        try:
            return ready is not None and len(ready) > 0
        except TypeError:
            return False

    def _resolve_all(self, items, step_error: Optional[BaseException] = None):
        """Hand a poll's results to their futures: ONE call_soon_threadsafe per event loop (each one is a write to the loop's wake-up
        pipe; 64 utterances finishing in one vocoder batch used to be 64 of them)."""
        by_loop: Dict[asyncio.AbstractEventLoop, list] = {}
        with self._lock:
            ents = [(item, self._pending.pop(item["seq_id"], None)) for item in items]
        for item, ent in ents:
            if ent is None:
                lease = item.get("lease")
                if lease is not None:      # nobody waits for it (cancelled / unknown id): give the block back at once
                    lease.release()
                continue
            by_loop.setdefault(ent[0], []).append((ent[1], item))

        def done(pairs):
            for fut, item in pairs:
                if fut.done():             # cancelled consumer: the result is dropped, the lease goes with it
                    continue
                if item.get("error"):
                    exc = RuntimeError(f"sequence {item['seq_id']} failed with code {item['error']}"
                                       + (f": {step_error}" if step_error is not None else ""))
                    if step_error is not None:
                        exc.__cause__ = step_error
                    fut.set_exception(exc)
                else:
                    fut.set_result(item)
        for loop, pairs in by_loop.items():
            loop.call_soon_threadsafe(done, pairs)

    def _fail_all(self, exc: BaseException):
        with self._lock:
            pend, self._pending = self._pending, {}
        for loop, fut in pend.values():
            loop.call_soon_threadsafe(lambda f=fut: (not f.done()) and f.set_exception(exc))

    def _drain(self, step_error: Optional[BaseException] = None):
        while True:                    # a vocoder batch (or a failed step) can finish more than one poll's worth at once
            got = self.engine.poll(cap=64, copy=False)
            if got:
                self._resolve_all(got, step_error)
            if len(got) < 64:
                return

    def _run(self):
        try:
            self._loop()
        except BaseException as e:  # noqa: BLE001 - a driver that dies silently would leave every future pending for ever
            self._error = e
            self._fail_all(e)

    def _loop(self):
        while not self._stop:
            self._wake.wait(timeout=0.5)
            self._wake.clear()
            t_end = time.perf_counter() + self.burst_max_s
            while not self._stop:     # the burst that woke an idle engine is still arriving
                now = time.perf_counter()
                if self._urgent or now >= t_end:
                    break
                if self._full_house():     # every slot is spoken for: whatever arrives next waits for a free one anyway
                    break
                # quiet for burst_gap_s AND the submitting loop has nothing queued: a loop thread that stalls in the middle of a burst
                # (a garbage collection, a slow tokenizer call) still has the other requests' tasks in its ready queue
                if now - self._last_submit >= self.burst_gap_s:
                    with self._lock:      # (submit() adds to the set under the same lock)
                        loops = tuple(self._burst_loops)
                    if not any(self._loop_busy(lp) for lp in loops):
                        break
                time.sleep(self.burst_gap_s * 0.25)
            self._urgent = False
            with self._lock:
                self._burst_loops.clear()
            last_fin = None
            in_a_row = 0
            while not self._stop:
                try:
                    live, fin = self.engine.step()
                    in_a_row = 0
                except Exception as e:  # noqa: BLE001 - handed to the futures of the sequences it hit
                    self.failed_steps += 1
                    in_a_row += 1
                    if in_a_row >= self.max_consecutive_failures:
                        raise           # nothing but failures: the device is gone (the reference stops at its FIRST error,
                                        # two_phase_scheduler.py:279-291)
                    self._drain(e)      # the sequences that were in flight, failed by the engine
                    with self._lock:
                        if not self._pending:
                            break
                    last_fin = None
                    continue
                if fin != last_fin or live == 0:
                    self._drain()
                    last_fin = fin
                if live == 0:
                    break

    def shutdown(self):
        self._stop = True
        self._wake.set()
        self._thread.join(timeout=5)
