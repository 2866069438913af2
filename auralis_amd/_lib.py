"""ctypes binding of include/auralis_amd.h (the stub a reference maintainer would add).

There is deliberately no fallback: if the HIP library is missing or no MI355X is visible, loading /
engine creation raises.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AURALIS_AMD_LIB") or os.path.join(_HERE, "_C", "libauralis_amd.so")   # override: A/B builds


class AurError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"auralis_amd error {code}: {msg}")
        self.code = code


class aur_config(C.Structure):
    _fields_ = [("n_layer", C.c_int32), ("max_seqs", C.c_int32), ("max_prefill_rows", C.c_int32),
                ("max_speakers", C.c_int32), ("vocoder_min_batch", C.c_int32), ("profile", C.c_int32),
                ("vocoder_fp16", C.c_int32), ("second_pass", C.c_int32), ("return_latents", C.c_int32), ("kv_fp16", C.c_int32),
                ("gemm_f32_exact", C.c_int32), ("gelu_erf", C.c_int32), ("admit_min_batch", C.c_int32), ("urgent_rows", C.c_int32)]


class aur_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class aur_cond_params(C.Structure):
    _fields_ = [("max_ref_length", C.c_int32), ("gpt_cond_len", C.c_int32), ("gpt_cond_chunk_len", C.c_int32),
                ("sound_norm_refs", C.c_int32)]


class aur_seq_desc(C.Structure):
    _fields_ = [("text_ids", C.POINTER(C.c_int32)), ("n_text", C.c_int32), ("speaker_key", C.c_uint64),
                ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("repetition_penalty", C.c_float), ("max_tokens", C.c_int32), ("seed", C.c_uint32),
                ("ignore_stop", C.c_int32), ("priority", C.c_int32)]


class aur_result(C.Structure):
    _fields_ = [("seq_id", C.c_uint64), ("n_tokens", C.c_int32), ("tokens", C.POINTER(C.c_int32)),
                ("n_samples", C.c_int32), ("wav", C.POINTER(C.c_float)), ("n_latent_rows", C.c_int32),
                ("latents", C.POINTER(C.c_float)), ("error", C.c_int32)]


class aur_stats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("prefill_rows", C.c_int64), ("decode_rows", C.c_int64),
                ("tokens_generated", C.c_int64), ("samples_generated", C.c_int64), ("vocoder_batches", C.c_int64),
                ("conv_launches", C.c_int64), ("conv_ms", C.c_double), ("conv_flops", C.c_double),
                ("conv_bytes", C.c_double), ("gemm_launches", C.c_int64), ("gemm_ms", C.c_double), ("gemm_ms_raw", C.c_double),
                ("event_pair_overhead_ms", C.c_double), ("gemm_flops", C.c_double),
                ("gemm_bytes", C.c_double), ("vocoder_ms", C.c_double), ("gpt_ms", C.c_double),
                ("kv_blocks_total", C.c_int64), ("kv_blocks_free", C.c_int64),
                ("gemm_kind_launches", C.c_int64 * 5), ("gemm_kind_ms", C.c_double * 5), ("gemm_kind_bytes", C.c_double * 5),
                ("gemm_kind_flops", C.c_double * 5), ("attn_launches", C.c_int64), ("attn_ms", C.c_double),
                ("attn_bytes", C.c_double), ("decode_steps", C.c_int64), ("decode_ms", C.c_double), ("prefill_ms", C.c_double),
                ("decode_weight_bytes", C.c_double), ("decode_kv_bytes", C.c_double),
                ("conv_class_launches", C.c_int64 * 5), ("conv_class_ms", C.c_double * 5), ("conv_class_bytes", C.c_double * 5),
                ("conv_class_flops", C.c_double * 5), ("prefill_batches", C.c_int64),
                ("result_blocks", C.c_int64), ("result_blocks_free", C.c_int64), ("result_block_bytes", C.c_int64),
                ("speakers", C.c_int64), ("sequences_tracked", C.c_int64)]

    def as_dict(self) -> Dict[str, float]:
        out = {}
        for n, _ in self._fields_:
            v = getattr(self, n)
            out[n] = list(v) if hasattr(v, "__len__") else v
        return out


# every symbol include/auralis_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "aur_last_error", "aur_version", "aur_engine_create", "aur_engine_destroy", "aur_load_weights",
    "aur_set_conditioning", "aur_set_conditioning_device", "aur_has_conditioning", "aur_compute_conditioning", "aur_comm_unique_id", "aur_comm_init", "aur_broadcast_conditioning",
    "aur_comm_info", "aur_conditioning_checksum",
    "aur_submit", "aur_step", "aur_poll_finished",
    "aur_release", "aur_cancel", "aur_vocode", "aur_sync", "aur_get_stats", "aur_reset_stats", "aur_set_profile", "aur_dbg_gemm", "aur_dbg_gemm_rows", "aur_dbg_gemm_rows_ksplit_stress", "aur_dbg_lane_xor_selftest",
    "aur_dbg_paged_attention", "aur_dbg_prompt_attention", "aur_dbg_layernorm", "aur_dbg_conv1d", "aur_dbg_conv1d_f16", "aur_dbg_prefill", "aur_dbg_sample",
]

_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the in-tree HIP library; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch first: it bundles its own libamdhip64; loading ours after it makes both share ONE HIP runtime
    # (loading the system runtime first and torch's second leaves the process with no visible device).
    import torch  # noqa: F401
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -m auralis_amd.build` (hipcc --offload-arch=gfx950). "
            "The MI355X path has no CPU fallback.")
    lib = C.CDLL(p)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    eng = C.c_void_p
    lib.aur_last_error.restype = C.c_char_p
    lib.aur_last_error.argtypes = []
    lib.aur_version.restype = C.c_int
    sig = {
        "aur_engine_create": [C.POINTER(aur_config), C.c_int, C.POINTER(eng)],
        "aur_engine_destroy": [eng],
        "aur_load_weights": [eng, C.POINTER(aur_tensor_desc), C.c_size_t],
        "aur_set_conditioning": [eng, C.c_uint64, fp, fp],
        "aur_set_conditioning_device": [eng, C.c_uint64, C.c_void_p, C.c_void_p],
        "aur_has_conditioning": [eng, C.c_uint64, ip],
        "aur_comm_unique_id": [C.POINTER(C.c_uint8)],
        "aur_comm_init": [eng, C.POINTER(C.c_uint8), C.c_int32, C.c_int32],
        "aur_broadcast_conditioning": [eng, C.c_uint64, C.c_int32],
        "aur_comm_info": [eng, ip, ip],
        "aur_conditioning_checksum": [eng, C.c_uint64, C.POINTER(C.c_uint64)],
        "aur_compute_conditioning": [eng, C.POINTER(fp), ip, C.c_int32, C.POINTER(aur_cond_params), fp, fp],
        "aur_submit": [eng, C.POINTER(aur_seq_desc), C.POINTER(C.c_uint64)],
        "aur_step": [eng, ip, ip],
        "aur_poll_finished": [eng, C.POINTER(aur_result), C.c_size_t, C.POINTER(C.c_size_t)],
        "aur_release": [eng, C.c_uint64],
        "aur_cancel": [eng, C.c_uint64],
        "aur_vocode": [eng, fp, ip, C.c_int32, C.c_int32, C.c_uint64, fp, C.c_int64, ip],
        "aur_sync": [eng],
        "aur_get_stats": [eng, C.POINTER(aur_stats)],
        "aur_reset_stats": [eng],
        "aur_set_profile": [eng, C.c_int32],
        "aur_dbg_gemm": [eng, fp, fp, fp, C.c_int32, C.c_int32, C.c_int32],
        "aur_dbg_gemm_rows": [eng, fp, fp, fp, fp, fp, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32],
        "aur_dbg_gemm_rows_ksplit_stress": [eng, C.c_int32, C.c_int32, C.POINTER(C.c_int64)],
        "aur_dbg_lane_xor_selftest": [eng, C.c_int32, C.POINTER(C.c_int64)],
        "aur_dbg_paged_attention": [eng, fp, fp, fp, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp],
        "aur_dbg_prompt_attention": [eng, fp, fp, fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, fp],
        "aur_dbg_layernorm": [eng, fp, fp, fp, fp, C.c_int32],
        "aur_dbg_conv1d": [eng, fp, fp, fp, fp, fp, ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                           C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32],
        "aur_dbg_conv1d_f16": [eng, fp, C.c_void_p, fp, fp, fp, ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32],
        "aur_dbg_prefill": [eng, ip, C.c_int32, C.c_uint64, C.c_float, fp, fp],
        "aur_dbg_sample": [eng, fp, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float,
                           C.POINTER(C.c_uint8), C.c_uint32, C.c_int32, ip],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class ResultLease:
    """One finished sequence's hold on its pinned result block (aur_result.wav / .latents stay valid until aur_release).
    `release()` is idempotent; dropping the last reference releases too, so an array built on the lease (poll(copy=False))
    returns the block when the array -- and every view of it -- is gone.  The reference hands out plain numpy copies
    (XTTSv2.py:804-811); the lease is what lets this path hand out the engine's own buffer instead."""
    __slots__ = ("_eng", "seq_id", "nbytes", "__weakref__")

    def __init__(self, eng: "NativeEngine", seq_id: int, nbytes: int):
        self._eng, self.seq_id, self.nbytes = eng, seq_id, nbytes

    def release(self):
        eng, self._eng = self._eng, None
        if eng is not None:
            eng._lease_dropped(self)

    def __del__(self):
        try:
            self.release()
        except Exception:   # interpreter shutdown
            pass


class _PinnedView:
    """__array_interface__ over `n` elements at `addr`; numpy keeps this object (and through it the lease) as the array's base."""
    __slots__ = ("__array_interface__", "lease")

    def __init__(self, addr: int, shape, typestr: str, lease: ResultLease):
        self.__array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (addr, False), "version": 3}
        self.lease = lease


class NativeEngine:
    """Thin object wrapper over the C ABI; one instance per GPU."""
    # poll(copy=False) hands out views of pinned host memory; past this many leased bytes (waveforms somebody keeps) new
    # results are copied instead, so that kept outputs cost pageable memory like the reference's, not pinned blocks
    LEASE_BUDGET_BYTES = 2 << 30

    def __init__(self, n_layer: int = 30, max_seqs: int = 64, device: int = 0, max_prefill_rows: int = 0,
                 max_speakers: int = 0, vocoder_min_batch: int = 0, profile: bool = False, vocoder_fp16: bool = False,
                 second_pass: bool = False, return_latents: bool = True, kv_fp16: bool = False,
                 gemm_f32_exact: bool = False, gelu_erf: bool = False, admit_min_batch: int = 0, urgent_rows: int = 0):
        self.lib = load_library()
        cfg = aur_config(n_layer, max_seqs, max_prefill_rows, max_speakers, vocoder_min_batch, int(profile),
                         int(vocoder_fp16), int(second_pass), int(return_latents), int(kv_fp16), int(gemm_f32_exact), int(gelu_erf), int(admit_min_batch), int(urgent_rows))
        h = C.c_void_p()
        self._check(self.lib.aur_engine_create(C.byref(cfg), device, C.byref(h)))
        self.h = h
        self.max_seqs = max_seqs
        self.n_layer = n_layer
        self._lease_lock = threading.Lock()
        self._leases: Dict[int, "weakref.ref"] = {}   # seq_id -> lease handed out by poll(copy=False)
        self._leased_bytes = 0
        self._closing = False

    def _check(self, rc: int):
        if rc != 0:
            raise AurError(rc, self.lib.aur_last_error().decode("utf-8", "replace"))

    def close(self):
        """Destroy the engine.  While arrays handed out by poll(copy=False) are alive their memory belongs to the engine, so the
        destruction waits for the last of them (it then happens on the thread that drops it)."""
        if not getattr(self, "h", None):
            return
        with self._lease_lock:
            self._closing = True
            if self._leases:
                return
            h, self.h = self.h, None
        self.lib.aur_engine_destroy(h)

    def _lease_dropped(self, lease: "ResultLease"):
        with self._lease_lock:
            if self._leases.pop(lease.seq_id, None) is None:
                return
            self._leased_bytes -= lease.nbytes
            h = self.h
            last = self._closing and not self._leases
            if last:
                self.h = None
        if h:
            self.lib.aur_release(h, lease.seq_id)
            if last:
                self.lib.aur_engine_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights / conditioning ------------------------------------------------------------------------
    def load_weights(self, packed: Dict[str, np.ndarray]):
        names = list(packed.keys())
        arrs = [_f32(packed[n]) for n in names]
        descs = (aur_tensor_desc * len(names))()
        keep = []
        for i, (n, a) in enumerate(zip(names, arrs)):
            b = n.encode()
            keep.append(b)
            descs[i].name = b
            descs[i].data = _fp(a)
            descs[i].numel = a.size
        self._check(self.lib.aur_load_weights(self.h, descs, len(names)))

    def set_conditioning(self, key: int, gpt_cond_latent, speaker_embedding):
        g = _f32(np.asarray(gpt_cond_latent).reshape(32, 1024))
        s = _f32(np.asarray(speaker_embedding).reshape(512))
        self._check(self.lib.aur_set_conditioning(self.h, key, _fp(g), _fp(s)))

    def has_conditioning(self, key: int) -> bool:
        out = C.c_int32()
        self._check(self.lib.aur_has_conditioning(self.h, key, C.byref(out)))
        return bool(out.value)

    def compute_conditioning(self, references, max_ref_length: int = 30, gpt_cond_len: int = 6, gpt_cond_chunk_len: int = 6,
                             sound_norm_refs: bool = False):
        """Reference audio (list of mono float32 arrays at 22 050 Hz) -> (gpt_cond_latent [1,32,1024], speaker_embedding
        [1,512,1]) computed by the HIP kernels (needs weights.pack_conditioning in load_weights)."""
        arrs = [_f32(np.asarray(r).reshape(-1)) for r in references]
        ptrs = (C.POINTER(C.c_float) * len(arrs))(*[_fp(a) for a in arrs])
        ns = _i32([a.size for a in arrs])
        p = aur_cond_params(int(max_ref_length), int(gpt_cond_len), int(gpt_cond_chunk_len), int(bool(sound_norm_refs)))
        g = np.empty((1, 32, 1024), np.float32)
        s = np.empty((1, 512, 1), np.float32)
        self._check(self.lib.aur_compute_conditioning(self.h, ptrs, _ip(ns), len(arrs), C.byref(p), _fp(g), _fp(s)))
        return g, s

    # -- RCCL inside the library (one engine per GPU per process) -------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = load_library().aur_comm_unique_id(buf)
        if rc != 0:
            raise AurError(rc, load_library().aur_last_error().decode("utf-8", "replace"))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.aur_comm_init(self.h, buf, rank, world))

    def broadcast_conditioning(self, key: int, root: int = 0):
        """Collective over the engine's communicator: the voice `key` registered on `root` arrives on every other rank."""
        self._check(self.lib.aur_broadcast_conditioning(self.h, key, root))

    def comm_info(self):
        """(ranks, rank) as the engine's RCCL communicator reports them (ncclCommCount / ncclCommUserRank); (0, -1) before comm_init."""
        n, r = C.c_int32(0), C.c_int32(-1)
        self._check(self.lib.aur_comm_info(self.h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def conditioning_checksum(self, key: int) -> int:
        """FNV-1a over the voice's conditioning as it sits in device memory (cross-rank equality check after a broadcast)."""
        out = C.c_uint64(0)
        self._check(self.lib.aur_conditioning_checksum(self.h, key, C.byref(out)))
        return int(out.value)

    def set_conditioning_device(self, key: int, d_gpt_cond_ptr: int, d_spk_ptr: int):
        self._check(self.lib.aur_set_conditioning_device(self.h, key, C.c_void_p(d_gpt_cond_ptr), C.c_void_p(d_spk_ptr)))

    # -- sequences -------------------------------------------------------------------------------------
    def submit(self, text_ids: Sequence[int], speaker_key: int, temperature: float = 0.75, top_p: float = 0.85,
               top_k: int = 50, repetition_penalty: float = 5.0, max_tokens: int = 605, seed: int = 0,
               ignore_stop: bool = False, priority: int = 0) -> int:
        ids = _i32(list(text_ids))
        d = aur_seq_desc(_ip(ids), len(ids), speaker_key, temperature, top_p, top_k, repetition_penalty,
                         max_tokens, seed & 0xFFFFFFFF, int(ignore_stop), int(priority))
        sid = C.c_uint64()
        self._check(self.lib.aur_submit(self.h, C.byref(d), C.byref(sid)))
        return sid.value

    def step(self):
        live, fin = C.c_int32(), C.c_int32()
        self._check(self.lib.aur_step(self.h, C.byref(live), C.byref(fin)))
        return live.value, fin.value

    def poll(self, cap: int = 64, want_latents: bool = True, copy: bool = True) -> List[dict]:
        """Finished sequences.  copy=True: owned numpy arrays, the engine's result block is released at once.  copy=False: `wav`
        (and `latents`) are VIEWS of the engine's pinned result block (aur_result.wav / .latents); the item's `lease`
        (ResultLease) gives the block back -- explicitly (lease.release() / release(seq_id)) or when the last array over it is
        dropped.  Token ids are always copied (a few hundred int32)."""
        res = (aur_result * cap)()
        n = C.c_size_t()
        self._check(self.lib.aur_poll_finished(self.h, res, cap, C.byref(n)))
        out = []
        for i in range(n.value):
            r = res[i]
            nbytes = 4 * (r.n_samples + (r.n_latent_rows * 1024 if want_latents else 0))
            lease = None
            if not copy and nbytes:
                with self._lease_lock:
                    if self._leased_bytes + nbytes <= self.LEASE_BUDGET_BYTES:
                        lease = ResultLease(self, r.seq_id, nbytes)
                        self._leases[r.seq_id] = weakref.ref(lease)
                        self._leased_bytes += nbytes

            def arr(ptr, shape):
                if lease is None:
                    return np.ctypeslib.as_array(ptr, shape=shape).copy()
                return np.asarray(_PinnedView(C.addressof(ptr.contents), shape, "<f4", lease))
            item = {
                "seq_id": r.seq_id,
                "tokens": (np.ctypeslib.as_array(r.tokens, shape=(r.n_tokens,)).copy() if r.n_tokens
                           else np.zeros(0, dtype=np.int32)),
                "wav": (arr(r.wav, (r.n_samples,)) if r.n_samples
                        else np.zeros(0, dtype=np.float32)),   # failed sequences carry no audio (error != 0)
                "error": r.error,
            }
            if want_latents and r.n_latent_rows:
                item["latents"] = arr(r.latents, (r.n_latent_rows, 1024))
            if lease is None:
                self._check(self.lib.aur_release(self.h, r.seq_id))
            else:
                item["lease"] = lease
            out.append(item)
        return out

    def release(self, seq_id: int):
        """Give a sequence's result block back now (after poll(copy=False)); the arrays over it must not be used afterwards.
        A sequence whose lease is already gone is not an error."""
        with self._lease_lock:
            ref = self._leases.get(seq_id)
        lease = ref() if ref is not None else None
        if lease is not None:
            lease.release()
        elif ref is not None:     # the lease object is being finalised: its __del__ releases
            return

    def cancel(self, seq_id: int):
        """aur_cancel: stop a sequence nobody waits for any more.  It is still reported by poll() (error = AUR_E_CANCELLED = -5, the
        tokens generated so far, no audio).  Does not wait for a running step."""
        self._check(self.lib.aur_cancel(self.h, seq_id))

    def run_until_done(self, max_steps: int = 100000, copy: bool = True) -> List[dict]:
        """Drive aur_step until nothing is live; returns finished results in completion order (copy: see poll)."""
        done: List[dict] = []
        last_fin = -1
        for _ in range(max_steps):
            live, fin = self.step()
            if fin != last_fin or live == 0:   # aur_step's second output counts finished sequences: poll only when it moved
                while True:                    # a vocoder batch (or a failed step) can finish more than one poll's worth at once
                    got = self.poll(cap=64, copy=copy)
                    done.extend(got)
                    if len(got) < 64:
                        break
                last_fin = fin
            if live == 0:
                break
        else:
            raise RuntimeError("run_until_done: step limit reached")
        return done

    def vocode(self, latents: np.ndarray, n_lat: Optional[Sequence[int]], speaker_key: int) -> List[np.ndarray]:
        lat = _f32(latents)
        if lat.ndim == 2:
            lat = lat[None]
        B, t_max, _ = lat.shape
        nl = _i32(n_lat if n_lat is not None else [t_max] * B)
        stride = int(np.floor(np.floor(t_max * 4.0) * (24000 / 22050))) * 256
        wav = np.zeros((B, stride), dtype=np.float32)
        ns = np.zeros(B, dtype=np.int32)
        self._check(self.lib.aur_vocode(self.h, _fp(lat), _ip(nl), B, t_max, speaker_key, _fp(wav), stride, _ip(ns)))
        return [wav[b, : ns[b]].copy() for b in range(B)]

    def sync(self):
        self._check(self.lib.aur_sync(self.h))

    def stats(self) -> Dict[str, float]:
        s = aur_stats()
        self._check(self.lib.aur_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self.lib.aur_reset_stats(self.h))

    def set_profile(self, every: int):
        """Profile mode on (every > 0: conv launches event-timed, replay batches after every `every`-th decode step) / off (0)."""
        self._check(self.lib.aur_set_profile(self.h, int(every)))

    # -- per-kernel debug entry points -------------------------------------------------------------------
    def dbg_gemm(self, X, W) -> np.ndarray:
        """prefill-regime GEMM (gemm_tile_kernel): X @ W"""
        X, W = _f32(X), _f32(W)
        M, K = X.shape
        K2, N = W.shape
        assert K == K2
        out = np.empty((M, N), dtype=np.float32)
        self._check(self.lib.aur_dbg_gemm(self.h, _fp(X), _fp(W), _fp(out), M, N, K))
        return out

    def dbg_gemm_rows(self, X, W, bias=None, gamma=None, beta=None, res=None, epi: int = 0) -> np.ndarray:
        """Decode-regime GEMM: epi 0 bias, 1 bias + gelu_new, 2 res + (X @ W + bias); gamma/beta => LayerNorm prologue."""
        X, W = _f32(X), _f32(W)
        M, K = X.shape
        K2, N = W.shape
        assert K == K2
        out = _f32(res).copy() if res is not None else np.zeros((M, N), dtype=np.float32)
        bias = None if bias is None else _f32(bias)
        ln = gamma is not None
        gamma = None if gamma is None else _f32(gamma)
        beta = None if beta is None else _f32(beta)
        self._check(self.lib.aur_dbg_gemm_rows(self.h, _fp(X), _fp(W), _fp(bias), _fp(gamma), _fp(beta), _fp(out), M, N, K,
                                               epi, int(ln)))
        return out

    def dbg_gemm_rows_ksplit_stress(self, M: int, iters: int) -> int:
        """`iters` back-to-back K-split projection launches at M rows vs the unsplit kernel, bitwise on the device: differing words."""
        bad = C.c_int64(-1)
        self._check(self.lib.aur_dbg_gemm_rows_ksplit_stress(self.h, int(M), int(iters), C.byref(bad)))
        return int(bad.value)

    def dbg_lane_xor_selftest(self, blocks: int = 64) -> int:
        """lane_xor<J> / wave_sum / wave_max (DPP + lane swaps) vs the shuffles they replace, on the GPU: results that differ bitwise."""
        bad = C.c_int64(-1)
        self._check(self.lib.aur_dbg_lane_xor_selftest(self.h, int(blocks), C.byref(bad)))
        return int(bad.value)

    def dbg_paged_attention(self, q, k, v, ctx, shared=0, kv_half=False) -> np.ndarray:
        """q [M][1024], k / v [M][ctx_max][1024], ctx [M] -> [M][1024]: the decode attention kernel on a paged pool built from k / v."""
        q, k, v, ctx = _f32(q), _f32(k), _f32(v), _i32(ctx)
        out = np.empty_like(q)
        self._check(self.lib.aur_dbg_paged_attention(self.h, _fp(q), _fp(k), _fp(v), ctx.ctypes.data_as(C.POINTER(C.c_int32)), q.shape[0],
                                                     k.shape[1], int(shared), int(bool(kv_half)), _fp(out)))
        return out

    def dbg_prompt_attention(self, q, k, v, row_seq, row_pos, shared=0, kv_half=False) -> np.ndarray:
        """q [M][1024], k / v [n_seq][ctx_max][1024], row_seq / row_pos [M] -> [M][1024]: the prefill attention kernel on a paged pool."""
        q, k, v, row_seq, row_pos = _f32(q), _f32(k), _f32(v), _i32(row_seq), _i32(row_pos)
        out = np.empty_like(q)
        ip = C.POINTER(C.c_int32)
        self._check(self.lib.aur_dbg_prompt_attention(self.h, _fp(q), _fp(k), _fp(v), row_seq.ctypes.data_as(ip), row_pos.ctypes.data_as(ip),
                                                      q.shape[0], k.shape[0], k.shape[1], int(shared), int(bool(kv_half)), _fp(out)))
        return out

    def dbg_layernorm(self, h, gamma, beta) -> np.ndarray:
        h, gamma, beta = _f32(h), _f32(gamma), _f32(beta)
        out = np.empty_like(h)
        self._check(self.lib.aur_dbg_layernorm(self.h, _fp(h), _fp(gamma), _fp(beta), _fp(out), h.shape[0]))
        return out

    def dbg_conv1d(self, x, wp, bias, res, lens, ks, dil, padl, slope, cout, ups_s=0, ups_p=0) -> np.ndarray:
        x, wp = _f32(x), _f32(wp)
        B, cin, L = x.shape
        mtot = wp.shape[0] * wp.shape[3]
        lout = L * max(1, ups_s)
        bias = None if bias is None else _f32(bias)
        res = None if res is None else _f32(res)
        out = np.empty((B, cout, lout), dtype=np.float32)
        lens = _i32(lens)
        self._check(self.lib.aur_dbg_conv1d(self.h, _fp(x), _fp(wp), _fp(bias), _fp(res), _fp(out), _ip(lens), B, cin,
                                            mtot, cout, L, ks, dil, padl, slope, ups_s, ups_p))
        return out

    def dbg_conv1d_f16(self, x, wp16, bias, res, lens, ks, dil, padl, slope, cout, mtot, ups_s=0, ups_p=0) -> np.ndarray:
        x = _f32(x)
        wp16 = np.ascontiguousarray(wp16, dtype=np.float16)
        B, cin, L = x.shape
        lout = L * max(1, ups_s)
        bias = None if bias is None else _f32(bias)
        res = None if res is None else _f32(res)
        out = np.empty((B, cout, lout), dtype=np.float32)
        lens = _i32(lens)
        self._check(self.lib.aur_dbg_conv1d_f16(self.h, _fp(x), wp16.ctypes.data_as(C.c_void_p), _fp(bias), _fp(res),
                                                _fp(out), _ip(lens), B, cin, mtot, cout, L, ks, dil, padl, slope, ups_s, ups_p))
        return out

    def dbg_prefill(self, text_ids, speaker_key: int, repetition_penalty: float = 1.0):
        ids = _i32(list(text_ids))
        n_rows = 32 + len(ids) + 1
        rows = np.empty((n_rows, 1024), dtype=np.float32)
        logits = np.empty(1026, dtype=np.float32)
        self._check(self.lib.aur_dbg_prefill(self.h, _ip(ids), len(ids), speaker_key, repetition_penalty, _fp(rows), _fp(logits)))
        return rows, logits

    def dbg_sample(self, logits, temperature, top_p, top_k, repetition_penalty=1.0, seen=None, seed=0, step=0):
        lg = _f32(logits)
        if lg.ndim == 1:
            lg = lg[None]
        B = lg.shape[0]
        toks = np.zeros(B, dtype=np.int32)
        sp = None
        if seen is not None:
            seen = np.ascontiguousarray(seen, dtype=np.uint8).reshape(B, 1026)
            sp = seen.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self.lib.aur_dbg_sample(self.h, _fp(lg), B, temperature, top_p, top_k, repetition_penalty, sp,
                                            seed & 0xFFFFFFFF, step, _ip(toks)))
        return toks
