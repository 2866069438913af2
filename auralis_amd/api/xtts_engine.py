"""XTTSv2Engine — the reference's engine plugin (src/auralis/models/xttsv2/XTTSv2.py:39-820) re-based on the HIP
library: same public methods, same return shapes; vLLM, the second GPT pass and the torch HiFi-GAN are replaced by
aur_submit / aur_step / aur_poll_finished.

`speaker_files` may be reference audio (paths or bytes; RIFF/WAVE and FLAC natively, api/codecs.py): conditioning is computed
once per speaker on the GPU by the HIP kernels behind aur_compute_conditioning (csrc/cond_net.h, SURVEY §8f #1;
auralis_amd/conditioning.py is the PyTorch restatement the tests compare it with), a path/bytes of an .npz holding
`gpt_cond_latent` [1,32,1024] and `speaker_embedding` [1,512,1], or a dict / tuple with those arrays."""
from __future__ import annotations

import asyncio
import collections
import hashlib
import json
import os
import time
from typing import Any, AsyncGenerator, List, Optional, Tuple

import numpy as np

from .driver import EngineDriver
from .engine_base import BaseAsyncTTSEngine, ConditioningConfig, register_model
from .output import TTSOutput
from .requests import TTSRequest
from .text import XTTSTokenizer


try:                       # 133 KB per request: xxh3 takes ~10 us, blake2b ~190 us of the event-loop thread
    from xxhash import xxh3_64 as _Hash64
except ImportError:        # same role, slower
    def _Hash64():
        return hashlib.blake2b(digest_size=8)


def _content_key(g: np.ndarray, s: np.ndarray) -> int:
    """64-bit key of a voice's conditioning CONTENT (fed through the buffer protocol: no copy).  Keys are local to a process."""
    h = _Hash64()
    h.update(np.ascontiguousarray(g).data)
    h.update(np.ascontiguousarray(s).data)
    return int.from_bytes(h.digest(), "little")


class ChunkHandle:
    """Opaque per-chunk "token generator" handed to the facade (the reference passes a vLLM async generator)."""

    def __init__(self, future: "asyncio.Future", request_id: str, n_text: int):
        self.future = future
        self.request_id = request_id
        self.n_text = n_text

    def cancel(self):
        """Nobody will read this chunk: stop it in the engine (EngineDriver -> aur_cancel).  Nothing happens once it is resolved."""
        fut = self.future
        if not fut.done():
            fut.get_loop().call_soon_threadsafe(fut.cancel)


class XTTSv2Engine(BaseAsyncTTSEngine):
    model_type = "xtts"

    def __init__(self, native_engine: Any, tokenizer: XTTSTokenizer, max_concurrency: int = 10,
                 gpt_max_audio_tokens: int = 605, conditioning_weights: Optional[dict] = None):
        self.native = native_engine
        self.conditioning_weights = conditioning_weights   # xtts-v2.safetensors tensors of the once-per-speaker modules
        self._cond_cache = collections.OrderedDict()   # LRU, bounded (a serving process sees an open-ended set of voices)
        self._cond_cache_max = 64
        self.tokenizer = tokenizer
        self.max_concurrency = max_concurrency
        self.gpt_max_audio_tokens = gpt_max_audio_tokens
        self.driver = EngineDriver(native_engine)
        self._speakers = {}
        self._seed_counter = 0
        # bench / test knob (aur_seq_desc.ignore_stop): every chunk runs to gpt_max_audio_tokens whatever it samples, so that
        # runs do identical work (SURVEY 8d "fixed-length mode"); never set by from_pretrained
        self.fixed_length = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, max_concurrency: int = 10, device: int = 0,
                        vocoder: str = "fp16", **kwargs) -> "XTTSv2Engine":
        """Load a checkpoint directory in the reference's on-disk format (checkpoint.py docstring).  kwargs the
        reference forwards to vLLM (tensor_parallel_size, pipeline_parallel_size, gpt_model, torch_dtype, device_map;
        XTTSv2.py:235-243) are accepted; tp/pp other than 1 are rejected (the path shards by utterance).
        `admit_min_batch` / `vocoder_min_batch` are the batcher's two grouping knobs, `max_speakers` the size of the engine's voice
        table (include/auralis_amd.h, aur_config)."""
        from .._lib import NativeEngine
        from ..checkpoint import load_checkpoint, read_checkpoint_config
        from ..weights import pack_all
        if kwargs.get("tensor_parallel_size", 1) != 1 or kwargs.get("pipeline_parallel_size", 1) != 1:
            raise ValueError("the MI355X path replicates the 0.4 B-parameter model per GPU; use one engine per GPU")
        gpt_sd, xtts_sd = load_checkpoint(pretrained_model_name_or_path)
        # both config.json are read and checked against what the kernels are compiled for (XTTSv2.py:276-277 builds
        # XTTSGPTConfig / XTTSConfig from them); the activation and the generation length come from the file
        ck = read_checkpoint_config(pretrained_model_name_or_path, gpt_sd)
        # vocoder="fp16": HiFi-GAN convs on fp16-input / fp32-accumulate MFMA (waveform within 1e-5 RMS of the fp32
        # path, the reference's own GPU path autocasts to fp16); vocoder="fp32": exact-f32 MFMA parity mode
        native = NativeEngine(n_layer=ck.n_layer, max_seqs=max(1, max_concurrency), device=device,
                              vocoder_fp16=(vocoder == "fp16"), return_latents=False, gelu_erf=ck.gelu_erf,
                              admit_min_batch=int(kwargs.get("admit_min_batch", 0)), vocoder_min_batch=int(kwargs.get("vocoder_min_batch", 0)),
                              max_speakers=int(kwargs.get("max_speakers", 0)), urgent_rows=int(kwargs.get("urgent_rows", 0)))
        native.load_weights(pack_all(gpt_sd, xtts_sd))
        if any(k.startswith("conditioning_encoder.") for k in xtts_sd):
            from ..weights import pack_conditioning
            native.load_weights(pack_conditioning(xtts_sd))
        tok_file = None
        for cand in ("tokenizer.json", os.path.join("gpt", "tokenizer.json")):
            p = os.path.join(pretrained_model_name_or_path, cand)
            if os.path.isfile(p):
                tok_file = p
        synthetic_tok = False
        cfg_path = os.path.join(pretrained_model_name_or_path, "core_xttsv2", "config.json")
        if tok_file is None and os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                synthetic_tok = bool(json.load(f).get("synthetic_tokenizer", False))
        vocab = xtts_sd["text_embedding.weight"].shape[0]
        cond_w = {k: v for k, v in xtts_sd.items()
                  if k.startswith(("conditioning_", "hifigan_decoder.speaker_encoder.", "mel_stats"))}
        return cls(native, XTTSTokenizer(tok_file, vocab_size=vocab, synthetic=synthetic_tok), max_concurrency=max_concurrency,
                   gpt_max_audio_tokens=ck.gpt_max_audio_tokens,
                   conditioning_weights=cond_w if any(k.startswith("conditioning_encoder.") for k in cond_w) else None)

    @property
    def conditioning_config(self) -> ConditioningConfig:
        return ConditioningConfig(speaker_embeddings=True, gpt_like_decoder_conditioning=True)

    @property
    def device(self):
        return "cuda"

    @property
    def dtype(self):
        return "float32"

    def get_memory_usage_curve(self):
        """KV blocks per concurrent sequence (replaces the vLLM gpu_memory_utilization heuristic, XTTSv2.py:152-171)."""
        s = self.native.stats()
        per_seq_mb = 66 * 2 * 16 * 16 * 64 * 4 * self.native.n_layer / 2 ** 20
        return {"kv_mb_per_sequence": per_seq_mb, "kv_blocks_total": s.get("kv_blocks_total", 0)}

    # ------------------------------------------------------------------ conditioning
    async def get_audio_conditioning(self, audio_reference, max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6,
                                     librosa_trim_db=None, sound_norm_refs=False, load_sr=22050):
        """-> (gpt_cond_latent [1,32,1024], speaker_embedding [1,512,1]) as float32 numpy arrays."""
        refs = list(audio_reference) if isinstance(audio_reference, (list, tuple)) else [audio_reference]
        if len(refs) == 2 and hasattr(refs[0], "shape") and hasattr(refs[1], "shape"):
            refs = [{"gpt_cond_latent": refs[0], "speaker_embedding": refs[1]}]
        ref = refs[0]

        def _is_npz(r):
            return (isinstance(r, str) and r.endswith(".npz")) or (isinstance(r, (bytes, bytearray)) and r[:2] == b"PK")
        if isinstance(ref, dict):
            g, s = ref["gpt_cond_latent"], ref["speaker_embedding"]
        elif _is_npz(ref):
            import io
            z = np.load(ref if isinstance(ref, str) else io.BytesIO(ref))
            g, s = z["gpt_cond_latent"], z["speaker_embedding"]
        else:
            if self.conditioning_weights is None:
                raise NotImplementedError(
                    "this checkpoint carries no conditioning_encoder / perceiver / speaker_encoder weights: pass "
                    "precomputed conditioning (.npz or dict with gpt_cond_latent [1,32,1024], speaker_embedding [1,512,1])")
            from .. import conditioning as Cn
            # cache key = full content of every reference (a path is keyed by the bytes it holds now, not by its name) + every
            # parameter that changes the result
            hk = hashlib.blake2b(digest_size=16)
            for r in refs:
                if isinstance(r, (bytes, bytearray, memoryview)):
                    data = bytes(r)
                else:
                    with open(r, "rb") as f:
                        data = f.read()
                hk.update(len(data).to_bytes(8, "little"))
                hk.update(data)
            hk.update(f"{max_ref_length}/{gpt_cond_len}/{gpt_cond_chunk_len}/{librosa_trim_db}/{sound_norm_refs}/{load_sr}".encode())
            key = hk.digest()
            if key in self._cond_cache:
                self._cond_cache.move_to_end(key)
            else:
                hip = getattr(self.native, "compute_conditioning", None)
                if hip is None:
                    raise RuntimeError("the engine has no compute_conditioning entry point: speaker conditioning from reference "
                                       "audio runs in HIP (aur_compute_conditioning); there is no PyTorch fallback in the product")

                # loader on the host (decode, mono, resample to 22 050 Hz, clip: utilities.py:74-98), networks in HIP
                def _run():
                    pcm = []
                    for r in refs:
                        a = Cn.load_audio(r, load_sr)
                        if load_sr != 22050:
                            a = Cn.resample(a, load_sr, 22050)
                        pcm.append(a[0].numpy())
                    return hip(pcm, max_ref_length=max_ref_length, gpt_cond_len=gpt_cond_len,
                               gpt_cond_chunk_len=gpt_cond_chunk_len, sound_norm_refs=sound_norm_refs)
                self._cond_cache[key] = await asyncio.to_thread(_run)
                while len(self._cond_cache) > self._cond_cache_max:
                    self._cond_cache.popitem(last=False)
            g, s = self._cond_cache[key]
        g = np.asarray(getattr(g, "numpy", lambda: g)(), dtype=np.float32).reshape(1, 32, 1024)
        s = np.asarray(getattr(s, "numpy", lambda: s)(), dtype=np.float32).reshape(1, 512, 1)
        return g, s

    def _register_speaker(self, g: np.ndarray, s: np.ndarray) -> int:
        """Content-addressed speaker key.  The engine's speaker table is bounded (aur_config.max_speakers) and evicts the least
        recently used idle voice, so presence is asked of the engine itself (aur_has_conditioning, which also refreshes the
        voice's LRU stamp) instead of being mirrored in an ever-growing Python dict; an evicted voice is simply re-registered."""
        key = _content_key(g, s)
        has = getattr(self.native, "has_conditioning", None)
        if has is not None:
            if not has(key):
                self.native.set_conditioning(key, g, s)
        elif key not in self._speakers:   # engines without the query (test doubles)
            self.native.set_conditioning(key, g, s)
            self._speakers[key] = True
        return key

    # ------------------------------------------------------------------ phase 1
    async def get_generation_context(self, request: TTSRequest, gpt_cond_latent=None, speaker_embeddings=None
                                     ) -> Tuple[List[ChunkHandle], List[str], Any, Any]:
        if gpt_cond_latent is None or speaker_embeddings is None:
            gpt_cond_latent, speaker_embeddings = await self.get_audio_conditioning(
                request.speaker_files, request.max_ref_length, request.gpt_cond_len, request.gpt_cond_chunk_len)
        g_np = np.asarray(gpt_cond_latent, dtype=np.float32)
        s_np = np.asarray(speaker_embeddings, dtype=np.float32)
        key = self._register_speaker(g_np, s_np)
        chunks = self.tokenizer.batch_encode_with_split(request.text, request.language)
        loop = asyncio.get_running_loop()
        handles, ids = [], []
        for idx, text_ids in enumerate(chunks):
            if request.seed is None:
                self._seed_counter += 1
                # stable across processes (Python's hash() of a str is salted per process)
                seed = (int.from_bytes(hashlib.blake2b(str(request.request_id).encode(), digest_size=4).digest(), "little")
                        ^ self._seed_counter) & 0xFFFFFFFF
            else:
                seed = (request.seed + idx) & 0xFFFFFFFF
            fut = self.driver.submit(loop, reregister=lambda: self.native.set_conditioning(key, g_np, s_np),
                                     text_ids=text_ids, speaker_key=key, temperature=request.temperature,
                                     top_p=request.top_p, top_k=request.top_k,
                                     repetition_penalty=request.repetition_penalty,
                                     max_tokens=self.gpt_max_audio_tokens, seed=seed,
                                     **({"ignore_stop": True} if self.fixed_length else {}),
                                     # a latency-critical request: its first chunk is the one somebody waits for
                                     **({"priority": int(request.priority)} if idx == 0 and getattr(request, "priority", 0) > 0 else {}))
            rid = f"{request.request_id}_{idx}"
            handles.append(ChunkHandle(fut, rid, len(text_ids)))
            ids.append(rid)
        return handles, ids, speaker_embeddings, gpt_cond_latent

    # ------------------------------------------------------------------ phase 2
    async def process_tokens_to_speech(self, generator: ChunkHandle, speaker_embeddings=None, multimodal_data=None,
                                       request: Optional[TTSRequest] = None) -> AsyncGenerator[TTSOutput, None]:
        assert speaker_embeddings is not None, "Speaker embeddings must be provided for speech generation with XTTSv2."
        item = await generator.future
        # item["wav"] is the engine's own pinned result block (EngineDriver polls with copy=False): the TTSOutput's array is a view
        # of it and gives the block back when it is dropped (ResultLease); nothing is copied between the GPU's store and the caller
        yield TTSOutput(array=item["wav"], sample_rate=24000,
                        start_time=request.start_time if request is not None else None,
                        token_length=int(len(item["tokens"])))

    async def shutdown(self):
        self.driver.shutdown()
        close = getattr(self.native, "close", None)
        if close:
            close()


register_model("xtts", XTTSv2Engine)
