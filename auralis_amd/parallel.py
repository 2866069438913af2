"""Multi-GPU: utterance-level data parallelism, one process (one engine) per GPU.

The path shards by independent units (each <=250-char chunk is its own sequence with its own KV and sampler
state); the only shared datum is the speaker conditioning, sent with ONE collective: a broadcast of
{gpt_cond_latent [32,1024], speaker_embedding [512]} = 133 120 B from the rank that computed it.  On the GPU
box the backend is "nccl" (= RCCL over xGMI) and the receive buffer is handed to the engine as a device
pointer (aur_set_conditioning_device); on CPU (gloo, used by the world_size-2 tests) it goes through host memory.
The reference has no counterpart (it forwards tensor_parallel_size to vLLM, XTTSv2.py:214-215, default 1).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

COND_ELEMS = 32 * 1024
SPK_ELEMS = 512
PAYLOAD_BYTES = (COND_ELEMS + SPK_ELEMS) * 4


def pack_conditioning(gpt_cond_latent: torch.Tensor, speaker_embedding: torch.Tensor) -> torch.Tensor:
    return torch.cat([gpt_cond_latent.reshape(-1).float(), speaker_embedding.reshape(-1).float()])


def broadcast_conditioning(engine, speaker_key: int, gpt_cond_latent: Optional[torch.Tensor],
                           speaker_embedding: Optional[torch.Tensor], src: int = 0,
                           device: Optional[torch.device] = None) -> torch.Tensor:
    """Broadcast one speaker's conditioning from `src` and register it with the local engine.

    Non-source ranks pass None for the tensors.  Returns the flat [33280] buffer (on `device`).
    """
    import torch.distributed as dist
    device = device or torch.device("cpu")
    buf = torch.empty(COND_ELEMS + SPK_ELEMS, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        buf.copy_(pack_conditioning(gpt_cond_latent, speaker_embedding))
    dist.broadcast(buf, src=src)
    if engine is not None:
        if buf.is_cuda:
            torch.cuda.synchronize(buf.device)
            engine.set_conditioning_device(speaker_key, buf.data_ptr(), buf.data_ptr() + COND_ELEMS * 4)
        else:
            engine.set_conditioning(speaker_key, buf[:COND_ELEMS].numpy(), buf[COND_ELEMS:].numpy())
    return buf


def _collective_device(device: Optional[torch.device]) -> torch.device:
    """Where the small agreement tensors live when the caller names no device: an NCCL-only process group (the backend
    INTEGRATION.md names) cannot reduce CPU tensors, so the default follows the backend."""
    import torch.distributed as dist
    if device is not None:
        return device
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _agree(ok: bool, device: torch.device) -> bool:
    """True on every rank iff every rank passed True.  Collective."""
    import torch.distributed as dist
    bad = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    return int(bad.item()) == 0


def comm_init_agreed(engine, device: Optional[torch.device] = None) -> str:
    """Build the engine's own RCCL communicator on every rank (aur_comm_init; torch.distributed only carries the 128-byte id)
    and AGREE on the outcome with one all_reduce: returns "" on every rank iff every rank's communicator is up, otherwise a
    non-empty reason on every rank (the failing rank's own message there, "failed on another rank" elsewhere).  Nothing enters a
    collective of the new communicator before this agreement, so a rank whose aur_comm_init raised cannot leave the others
    waiting inside ncclBroadcast.  Collective on its first call per engine; later calls return the agreed outcome.  Whatever
    happens in here -- including an exception out of the agreement itself -- the engine is left marked ready or failed, never
    half-initialised: a retry does not run aur_comm_init a second time."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    device = _collective_device(device)
    # the outcome of the first call is shared by construction (it was all-reduced), so later calls take the same branch on every
    # rank without a collective: up everywhere, or failed everywhere (a communicator that is up on SOME ranks is useless)
    if getattr(engine, "_comm_ready", False):
        return ""
    if getattr(engine, "_comm_failed", ""):
        return engine._comm_failed
    err = "comm_init_agreed did not complete"
    try:
        err = ""
        uid = None
        if rank == 0:   # a failure to create the id must not leave the other ranks waiting in the broadcast below
            try:
                uid = type(engine).comm_unique_id()
            except Exception as ex:   # noqa: BLE001 - reported on every rank
                uid = ex
        ids = [uid]
        dist.broadcast_object_list(ids, src=0, **({"device": device} if device.type != "cpu" else {}))
        if isinstance(ids[0], Exception):
            err = f"rank 0 could not create the RCCL communicator id: {ids[0]}"
        else:
            try:
                engine.comm_init(ids[0], rank, world)
            except Exception as ex:   # noqa: BLE001 - agreed on below
                err = f"rank {rank}: aur_comm_init: {type(ex).__name__}: {ex}"
        if not _agree(not err, device) and not err:
            err = "aur_comm_init failed on another rank"
    except BaseException as ex:
        err = f"rank {rank}: communicator agreement failed: {type(ex).__name__}: {ex}"
        raise
    finally:
        if err:
            engine._comm_failed = err
        else:
            engine._comm_ready = True
    return err


def broadcast_conditioning_native(engine, speaker_key: int, gpt_cond_latent: Optional[torch.Tensor],
                                  speaker_embedding: Optional[torch.Tensor], src: int = 0, device: Optional[torch.device] = None) -> None:
    """The same exchange with the collective INSIDE the library (aur_comm_init / aur_broadcast_conditioning: one ncclBroadcast on
    the engine's own RCCL communicator).  torch.distributed is used to hand the 128-byte communicator id to every rank and to
    agree that every rank's communicator is up (comm_init_agreed) BEFORE any rank enters the broadcast; raises on every rank
    otherwise.  The source registers its voice first and the ranks agree on THAT too: a source that cannot (tensors missing or
    mis-shaped, speaker table full) makes every rank raise instead of leaving the others inside ncclBroadcast.  `device`: where
    the agreement tensors live (default: the current CUDA device under the nccl backend, the CPU otherwise)."""
    import torch.distributed as dist
    device = _collective_device(device)
    err = comm_init_agreed(engine, device)
    if err:
        raise RuntimeError(err)
    src_err = ""
    if dist.get_rank() == src:
        try:
            engine.set_conditioning(speaker_key, gpt_cond_latent.reshape(32, 1024).float().cpu().numpy(),
                                    speaker_embedding.reshape(512).float().cpu().numpy())
        except Exception as ex:   # noqa: BLE001 - agreed on below
            src_err = f"rank {src}: could not register the voice to broadcast: {type(ex).__name__}: {ex}"
    if not _agree(not src_err, device):
        raise RuntimeError(src_err or f"rank {src} could not register the voice to broadcast")
    engine.broadcast_conditioning(speaker_key, src)


def all_ranks_equal(value) -> bool:
    """True on every rank iff every rank passed the same (picklable) value: the cross-rank check bench.py prints after the
    conditioning broadcast and after a fixed-seed verification batch.  Collective."""
    import torch.distributed as dist
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, value)
    return all(v == got[0] for v in got)


def shard_units(n_units: int, world: int, rank: int, per_gpu_batch: int = 64) -> List[int]:
    """Indices of the units (utterance chunks) owned by `rank`: blocks of `per_gpu_batch` dealt round-robin,
    so that C4 (512 utterances, 64/GPU x 8) gives every GPU one full batch and long-form streams stay ordered
    inside a block.  No data-path collective: results return per GPU over PCIe."""
    out = []
    for start in range(0, n_units, per_gpu_batch):
        if (start // per_gpu_batch) % world == rank:
            out.extend(range(start, min(n_units, start + per_gpu_batch)))
    return out


def merge_ordered(per_rank: Sequence[Sequence[tuple]]) -> List[tuple]:
    """Re-order (unit_index, payload) pairs from all ranks by unit index (what _yield_ordered_outputs does for
    chunk outputs in the reference, two_phase_scheduler.py:308-388)."""
    flat = [x for r in per_rank for x in r]
    return sorted(flat, key=lambda t: t[0])
