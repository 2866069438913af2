"""CPU: the N>1 path (utterance sharding + the single conditioning broadcast) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from auralis_amd.parallel import PAYLOAD_BYTES, broadcast_conditioning, merge_ordered, shard_units
from tests.fakes import FakeNativeEngine


def test_payload_size_matches_survey():
    assert PAYLOAD_BYTES == 133120


def test_shard_units_partitions_everything():
    for n, world in ((512, 8), (1900, 8), (64, 1), (10, 4), (130, 2)):
        seen = []
        for r in range(world):
            seen += shard_units(n, world, r, per_gpu_batch=64)
        assert sorted(seen) == list(range(n))
    assert [len(shard_units(512, 8, r)) for r in range(8)] == [64] * 8      # BASELINE config 4


def test_merge_ordered():
    assert merge_ordered([[(2, "c"), (0, "a")], [(1, "b")]]) == [(0, "a"), (1, "b"), (2, "c")]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = FakeNativeEngine()
    g = torch.arange(32 * 1024, dtype=torch.float32).reshape(1, 32, 1024) if rank == 0 else None
    s = torch.linspace(0, 1, 512).reshape(1, 512, 1) if rank == 0 else None
    broadcast_conditioning(eng, 7, g, s, src=0)
    got_g, got_s = eng.speakers[7]
    # every rank then works on its own shard; results are gathered only to check the ordering helper
    mine = [(i, f"r{rank}") for i in shard_units(10, world, rank, per_gpu_batch=2)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    from auralis_amd.parallel import all_ranks_equal
    same = all_ranks_equal(got_g.tobytes())                 # every rank holds the broadcast bytes
    differ = all_ranks_equal(f"rank {rank}")                # ... and the check can fail
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), g=got_g, s=got_s,
             order=np.array([i for i, _ in merge_ordered(gathered)]), same=np.bool_(same), differ=np.bool_(differ))
    dist.destroy_process_group()


def test_broadcast_conditioning_gloo_world2(tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["g"].reshape(-1), np.arange(32 * 1024, dtype=np.float32))
        assert np.allclose(z["s"].reshape(-1), np.linspace(0, 1, 512, dtype=np.float32))
        assert z["order"].tolist() == list(range(10))
        assert bool(z["same"]) and not bool(z["differ"])


class _CommEngine:
    """Stand-in for NativeEngine's communicator calls: comm_init fails on the ranks named in `bad`."""
    bad = ()

    def __init__(self, rank):
        self.rank, self.inits, self.bcasts = rank, 0, []

    @staticmethod
    def comm_unique_id():
        return b"\x01" * 128

    def comm_init(self, uid, rank, world):
        assert uid == b"\x01" * 128
        self.inits += 1
        if rank in self.bad:
            raise RuntimeError("aur_comm_init: injected failure")

    def set_conditioning(self, key, g, s):
        pass

    def broadcast_conditioning(self, key, src):
        self.bcasts.append((key, src))


def _agree_worker(rank, world, port, out_dir, bad):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auralis_amd.parallel import broadcast_conditioning_native, comm_init_agreed
    _CommEngine.bad = tuple(bad)
    eng = _CommEngine(rank)
    err = comm_init_agreed(eng)
    err2 = comm_init_agreed(eng)            # idempotent: a communicator that is up is not built twice
    entered = False
    try:
        g = torch.zeros(1, 32, 1024) if rank == 0 else None
        s = torch.zeros(1, 512, 1) if rank == 0 else None
        broadcast_conditioning_native(eng, 5, g, s, src=0)
        entered = True
    except RuntimeError:
        pass
    with open(os.path.join(out_dir, f"a{rank}.txt"), "w") as f:
        f.write(f"{err}|{err2}|{int(entered)}|{len(eng.bcasts)}|{eng.inits}")
    dist.destroy_process_group()


def test_native_route_is_entered_only_when_every_rank_built_its_communicator(tmp_path):
    """parallel.comm_init_agreed: the ranks all-reduce the outcome of aur_comm_init BEFORE any of them enters ncclBroadcast -- with a
    failure injected on rank 1 every rank gets a non-empty reason and NO rank calls aur_broadcast_conditioning (a rank that
    entered the collective alone would hang there); with no failure every rank enters it exactly once."""
    for bad, sub in (((1,), "bad"), ((), "ok")):
        d = tmp_path / sub
        d.mkdir()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_agree_worker, args=(2, port, str(d), list(bad)), nprocs=2, join=True)
        for r in range(2):
            err, err2, entered, n_bcast, inits = (d / f"a{r}.txt").read_text().split("|")
            if bad:
                assert err and err2, (r, err)
                assert ("rank 1" in err) == (r == 1) and ("another rank" in err) == (r == 0)
                assert entered == "0" and n_bcast == "0"
            else:
                assert err == "" and err2 == "" and entered == "1" and n_bcast == "1" and inits == "1"
            if bad:
                assert inits == "1"   # the agreed failure is remembered: nobody re-enters the id exchange alone


def _src_fail_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auralis_amd.parallel import broadcast_conditioning_native

    class Eng(_CommEngine):
        def set_conditioning(self, key, g, s):
            raise RuntimeError("speaker table full: every registered voice has undelivered sequences")
    _CommEngine.bad = ()
    eng = Eng(rank)
    err = ""
    try:
        g = torch.zeros(1, 32, 1024) if rank == 0 else None
        s = torch.zeros(1, 512, 1) if rank == 0 else None
        broadcast_conditioning_native(eng, 5, g, s, src=0)
    except RuntimeError as e:
        err = str(e)
    with open(os.path.join(out_dir, f"s{rank}.txt"), "w") as f:
        f.write(f"{err}|{len(eng.bcasts)}")
    dist.destroy_process_group()


def test_native_route_is_not_entered_when_the_source_cannot_register_its_voice(tmp_path):
    """the source registers the voice BEFORE the collective; if that raises on the source only (mis-shaped tensors, a full
    speaker table) the other ranks must not be left inside ncclBroadcast: the ranks agree on the source's outcome first, every rank
    raises, nobody calls aur_broadcast_conditioning."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_src_fail_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    e0, n0 = (tmp_path / "s0.txt").read_text().split("|")
    e1, n1 = (tmp_path / "s1.txt").read_text().split("|")
    assert "speaker table full" in e0 and "rank 0" in e0 and n0 == "0"
    assert "rank 0 could not register" in e1 and n1 == "0"
