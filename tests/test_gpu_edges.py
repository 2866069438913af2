"""GPU: edge cases — maximum sizes, error paths, several speakers in one batch, seeding."""
import numpy as np
import pytest
import torch

from auralis_amd._lib import AurError
from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_text_ids
from tests.gpu_util import SPK_KEY, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    e, gpt_sd, xtts_sd, cond, spk = make_engine(2, max_seqs=3)
    from oracle import xtts_oracle as O
    yield e, O.GPTOracle(gpt_sd, xtts_sd), cond, spk
    e.close()


def test_maximum_lengths(ctx, dims):
    """605 generated tokens (gpt_max_audio_tokens) after a 300-token text: last KV blocks, last latent rows, longest
    vocoder input; ids bit-exact against the oracle."""
    from oracle import xtts_oracle as O
    e, gpt, cond, _ = ctx
    ids = make_synthetic_text_ids(dims, n_text=300, seed=5)
    ref = gpt.generate(gpt.build_cond(cond, ids), O.SamplingCfg(temperature=0.0, max_tokens=605, ignore_stop=True))
    e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=605, ignore_stop=True)
    got = e.run_until_done()[0]
    assert len(got["tokens"]) == 605 and got["tokens"].tolist() == ref["tokens"]
    assert got["wav"].shape == (dims.voc.samples_for_latents(605),) and np.isfinite(got["wav"]).all()
    assert got["latents"].shape == (605, 1024)


def test_longest_prompt_accepted_and_overflow_rejected(ctx, dims):
    e, _, _, _ = ctx
    ids = make_synthetic_text_ids(dims, n_text=402, seed=6)          # 32 + 402 + 1 + 605 = 1040 <= 1047
    e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=3, ignore_stop=True)
    assert len(e.run_until_done()[0]["tokens"]) == 3
    with pytest.raises(AurError):                                      # text position table has 404 rows
        e.submit(make_synthetic_text_ids(dims, n_text=405, seed=6), SPK_KEY, max_tokens=3)
    with pytest.raises(AurError):
        e.submit([261, 999999, 0], SPK_KEY, max_tokens=3)              # id outside the text vocabulary
    assert e.step() == (0, e.step()[1])                                # nothing was queued


def test_error_paths(ctx, dims):
    e, _, _, _ = ctx
    ids = make_synthetic_text_ids(dims, n_text=8)
    with pytest.raises(AurError) as ei:
        e.submit(ids, 987654321)                                      # speaker never registered
    assert ei.value.code == -1 and "speaker" in str(ei.value)
    for bad in (dict(max_tokens=0), dict(max_tokens=606), dict(repetition_penalty=0.0)):
        with pytest.raises(AurError):
            e.submit(ids, SPK_KEY, **bad)
    with pytest.raises(AurError):
        e._check(e.lib.aur_release(e.h, 123456))
    # the engine is still healthy afterwards
    e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=4, ignore_stop=True)
    assert len(e.run_until_done()) == 1


def test_two_speakers_in_one_batch(ctx, dims):
    e, _, cond, spk = ctx
    g = torch.Generator().manual_seed(77)
    cond2 = torch.randn(1, 32, 1024, generator=g) * 0.02
    spk2 = torch.randn(1, 512, 1, generator=g)
    spk2 = spk2 / spk2.norm()
    e.set_conditioning(4242, cond2.numpy(), spk2.numpy())
    ids = make_synthetic_text_ids(dims, n_text=14, seed=9)
    solo = {}
    for key in (SPK_KEY, 4242):
        e.submit(ids, key, temperature=0.0, max_tokens=10, ignore_stop=True)
        solo[key] = e.run_until_done()[0]
    assert solo[SPK_KEY]["tokens"].tolist() != solo[4242]["tokens"].tolist() or \
        not np.array_equal(solo[SPK_KEY]["wav"], solo[4242]["wav"])
    sid = {e.submit(ids, key, temperature=0.0, max_tokens=10, ignore_stop=True): key for key in (SPK_KEY, 4242, SPK_KEY)}
    outs = e.run_until_done()
    assert len(outs) == 3
    for o in outs:
        s = solo[sid[o["seq_id"]]]
        assert o["tokens"].tolist() == s["tokens"].tolist() and np.array_equal(o["wav"], s["wav"])
    # a speaker that live sequences still use cannot be replaced under the same key
    e.submit(ids, 4242, temperature=0.0, max_tokens=30, ignore_stop=True)
    e.step()
    with pytest.raises(AurError):
        e.set_conditioning(4242, cond.numpy(), spk.numpy())
    e.run_until_done()


def test_seed_controls_sampling(ctx, dims):
    e, _, _, _ = ctx
    ids = make_synthetic_text_ids(dims, n_text=10, seed=3)
    runs = []
    for seed in (1, 1, 2):
        e.submit(ids, SPK_KEY, temperature=0.9, top_k=50, top_p=0.9, max_tokens=16, seed=seed, ignore_stop=True)
        runs.append(e.run_until_done()[0]["tokens"].tolist())
    assert runs[0] == runs[1] and runs[0] != runs[2]


def test_pipelined_decode_equals_synchronous(dims, monkeypatch):
    """The engine launches decode step s+1 before it has read back step s (a sequence that finished in s rides along
    as a ghost row).  Ragged finishes (natural stop tokens + different max_tokens), more sequences than batcher slots
    (admissions while a speculative step is in flight), sampling: tokens and audio must equal the synchronous engine."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.weights import pack_all
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 3.0                    # natural stops after a few (sampled) tokens
    packed = pack_all(gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd))
    cond, spk = make_synthetic_conditioning(dims)

    def run(pipeline):
        monkeypatch.setenv("AUR_DECODE_PIPELINE", "1" if pipeline else "0")
        e = NativeEngine(n_layer=2, max_seqs=3)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
            sids = []
            for k in range(8):
                ids = make_synthetic_text_ids(dims, n_text=9 + 3 * k, seed=20 + k)
                sids.append(e.submit(ids, SPK_KEY, temperature=0.9 if k % 3 else 0.0, top_k=50, top_p=0.85,
                                     max_tokens=[1, 2, 7, 30, 12, 3, 30, 18][k], seed=100 + k, ignore_stop=(k == 3)))
            outs = {o["seq_id"]: o for o in e.run_until_done()}
            return [outs[s] for s in sids]
        finally:
            e.close()

    a, b = run(True), run(False)
    lens = [len(o["tokens"]) for o in a]
    print("token counts", lens)
    assert lens[0] == 1 and lens[1] <= 2 and lens[3] == 30 and any(n < m for n, m in zip(lens, [1, 2, 7, 30, 12, 3, 30, 18]))
    for x, y in zip(a, b):
        assert x["tokens"].tolist() == y["tokens"].tolist()
        assert np.array_equal(x["wav"], y["wav"]) and np.array_equal(x["latents"], y["latents"])


def test_failed_step_releases_slots_blocks_and_engine_stays_usable(dims, monkeypatch):
    """A throw inside aur_step (here injected: AUR_TEST_FAIL_STEP) fails the sequences that were in flight (error code in
    their results, no audio), returns their slots / KV blocks / pool entries, keeps queued sequences, and the engine
    goes on to produce the same output as a fresh one."""
    ids = [make_synthetic_text_ids(dims, n_text=10 + k, seed=50 + k) for k in range(3)]
    monkeypatch.delenv("AUR_TEST_FAIL_STEP", raising=False)
    ref_e, *_ = make_engine(2, max_seqs=2)
    try:
        ref_e.submit(ids[2], SPK_KEY, temperature=0.0, max_tokens=12, ignore_stop=True)
        ref = ref_e.run_until_done()[0]
    finally:
        ref_e.close()
    monkeypatch.setenv("AUR_TEST_FAIL_STEP", "3")
    e, *_ = make_engine(2, max_seqs=2)
    try:
        sids = [e.submit(i, SPK_KEY, temperature=0.0, max_tokens=12, ignore_stop=True) for i in ids]   # 2 slots: the third waits
        e.step()
        e.step()
        with pytest.raises(AurError) as ei:
            e.step()
        assert ei.value.code == -2 and "injected" in str(ei.value)
        failed = e.poll()
        assert sorted(o["seq_id"] for o in failed) == sids[:2]
        assert all(o["error"] == -2 and len(o["wav"]) == 0 for o in failed)
        outs = e.run_until_done()                                      # the queued sequence is admitted and finishes
        assert [o["seq_id"] for o in outs] == [sids[2]] and outs[0]["error"] == 0
        assert outs[0]["tokens"].tolist() == ref["tokens"].tolist() and np.array_equal(outs[0]["wav"], ref["wav"])
        st = e.stats()
        assert st["kv_blocks_total"] - st["kv_blocks_free"] == 2        # only the speaker's shared prefix blocks
        e.submit(ids[0], SPK_KEY, temperature=0.0, max_tokens=5, ignore_stop=True)
        assert len(e.run_until_done()[0]["tokens"]) == 5
    finally:
        e.close()


def test_throw_inside_vocoder_launch_fails_its_batch_and_releases_everything(dims, monkeypatch):
    """a throw inside voc_launch AFTER the sequences left the vocoder queue (here injected: AUR_TEST_FAIL_VOC) used to
    drop the batch without failing it: callers waited forever, latent-pool entries, speaker references and the pinned result
    block leaked.  Now every sequence of the batch is failed, and the engine keeps producing the fresh engine's output."""
    ids = [make_synthetic_text_ids(dims, n_text=10 + k, seed=70 + k) for k in range(2)]
    monkeypatch.delenv("AUR_TEST_FAIL_VOC", raising=False)
    monkeypatch.delenv("AUR_TEST_FAIL_STEP", raising=False)
    ref_e, *_ = make_engine(2, max_seqs=2)
    try:
        ref_e.submit(ids[1], SPK_KEY, temperature=0.0, max_tokens=9, ignore_stop=True)
        ref = ref_e.run_until_done()[0]
    finally:
        ref_e.close()
    monkeypatch.setenv("AUR_TEST_FAIL_VOC", "1")
    e, *_ = make_engine(2, max_seqs=2)
    try:
        sids = [e.submit(i, SPK_KEY, temperature=0.0, max_tokens=6, ignore_stop=True) for i in ids]
        with pytest.raises(AurError) as ei:
            for _ in range(40):
                e.step()
        assert ei.value.code == -2 and "AUR_TEST_FAIL_VOC" in str(ei.value)
        failed = e.poll()
        assert sorted(o["seq_id"] for o in failed) == sids
        assert all(o["error"] == -2 and len(o["wav"]) == 0 for o in failed)
        st = e.stats()
        assert st["kv_blocks_total"] - st["kv_blocks_free"] == 2        # only the speaker's shared prefix blocks
        for _ in range(3):                                               # pool entries / speaker refs / result block came back
            e.submit(ids[1], SPK_KEY, temperature=0.0, max_tokens=9, ignore_stop=True)
            out = e.run_until_done()[0]
            assert out["error"] == 0 and out["tokens"].tolist() == ref["tokens"].tolist() and np.array_equal(out["wav"], ref["wav"])
        e.set_conditioning(4242, np.zeros((1, 32, 1024), np.float32), np.ones((1, 512, 1), np.float32))   # a new voice still fits
    finally:
        e.close()


def test_speaker_table_evicts_lru_idle_voice_and_pins_busy_ones(dims):
    """aur_config.max_speakers voices at most; a new key evicts the least recently used voice without undelivered
    sequences (its prefix KV blocks are recycled), never a busy one; an evicted key must be registered again."""
    e, _, _, cond, spk = make_engine(2, max_seqs=2, max_speakers=2)      # SPK_KEY holds 1 of the 2 rows
    try:
        g = torch.Generator().manual_seed(5)

        def voice():
            c = torch.randn(1, 32, 1024, generator=g) * 0.02
            s = torch.randn(1, 512, 1, generator=g)
            return c.numpy(), (s / s.norm()).numpy()
        v1, v2, v3 = voice(), voice(), voice()
        ids = make_synthetic_text_ids(dims, n_text=12, seed=4)
        e.set_conditioning(1001, *v1)
        e.submit(ids, 1001, temperature=0.0, max_tokens=8, ignore_stop=True)
        first = e.run_until_done()[0]
        assert e.has_conditioning(SPK_KEY) and e.has_conditioning(1001)   # (the queries refresh the LRU stamps: 1001 newest)
        assert e.has_conditioning(SPK_KEY)                                # SPK_KEY newest, 1001 is now the LRU voice
        e.set_conditioning(1002, *v2)                                     # table full -> evicts 1001
        assert not e.has_conditioning(1001) and e.has_conditioning(1002) and e.has_conditioning(SPK_KEY)
        with pytest.raises(AurError):
            e.submit(ids, 1001, max_tokens=4)
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=20, ignore_stop=True)
        e.submit(ids, 1002, temperature=0.0, max_tokens=20, ignore_stop=True)
        e.step()
        with pytest.raises(AurError) as ei:                               # both voices have undelivered sequences
            e.set_conditioning(1003, *v3)
        assert ei.value.code == -3 and "speaker table full" in str(ei.value)
        assert len(e.run_until_done()) == 2
        e.set_conditioning(1001, *v1)                                     # back in, on a recycled row and prefix blocks
        e.submit(ids, 1001, temperature=0.0, max_tokens=8, ignore_stop=True)
        again = e.run_until_done()[0]
        assert again["tokens"].tolist() == first["tokens"].tolist() and np.array_equal(again["wav"], first["wav"])
    finally:
        e.close()


def test_rccl_world1_broadcast_and_device_pointer_registration_equal_host_path(dims):
    """The multi-GPU leg on one GPU: torch.distributed "nccl" (= RCCL) with world_size 1, the conditioning broadcast into a
    device buffer, aur_set_conditioning_device on that buffer: tokens and audio equal the host-pointer registration."""
    import os
    import socket

    import torch.distributed as dist

    from auralis_amd.parallel import broadcast_conditioning
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    e, _, _, cond, spk = make_engine(2, max_seqs=2)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        ids = make_synthetic_text_ids(dims, n_text=15, seed=8)
        e.submit(ids, SPK_KEY, temperature=0.8, top_k=50, top_p=0.85, max_tokens=14, seed=3, ignore_stop=True)
        host = e.run_until_done()[0]
        buf = broadcast_conditioning(e, 777, cond, spk, src=0, device=torch.device("cuda", 0))
        assert buf.is_cuda and buf.numel() == 32 * 1024 + 512 and e.has_conditioning(777)
        e.submit(ids, 777, temperature=0.8, top_k=50, top_p=0.85, max_tokens=14, seed=3, ignore_stop=True)
        dev = e.run_until_done()[0]
        assert dev["tokens"].tolist() == host["tokens"].tolist() and np.array_equal(dev["wav"], host["wav"])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        e.close()


def test_in_library_rccl_communicator_and_broadcast_world1(dims):
    """aur_comm_unique_id / aur_comm_init / aur_broadcast_conditioning: the RCCL communicator lives inside the library (loaded
    with dlopen).  World size 1 is what one GPU can exercise: id creation, ncclCommInitRank, the ncclBroadcast on the engine's
    stream and the error paths; the non-root registration path is the aur_set_conditioning_device code the test above covers."""
    from auralis_amd._lib import AurError, NativeEngine
    e, _, _, cond, spk = make_engine(2, max_seqs=2)
    try:
        with pytest.raises(AurError, match="aur_comm_init"):
            e.broadcast_conditioning(SPK_KEY, 0)
        uid = NativeEngine.comm_unique_id()
        assert isinstance(uid, bytes) and len(uid) == 128 and any(uid)
        with pytest.raises(AurError):
            e.comm_init(uid, 1, 1)                       # rank out of range
        assert e.comm_info() == (0, -1)
        e.comm_init(uid, 0, 1)
        assert e.comm_info() == (1, 0)                   # ncclCommCount / ncclCommUserRank of the communicator itself
        with pytest.raises(AurError, match="already"):
            e.comm_init(uid, 0, 1)
        # checksum of the voice as registered in device memory == FNV-1a of the bytes that were handed in
        x = 1469598103934665603
        for byte in cond.numpy().astype("<f4").tobytes() + spk.numpy().astype("<f4").tobytes():
            x = ((x ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert e.conditioning_checksum(SPK_KEY) == x
        with pytest.raises(AurError):
            e.conditioning_checksum(999999)
        ids = make_synthetic_text_ids(dims, n_text=15, seed=8)
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=10, ignore_stop=True)
        before = e.run_until_done()[0]
        e.broadcast_conditioning(SPK_KEY, 0)             # root == self: the payload goes through ncclBroadcast and back
        with pytest.raises(AurError, match="no such speaker"):
            e.broadcast_conditioning(424242, 0)
        with pytest.raises(AurError, match="root"):
            e.broadcast_conditioning(SPK_KEY, 3)
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=10, ignore_stop=True)
        after = e.run_until_done()[0]
        assert after["tokens"].tolist() == before["tokens"].tolist() and np.array_equal(after["wav"], before["wav"])
    finally:
        e.close()


def test_poll_views_of_the_result_block_equal_owned_copies(ctx, dims):
    """poll(copy=False) hands out views of the engine's pinned result block (aur_result.wav / .tokens), valid until
    release(seq_id); they hold what poll(copy=True) copies, and the blocks are reusable after the release."""
    e = ctx[0]
    ids = make_synthetic_text_ids(dims, n_text=12, seed=21)

    def run(copy):
        for k in range(3):
            e.submit(ids, SPK_KEY, max_tokens=6 + k, temperature=0.0)
        return sorted(e.run_until_done(copy=copy), key=lambda o: len(o["tokens"]))

    owned = run(True)
    for _ in range(2):   # twice: the second pass reuses the blocks released after the first
        views = run(False)
        assert len(views) == len(owned) == 3
        for v, o in zip(views, owned):
            assert v["error"] == 0 and not v["wav"].flags["OWNDATA"]
            assert np.array_equal(v["tokens"], o["tokens"]) and np.array_equal(v["wav"], o["wav"])
        for v in views:
            e.release(v["seq_id"])


def test_profile_mode_changes_nothing_but_the_statistics(ctx, dims):
    """aur_set_profile: the replay batches that time the decode kernels write to scratch only — ids, latents and audio of a ragged,
    over-subscribed, sampled workload are bit-identical with profile mode on and off, and the per-kind statistics are filled."""
    e, _, _, _ = ctx

    def run():
        for k in range(5):
            e.submit(make_synthetic_text_ids(dims, n_text=9 + 7 * k, seed=40 + k), SPK_KEY, temperature=(0.0 if k == 0 else 0.8), top_p=0.9,
                     top_k=40, repetition_penalty=5.0, max_tokens=12 + 5 * k, seed=77 + k, ignore_stop=True)
        return sorted(e.run_until_done(), key=lambda o: o["seq_id"])
    base = run()
    e.set_profile(2)
    e.reset_stats()
    prof = run()
    st = e.stats()
    e.set_profile(0)
    assert len(base) == len(prof) == 5
    for a, b in zip(base, prof):
        assert a["tokens"].tolist() == b["tokens"].tolist()
        assert np.array_equal(a["wav"], b["wav"]) and np.array_equal(a["latents"], b["latents"])
    n_layer = e.n_layer
    assert st["attn_launches"] > 0 and st["attn_launches"] % n_layer == 0 and st["attn_ms"] > 0
    assert all(n > 0 and n % n_layer == 0 for n in st["gemm_kind_launches"]) and all(t > 0 for t in st["gemm_kind_ms"])
    assert st["conv_launches"] > 0 and st["conv_ms"] > 0
    e.reset_stats()
    again = run()
    assert e.stats()["gemm_launches"] == 0                              # off again
    assert [o["tokens"].tolist() for o in again] == [o["tokens"].tolist() for o in base]


def test_admission_groups_prefill_passes_and_changes_no_output(dims):
    """aur_config.admit_min_batch: with a queue longer than the free slots the engine holds admission until a group of slots is
    free, so that one prefill pass serves several prompts.  24 ragged sequences (natural stop, sampled) on 6 slots: the grouped
    engine runs fewer prefill passes than the one-by-one engine, every sequence finishes, and ids / latents / audio are equal bit
    for bit (a sequence never depends on what it was admitted with)."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.weights import pack_all
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 2.0
    packed = pack_all(gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd))
    cond, spk = make_synthetic_conditioning(dims)

    def run(group):
        e = NativeEngine(n_layer=2, max_seqs=6, admit_min_batch=group)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
            sids = [e.submit(make_synthetic_text_ids(dims, n_text=8 + (5 * k) % 23, seed=60 + k), SPK_KEY, temperature=0.8, top_k=50, top_p=0.85,
                             repetition_penalty=5.0, max_tokens=4 + (7 * k) % 29, seed=300 + k, ignore_stop=(k % 5 == 0)) for k in range(24)]
            outs = {o["seq_id"]: o for o in e.run_until_done()}
            assert sorted(outs) == sorted(sids) and all(o["error"] == 0 for o in outs.values())
            return [outs[s] for s in sids], e.stats()["prefill_batches"]
        finally:
            e.close()

    one, n_one = run(1)
    grp, n_grp = run(3)
    dflt, n_dflt = run(9)        # larger than the slot count: capped, i.e. admission only when the engine has drained
    print("prefill passes: one by one", n_one, "groups of 3", n_grp, "whole engine", n_dflt)
    assert n_grp < n_one and n_dflt <= n_grp and n_dflt >= 4
    for a, b, c in zip(one, grp, dflt):
        assert a["tokens"].tolist() == b["tokens"].tolist() == c["tokens"].tolist()
        assert np.array_equal(a["wav"], b["wav"]) and np.array_equal(a["wav"], c["wav"]) and np.array_equal(a["latents"], b["latents"])


def test_admission_hold_is_bounded(dims):
    """A hold by admit_min_batch ends after 32 steps: one long sequence keeps a slot, the other slot is free, two requests wait for a
    group of two slots that will not be free for 120 steps -- they are admitted after the bound, not after the long sequence."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.weights import pack_all
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    packed = pack_all(gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd))
    cond, spk = make_synthetic_conditioning(dims)

    def run(group):
        e = NativeEngine(n_layer=2, max_seqs=2, admit_min_batch=group)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
            sids = [e.submit(make_synthetic_text_ids(dims, n_text=10 + k, seed=80 + k), SPK_KEY, temperature=0.0, max_tokens=n, seed=1, ignore_stop=True)
                    for k, n in enumerate([120, 2, 5, 5])]
            outs = {o["seq_id"]: o for o in e.run_until_done()}
            st = e.stats()
            return [outs[s] for s in sids], st["decode_steps"], st["prefill_batches"]
        finally:
            e.close()

    one, steps_one, _ = run(1)
    grp, steps_grp, passes = run(2)
    print("decode steps: one by one", steps_one, "groups of two", steps_grp, "prefill passes", passes)
    assert steps_grp <= 123 and passes >= 3          # without the bound: 120 + 5 steps and the two short requests behind the long one
    for a, b in zip(one, grp):
        assert a["tokens"].tolist() == b["tokens"].tolist() and np.array_equal(a["wav"], b["wav"])


def test_admission_urgent_sequence_jumps_the_queue_and_changes_no_output(dims):
    """aur_seq_desc.priority: 12 sequences submitted at once to 8 slots, the LAST one marked latency-critical.  It is admitted in the
    first prefill pass (in front of the queue), the engine fills only aur_config.urgent_rows slots while it runs, it is vocoded as soon
    as its tokens are done -- so it is the first result out -- and no sequence's ids, latents or audio depend on any of that."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.weights import pack_all
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    packed = pack_all(gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd))
    cond, spk = make_synthetic_conditioning(dims)

    def run(priority):
        e = NativeEngine(n_layer=2, max_seqs=8, urgent_rows=3, vocoder_min_batch=4)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
            sids = [e.submit(make_synthetic_text_ids(dims, n_text=9 + k, seed=90 + k), SPK_KEY, temperature=0.8, top_k=50, top_p=0.85, repetition_penalty=5.0,
                             max_tokens=(10 if k == 11 else 40), seed=500 + k, ignore_stop=True, priority=(priority if k == 11 else 0)) for k in range(12)]
            order, rows = [], []
            for _ in range(2000):
                live, _fin = e.step()
                st = e.stats()
                rows.append(st["decode_rows"])
                order.extend(o for o in e.poll())
                if live == 0:
                    break
            assert len(order) == 12 and all(o["error"] == 0 for o in order)
            per_step = np.diff([0] + rows)
            return {o["seq_id"]: o for o in order}, [o["seq_id"] for o in order], sids, per_step
        finally:
            e.close()

    plain, order_p, sids_p, rows_p = run(0)
    urg, order_u, sids_u, rows_u = run(1)
    assert order_u[0] == sids_u[11] and order_p[0] != sids_p[11]       # first out when urgent; behind the first wave otherwise
    assert rows_u[:8].max() <= 3 and rows_p[:8].max() == 8              # the engine fills 3 of its 8 slots while the urgent one runs
    assert rows_u.max() == 8                                           # ... and all of them afterwards
    for a, b in zip(sids_p, sids_u):
        assert plain[a]["tokens"].tolist() == urg[b]["tokens"].tolist()
        assert np.array_equal(plain[a]["wav"], urg[b]["wav"]) and np.array_equal(plain[a]["latents"], urg[b]["latents"])


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_cancel_stops_a_sequence_frees_its_resources_and_changes_nobody_else(dims, monkeypatch, pipeline):
    """aur_cancel: ten long sequences on six slots; after a few steps two running ones and one waiting one are cancelled.  The waiting
    one is reported at once, the running ones stop within a few decode steps (far short of their 150 tokens) and are not vocoded; all
    three come back with error AUR_E_CANCELLED and no audio.  The other seven finish with ids, latents and audio equal bit for bit to a
    run without any cancellation; slots, K/V blocks and the latent pool are all back."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.weights import pack_all
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    packed = pack_all(gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd))
    cond, spk = make_synthetic_conditioning(dims)

    monkeypatch.setenv("AUR_DECODE_PIPELINE", pipeline)     # (read when the engine is created) pipelined and synchronous decode loop

    def run(cancel):
        e = NativeEngine(n_layer=2, max_seqs=6)
        try:
            e.load_weights(packed)
            e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
            sids = [e.submit(make_synthetic_text_ids(dims, n_text=9 + k, seed=120 + k), SPK_KEY, temperature=0.8, top_k=50, top_p=0.85,
                             repetition_penalty=5.0, max_tokens=150, seed=700 + k, ignore_stop=True) for k in range(10)]
            got = {}
            for i in range(4000):
                live, _ = e.step()
                if cancel and i == 5:
                    for k in (1, 4, 8):          # 1 and 4 are running (slots 0..5 hold sequences 0..5), 8 still waits
                        e.cancel(sids[k])
                    e.cancel(sids[8])            # twice: harmless
                for o in e.poll():
                    got[o["seq_id"]] = o
                if live == 0:
                    break
            assert len(got) == 10
            st = e.stats()
            assert st["kv_blocks_total"] - st["kv_blocks_free"] == 2 and st["sequences_tracked"] == 0
            return [got[s] for s in sids]
        finally:
            e.close()

    base = run(False)
    cut = run(True)
    for k in range(10):
        if k in (1, 4, 8):
            assert cut[k]["error"] == -5 and len(cut[k]["wav"]) == 0
            assert len(cut[k]["tokens"]) <= (12 if k != 8 else 0), (k, len(cut[k]["tokens"]))
            assert cut[k]["tokens"].tolist() == base[k]["tokens"][: len(cut[k]["tokens"])].tolist()   # what it did generate is what it would have
        else:
            assert cut[k]["error"] == 0 and cut[k]["tokens"].tolist() == base[k]["tokens"].tolist()
            assert np.array_equal(cut[k]["wav"], base[k]["wav"]) and np.array_equal(cut[k]["latents"], base[k]["latents"])
    with pytest.raises(AurError, match="unknown seq_id"):
        e2 = NativeEngine(n_layer=2, max_seqs=2)
        try:
            e2.cancel(12345)
        finally:
            e2.close()


def test_kv_pool_above_the_32_bit_offset_range_is_refused_at_creation():
    """paged_attention_kernel addresses a layer's K/V pool with 32-bit byte offsets: a pool of 4 GiB or more per
    layer -- 495 slots with the fp32 pool, 992 with fp16 -- must be refused when the engine is created, not read past 4 GiB later."""
    from auralis_amd._lib import NativeEngine
    with pytest.raises(AurError, match="4 GiB"):
        NativeEngine(n_layer=1, max_seqs=495)
    with pytest.raises(AurError, match="4 GiB"):
        NativeEngine(n_layer=1, max_seqs=992, kv_fp16=True)
