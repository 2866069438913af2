"""Long-form / mixed-language streaming (BASELINE config 5 at test scale): CPU with the fake engine, GPU with the real one."""
import numpy as np
import pytest

from auralis_amd import TTS, TTSOutput
from auralis_amd.api.text import CHAR_LIMITS, XTTSTokenizer, split_sentence
from auralis_amd.api.xtts_engine import XTTSv2Engine
from auralis_amd.longform import build_requests, split_paragraphs, stream_longform
from tests.fakes import FakeNativeEngine

EN = ("It was a bright cold day in April, and the clocks were striking thirteen. Nobody in the street seemed to notice, "
      "and the wind kept pushing the dust along the old road as if nothing had happened at all. " * 2)
FR = ("Il était une fois, dans une petite ville que nous ne connaissons pas, un homme qui avait beaucoup d'idées et très peu "
      "de temps pour les écrire. Il marchait chaque matin le long de la rivière avec son chien. " * 2)
DE = ("Es war einmal ein Mann, der nicht mit dem Zug fahren wollte und auch nicht zu Fuß gehen konnte, weil der Weg durch "
      "den Wald zu lang war. Also blieb er zu Hause und schrieb Briefe an seine Freunde. " * 2)
BOOK = "\n\n".join([EN, FR, DE, EN[:120], FR[:150], DE[:90]])
VOICE = {"gpt_cond_latent": np.zeros((1, 32, 1024), np.float32), "speaker_embedding": np.ones((1, 512, 1), np.float32)}


def _expected_chunks(reqs):
    return [len(split_sentence(r.text, r.language, CHAR_LIMITS[r.language])) for r in reqs]


def test_longform_requests_and_order_cpu():
    paras = split_paragraphs(BOOK)
    reqs = build_requests(paras, [VOICE], seed=7)
    assert [r.language for r in reqs] == ["en", "fr", "de", "en", "fr", "de"]
    fake = FakeNativeEngine(max_seqs=3)
    tts = TTS(scheduler_max_concurrency=3).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    try:
        got = list(stream_longform(tts, reqs, window=3))
        idx = [i for i, _ in got]
        assert idx == sorted(idx)                                       # paragraph order preserved
        counts = [idx.count(i) for i in range(len(reqs))]
        assert counts == _expected_chunks(reqs) and sum(counts) == len(fake.submitted)
        assert all(isinstance(c, TTSOutput) and len(c.array) > 0 for _, c in got)
    finally:
        tts.close()


@pytest.mark.gpu
def test_longform_mixed_languages_gpu(tmp_path, dims):
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts, save_checkpoint
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    gpt_sd["mel_head.bias"][1025] = 3.0          # natural stops after a handful of tokens
    save_checkpoint(str(tmp_path), gpt_sd, make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd), dims, synthetic_tokenizer=True)
    cond, spk = make_synthetic_conditioning(dims)
    voice = {"gpt_cond_latent": cond.numpy(), "speaker_embedding": spk.numpy()}
    reqs = build_requests(split_paragraphs(BOOK), [voice], seed=3, temperature=0.0)
    tts = TTS(scheduler_max_concurrency=8).from_pretrained(str(tmp_path))
    try:
        got = list(stream_longform(tts, reqs, window=4))
        idx = [i for i, _ in got]
        assert idx == sorted(idx) and [idx.count(i) for i in range(len(reqs))] == _expected_chunks(reqs)
        audio = TTSOutput.combine_outputs([c for _, c in got])
        assert len(audio.array) > 0 and np.isfinite(audio.array).all()
        again = TTSOutput.combine_outputs([c for _, c in stream_longform(tts, reqs, window=2)])
        assert np.array_equal(audio.array, again.array)                 # greedy: independent of the window / batching
    finally:
        tts.close()


def test_a_closed_stream_cancels_the_chunks_nobody_will_read():
    """A streaming consumer takes the first chunk of a long request and goes away (an HTTP client that disconnects): the chunks still
    queued or decoding are cancelled in the engine -- also those whose pump was still waiting for a scheduler slot and never looked at
    its chunk -- and the facade keeps serving."""
    from auralis_amd import TTSRequest
    fake = FakeNativeEngine(max_seqs=1, step_delay=0.02)
    tts = TTS(scheduler_max_concurrency=2).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    try:
        gen = tts.generate_speech(TTSRequest(text=" ".join([EN] * 6), speaker_files=[VOICE], language="en", stream=True, seed=3))
        first = next(gen)
        assert len(first.array) > 0
        n = len(fake.submitted)
        assert n >= 10
        gen.close()
        import time
        t0 = time.time()
        while (fake.finished_total < n or tts.tts_engine.driver._pending) and time.time() - t0 < 10:   # (the driver polls after its next step)
            time.sleep(0.01)
        assert len(fake.cancelled) >= n - 4 and len(set(fake.cancelled)) == len(fake.cancelled), (n, fake.cancelled, fake.finished_total)
        assert not fake.waiting and not fake.running
        assert not tts.tts_engine.driver._pending
        out = tts.generate_speech(TTSRequest(text="Still here.", speaker_files=[VOICE], language="en"))
        assert len(out.array) > 0
    finally:
        tts.close()


def test_book_scale_stream_host_overhead():
    """BASELINE config 5 size (~450 k characters, > 2 000 chunks) through the facade with the fake engine: every chunk comes
    back once and in order, and the host path stays far below the GPU's per-chunk time (~14 ms per chunk per GPU)."""
    import time
    paras = []
    while sum(len(p) for p in paras) < 450_000:
        paras += [EN, FR, DE]
    reqs = build_requests(paras, [VOICE], seed=1)
    fake = FakeNativeEngine(max_seqs=64)
    tts = TTS(scheduler_max_concurrency=64).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    try:
        t0 = time.perf_counter()
        idx = [i for i, _ in stream_longform(tts, reqs, window=32)]
        dt = time.perf_counter() - t0
    finally:
        tts.close()
    assert idx == sorted(idx) and len(idx) == sum(_expected_chunks(reqs)) == len(fake.submitted) > 2000
    assert dt / len(idx) < 5e-3, dt / len(idx)


def _sharded_worker(rank, world, port, out_dir):
    import os

    import torch.distributed as dist

    from auralis_amd.longform import synthesize_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    paras = [EN, FR, DE] * 9 + [EN[:100]]
    reqs = build_requests(paras, [VOICE], seed=1)
    fake = FakeNativeEngine(max_seqs=4)
    tts = TTS(scheduler_max_concurrency=4).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    import time

    from auralis_amd.longform import stream_sharded
    try:
        out = synthesize_sharded(tts, reqs, window=3, paragraphs_per_block=4)
        # streamed form: rank 0 must hand out the first chunk long before the last one exists anywhere
        t0 = time.perf_counter()
        stamps, idx, total = [], [], 0
        for i, pcm in stream_sharded(tts, reqs, window=3, paragraphs_per_block=4):
            stamps.append(time.perf_counter() - t0)
            idx.append(i)
            total += len(pcm)
        t_end = time.perf_counter() - t0
    finally:
        tts.close()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), n=np.int64(-1 if out is None else len(out.array)),
             submitted=np.int64(len(fake.submitted)), idx=np.asarray(idx, np.int64), stamps=np.asarray(stamps),
             t_end=np.float64(t_end), streamed=np.int64(total))
    dist.destroy_process_group()


def test_sharded_book_on_two_ranks_gloo(tmp_path):
    """world_size 2 over gloo: the two ranks split the paragraphs, rank 0 gets the whole book back in order."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    paras = [EN, FR, DE] * 9 + [EN[:100]]
    reqs = build_requests(paras, [VOICE], seed=1)
    fake = FakeNativeEngine(max_seqs=4)
    tts = TTS(scheduler_max_concurrency=4).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    try:
        single = TTSOutput.combine_outputs([c for _, c in stream_longform(tts, reqs, window=3)])
    finally:
        tts.close()
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert int(r1["n"]) == -1 and int(r0["n"]) == len(single.array)           # same audio length as one process
    assert int(r0["submitted"]) + int(r1["submitted"]) == 2 * len(fake.submitted)  # (two passes) every chunk exactly once per pass
    assert int(r0["submitted"]) > 0 and int(r1["submitted"]) > 0
    # streamed pass: chunks of all 28 paragraphs in paragraph order on rank 0, nothing on rank 1, same amount of audio,
    # and the first chunk is out well before the book is finished (ordered re-emission, not an end-of-book gather)
    idx = r0["idx"].tolist()
    assert idx == sorted(idx) and sorted(set(idx)) == list(range(len(paras))) and len(r1["idx"]) == 0
    assert len(idx) == sum(_expected_chunks(reqs)) and int(r0["streamed"]) == len(single.array)
    assert r0["stamps"][0] < 0.5 * float(r0["t_end"])


def _sharded_failure_worker(rank, world, port, out_dir, mode):
    """mode "remote_error": rank 1's engine fails after a few steps; mode "early_stop": rank 0 stops consuming after 3 chunks"""
    import os

    import torch.distributed as dist

    from auralis_amd.longform import stream_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    paras = [EN, FR, DE] * 6
    reqs = build_requests(paras, [VOICE], seed=1)
    fake = FakeNativeEngine(max_seqs=4, fail_on_step=(4 if (mode == "remote_error" and rank == 1) else None))
    tts = TTS(scheduler_max_concurrency=4).with_engine(XTTSv2Engine(fake, XTTSTokenizer(None, synthetic=True)))
    got, err = 0, ""
    try:
        it = stream_sharded(tts, reqs, window=3, paragraphs_per_block=2)
        try:
            for _ in it:
                got += 1
                if mode == "early_stop" and rank == 0 and got == 3:
                    break
        finally:
            it.close()                      # a consumer that walks away: the generator's cleanup must release the other ranks
    except BaseException as e:
        err = f"{type(e).__name__}: {e}"
    finally:
        tts.close()
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{got}\n{err}\n")
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["remote_error", "early_stop"])
def test_sharded_stream_never_hangs_on_a_failing_rank_or_a_consumer_that_stops(tmp_path, mode):
    """the point-to-point protocol had no error message (a failing non-dst rank left dst in recv forever) and no way
    out for senders when the consumer on dst stopped (they blocked in send).  Both cases must END, with the remote error
    text arriving on dst."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.spawn(_sharded_failure_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=1.0):
        assert time.time() - t0 < 90, "the sharded stream hung"
    r0 = open(tmp_path / "r0.txt").read().split("\n")
    r1 = open(tmp_path / "r1.txt").read().split("\n")
    if mode == "remote_error":
        assert "RuntimeError" in r0[1] and "rank 1" in r0[1] and "injected engine failure" in r0[1], r0
        assert "injected engine failure" in r1[1], r1          # the failing rank re-raises its own error
    else:
        assert int(r0[0]) == 3 and r0[1] == "" and r1[1] == "", (r0, r1)
