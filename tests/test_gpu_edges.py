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
