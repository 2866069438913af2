"""GPU: GPT token loop (prefill, paged decode, fused sampler, latent stash) and the whole engine vs the oracle."""
import numpy as np
import pytest
import torch

from auralis_amd.checkpoint import make_synthetic_text_ids
from tests.gpu_util import SPK_KEY, make_engine, rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    e, gpt_sd, xtts_sd, cond, spk = make_engine(3, max_seqs=4)
    from oracle import xtts_oracle as O
    yield e, O.GPTOracle(gpt_sd, xtts_sd), xtts_sd, cond, spk
    e.close()


def test_prefill_hidden_and_logits(small, dims):
    e, gpt, _, cond, _ = small
    ids = make_synthetic_text_ids(dims, n_text=20)
    rows, logits = e.dbg_prefill(ids, SPK_KEY, repetition_penalty=1.0)
    c = gpt.build_cond(cond, ids)
    x = torch.cat([c, gpt.mel_embed([1024], [0])], dim=0)
    h, _ = gpt.forward_rows(x, None)
    assert rows.shape == tuple(h.shape)
    assert np.abs(rows - h.numpy()).max() < 2e-4
    z = gpt.logits(h[-1]).numpy()
    assert np.abs(logits - z).max() < 2e-4


def test_prefill_presplit_operands_equal_split_on_the_fly(small, dims, monkeypatch):
    """Prompt rows: the GEMMs read their weights as the bf16 planes packed at load time; AUR_GEMM_PRESPLIT=0 splits the fp32 weights
    inside the GEMM.  Hidden states of every prompt row, the logits, and what the decode steps then read from the K / V pages
    (written by the QKV GEMM's epilogue) are equal bit for bit."""
    from tests.gpu_util import make_engine
    e, *_ = small
    monkeypatch.setenv("AUR_GEMM_PRESPLIT", "0")
    e2, *_ = make_engine(e.n_layer, max_seqs=4)
    try:
        for n_text in (20, 131):
            ids = make_synthetic_text_ids(dims, n_text=n_text, seed=3 + n_text)
            r1, l1 = e.dbg_prefill(ids, SPK_KEY, repetition_penalty=1.0)
            r2, l2 = e2.dbg_prefill(ids, SPK_KEY, repetition_penalty=1.0)
            assert np.array_equal(r1, r2) and np.array_equal(l1, l2), n_text
        ids = make_synthetic_text_ids(dims, n_text=40, seed=8)
        outs = []
        for eng in (e, e2):
            eng.submit(ids, SPK_KEY, temperature=0.8, top_p=0.9, top_k=40, repetition_penalty=5.0, max_tokens=24, seed=5, ignore_stop=True)
            outs.append(eng.run_until_done()[0])
        assert outs[0]["tokens"].tolist() == outs[1]["tokens"].tolist() and np.array_equal(outs[0]["latents"], outs[1]["latents"])
    finally:
        e2.close()


def _greedy_case(e, gpt, cond, dims, n_text, max_tokens, ignore_stop=True):
    from oracle import xtts_oracle as O
    ids = make_synthetic_text_ids(dims, n_text=n_text)
    c = gpt.build_cond(cond, ids)
    ref = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=max_tokens, ignore_stop=ignore_stop), return_debug=True)
    e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=max_tokens, ignore_stop=ignore_stop)
    out = e.run_until_done()
    assert len(out) == 1
    return ids, c, ref, out[0]


def test_greedy_tokens_bit_exact_and_latents(small, dims):
    """mel-token ids bit-exact under greedy; stashed latents == literal second pass (XTTSv2.py:617-687)."""
    e, gpt, xtts_sd, cond, spk = small
    ids, c, ref, got = _greedy_case(e, gpt, cond, dims, n_text=16, max_tokens=40)
    assert got["tokens"].tolist() == ref["tokens"], (min(ref["margins"]), got["tokens"][:8], ref["tokens"][:8])
    lat_ref = gpt.second_pass_latents(c, ref["tokens"])[0].numpy()
    assert got["latents"].shape == lat_ref.shape
    assert np.abs(got["latents"] - lat_ref).max() < 2e-3
    # waveform: engine output vs oracle vocoder on the oracle's second-pass latents (north_star 1e-3 RMS)
    from oracle import xtts_oracle as O
    w = O.vocoder_effective_weights(xtts_sd)
    wav_ref = O.hifi_decoder_forward(w, torch.from_numpy(lat_ref)[None], spk).reshape(-1).numpy()
    err, sig = rms(got["wav"] - wav_ref), rms(wav_ref)
    assert got["wav"].shape == wav_ref.shape
    assert err <= 1e-3 and err <= 1e-2 * sig, (err, sig)


def test_stop_token_natural_mode(dims):
    """stop id ends the sequence and stays in token_ids (XTTSv2.py:737)."""
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_conditioning
    from oracle import xtts_oracle as O
    from tests.gpu_util import packed_weights
    packed, gpt_sd, xtts_sd = packed_weights(3)
    packed = dict(packed)
    hb = packed["mel_head.b"].copy()
    hb[1025] = 4.0          # makes the stop id win after a few steps under the repetition penalty
    packed["mel_head.b"] = hb
    sd = {k: v.clone() for k, v in gpt_sd.items()}
    sd["mel_head.bias"][1025] = 4.0
    gpt = O.GPTOracle(sd, xtts_sd)
    e = NativeEngine(n_layer=3, max_seqs=2)
    try:
        e.load_weights(packed)
        cond, spk = make_synthetic_conditioning(dims)
        e.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
        ids = make_synthetic_text_ids(dims, n_text=10)
        ref = gpt.generate(gpt.build_cond(cond, ids), O.SamplingCfg(temperature=0.0, max_tokens=30))
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=30)
        got = e.run_until_done()[0]
        assert got["tokens"].tolist() == ref["tokens"]
        assert got["tokens"][-1] == 1025 or len(ref["tokens"]) == 30
    finally:
        e.close()


def test_sampled_tokens_match_oracle_with_shared_noise(small, dims):
    from oracle import xtts_oracle as O
    e, gpt, _, cond, _ = small
    ids = make_synthetic_text_ids(dims, n_text=12)
    c = gpt.build_cond(cond, ids)
    cfg = O.SamplingCfg(temperature=0.75, top_k=50, top_p=0.85, repetition_penalty=5.0, max_tokens=24, ignore_stop=True, seed=77)
    ref = gpt.generate(c, cfg)
    e.submit(ids, SPK_KEY, temperature=0.75, top_k=50, top_p=0.85, repetition_penalty=5.0, max_tokens=24, seed=77, ignore_stop=True)
    got = e.run_until_done()[0]
    assert got["tokens"].tolist() == ref["tokens"]


def test_continuous_batching_is_batch_invariant(small, dims):
    """Different prompts/lengths admitted together and late; each sequence equals its solo run bit for bit."""
    e, gpt, _, cond, _ = small
    specs = [(9, 12, 0), (20, 20, 1), (14, 7, 2), (30, 16, 3), (11, 18, 4), (17, 9, 5)]   # 6 seqs > 4 slots; prefill rows > 128
    solo = {}
    for n_text, mt, seed in specs:
        ids = make_synthetic_text_ids(dims, n_text=n_text, seed=100 + seed)
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=mt, ignore_stop=True)
        solo[seed] = e.run_until_done()[0]
    sid = {}
    for n_text, mt, seed in specs:
        ids = make_synthetic_text_ids(dims, n_text=n_text, seed=100 + seed)
        sid[e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=mt, ignore_stop=True)] = seed
    outs = e.run_until_done()
    assert len(outs) == len(specs)
    for o in outs:
        s = solo[sid[o["seq_id"]]]
        assert o["tokens"].tolist() == s["tokens"].tolist()
        assert np.array_equal(o["wav"], s["wav"])
    st = e.stats()
    assert st["kv_blocks_total"] - st["kv_blocks_free"] in (0, 2)     # only the speaker's 2 shared prefix blocks stay allocated


def test_full_depth_greedy_short(dims):
    """All 30 layers at true shapes: greedy ids bit-exact for a short run (oracle ~40 ms/token on CPU)."""
    from oracle import xtts_oracle as O
    e, gpt_sd, xtts_sd, cond, spk = make_engine(30, max_seqs=2)
    try:
        gpt = O.GPTOracle(gpt_sd, xtts_sd)
        ids, c, ref, got = _greedy_case(e, gpt, cond, dims, n_text=24, max_tokens=24)
        assert got["tokens"].tolist() == ref["tokens"], min(ref["margins"])
        lat_ref = gpt.latents_from_decode_rows(ref["decode_rows"], len(ref["tokens"]))[0].numpy()
        assert np.abs(got["latents"] - lat_ref).max() < 5e-3
    finally:
        e.close()


def test_literal_second_pass_mode_matches_stash_and_oracle(dims):
    """aur_config.second_pass = 1 recomputes the latents the reference's way (XTTSv2.py:617-687) on the GPU: same
    tokens, latents equal to the decode-time stash within fp32 reordering noise, and to the oracle's second pass."""
    from oracle import xtts_oracle as O
    e1, gpt_sd, xtts_sd, cond, spk = make_engine(3, max_seqs=3)
    e2, *_ = make_engine(3, max_seqs=3, second_pass=True)
    try:
        gpt = O.GPTOracle(gpt_sd, xtts_sd)
        outs = []
        for e in (e1, e2):
            for n_text, mt, seed in ((16, 30, 1), (9, 12, 2), (25, 21, 3)):
                e.submit(make_synthetic_text_ids(dims, n_text=n_text, seed=seed), SPK_KEY, temperature=0.0, max_tokens=mt, ignore_stop=True)
            outs.append(sorted(e.run_until_done(), key=lambda o: o["seq_id"]))
        for a, b in zip(*outs):
            assert a["tokens"].tolist() == b["tokens"].tolist()
            assert np.abs(a["latents"] - b["latents"]).max() < 2e-3
            assert rms(a["wav"] - b["wav"]) < 1e-4
        ids = make_synthetic_text_ids(dims, n_text=16, seed=1)
        lat_ref = gpt.second_pass_latents(gpt.build_cond(cond, ids), outs[1][0]["tokens"].tolist())[0].numpy()
        assert np.abs(outs[1][0]["latents"] - lat_ref).max() < 1e-3
    finally:
        e1.close()
        e2.close()


def test_erf_gelu_checkpoint_runs_the_erf_kernels(dims):
    """gpt/config.json "activation_function": "gelu" -> aur_config.gelu_erf: prompt rows (gemm_tile epilogue) and decode rows
    (gemm_rows epilogue) both apply the erf form; ids bit-exact against the oracle built with the same activation, and
    different from what the tanh-form engine emits on the same prompt (the test can tell the two apart)."""
    from oracle import xtts_oracle as O
    e, gpt_sd, xtts_sd, cond, spk = make_engine(3, max_seqs=2, gelu_erf=True)
    e_tanh, *_ = make_engine(3, max_seqs=2)
    try:
        gpt = O.GPTOracle(gpt_sd, xtts_sd, activation="gelu")
        ids = make_synthetic_text_ids(dims, n_text=18, seed=31)
        c = gpt.build_cond(cond, ids)
        ref = gpt.generate(c, O.SamplingCfg(temperature=0.0, max_tokens=32, ignore_stop=True), return_debug=True)
        e.submit(ids, SPK_KEY, temperature=0.0, max_tokens=32, ignore_stop=True)
        got = e.run_until_done()[0]
        assert got["tokens"].tolist() == ref["tokens"], min(ref["margins"])
        lat_ref = gpt.latents_from_decode_rows(ref["decode_rows"], len(ref["tokens"]))[0].numpy()
        assert np.abs(got["latents"] - lat_ref).max() < 2e-3
        e_tanh.submit(ids, SPK_KEY, temperature=0.0, max_tokens=32, ignore_stop=True)
        other = e_tanh.run_until_done()[0]
        assert np.abs(other["latents"] - got["latents"]).max() > 1e-3
    finally:
        e.close()
        e_tanh.close()
