"""Golden vectors produced by the REFERENCE's own GPT glue (oracle/make_golden_gpt.py: GPT2Model.forward,
LearnedPositionEmbeddings, LogitsRepetitionPenalizer executed from /root/reference; vllm's GPT2Block replaced by the
textbook block built from transformers modules).  CPU: the oracle restatement against them, prefill and incremental
decode; GPU: the HIP prefill through the C ABI against them."""
import os

import numpy as np
import pytest
import torch

from oracle import xtts_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gpt_glue_L2.npz"))
N_TEXT = len(G["text_ids"])
N_COND = 32 + N_TEXT


@pytest.fixture(scope="module")
def oracle_ctx(dims):
    from auralis_amd.checkpoint import make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_xtts
    assert int(G["weights_seed"]) == 1234 and int(G["n_layer"]) == 2
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=2)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    cond, _ = make_synthetic_conditioning(dims)
    return O.GPTOracle(gpt_sd, xtts_sd), cond


def test_prompt_rows_match_reference_text_embedding(oracle_ctx):
    gpt, cond = oracle_ctx
    c = gpt.build_cond(cond, G["text_ids"].tolist())
    assert c.shape == (N_COND, 1024)
    assert np.array_equal(c[32:].numpy(), G["cond_text_rows"])          # two table lookups and one add: exact


def test_prefill_and_teacher_forced_rows_match_reference_gpt2model(oracle_ctx):
    gpt, cond = oracle_ctx
    toks = G["tokens"].tolist()
    c = gpt.build_cond(cond, G["text_ids"].tolist())
    x = torch.cat([c, gpt.mel_embed([1024] + toks, list(range(len(toks) + 1)))], dim=0)
    h, _ = gpt.forward_rows(x, None)
    ref = torch.from_numpy(G["ln_f_rows"])
    assert h.shape == ref.shape and (h - ref).abs().max().item() < 2e-5
    z = torch.stack([gpt.logits(r) for r in h[N_COND:]])
    assert (z - torch.from_numpy(G["logits_gen_rows"])).abs().max().item() < 1e-4


def test_incremental_decode_reproduces_reference_rows(oracle_ctx):
    """The reference feeds generated tokens one per step (token k at mel position k); the oracle's KV-cached decode must
    land on the same ln_f rows as the reference's single pass over [prompt ; start ; tokens]."""
    gpt, cond = oracle_ctx
    toks = G["tokens"].tolist()
    c = gpt.build_cond(cond, G["text_ids"].tolist())
    h, cache = gpt.forward_rows(torch.cat([c, gpt.mel_embed([1024], [0])], dim=0), None)
    rows = [h[-1]]
    for k, t in enumerate(toks, start=1):
        h, cache = gpt.forward_rows(gpt.mel_embed([t], [k]), cache)
        rows.append(h[-1])
    ref = torch.from_numpy(G["ln_f_rows"])[N_COND:]
    assert (torch.stack(rows) - ref).abs().max().item() < 2e-5


def test_repetition_penalty_matches_reference_processor():
    z = O.apply_repetition_penalty(torch.from_numpy(G["pen_logits"].copy()),
                                   sorted(set(G["pen_prompt_ids"].tolist()) | set(G["pen_output_ids"].tolist())), 5.0)
    assert np.array_equal(z.numpy(), G["pen_result"])


@pytest.mark.gpu
def test_hip_prefill_matches_reference_gpt2model():
    from tests.gpu_util import SPK_KEY, make_engine
    e, *_ = make_engine(2, max_seqs=2)
    try:
        rows, logits = e.dbg_prefill(G["text_ids"].tolist(), SPK_KEY, repetition_penalty=1.0)
        ref = G["ln_f_rows"][: N_COND + 1]
        assert rows.shape == ref.shape and np.abs(rows - ref).max() < 2e-4
        assert np.abs(logits - G["logits_gen_rows"][0]).max() < 5e-4
        assert int(np.argmax(logits)) == int(np.argmax(G["logits_gen_rows"][0]))
    finally:
        e.close()


from oracle.ref_gpt_import import reference_gpt_available  # noqa: E402


@pytest.mark.skipif(not reference_gpt_available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("n_layer,n_text,n_tok", [(1, 3, 1), (3, 17, 9)])
def test_oracle_against_live_reference_gpt2model(dims, n_layer, n_text, n_tok):
    """Other depths / lengths than the committed fixture, with the reference module executed on the spot."""
    from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_text_ids,
                                        make_synthetic_xtts)
    from oracle.ref_gpt_import import build_reference_gpt2model
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=77, n_layer=n_layer)
    xtts_sd = make_synthetic_xtts(dims, seed=77, gpt_sd=gpt_sd)
    gpt = O.GPTOracle(gpt_sd, xtts_sd)
    cond, _ = make_synthetic_conditioning(dims)
    ids = list(make_synthetic_text_ids(dims, n_text=n_text, seed=3))
    toks = torch.randint(0, 1024, (n_tok,), generator=torch.Generator().manual_seed(n_tok)).tolist()
    c = gpt.build_cond(cond, ids)
    model = build_reference_gpt2model(gpt_sd, n_layer)
    with torch.no_grad():
        ref = model(input_ids=torch.tensor(toks), position_ids=torch.arange(1, n_tok + 1), kv_caches=[None] * n_layer,
                    attn_metadata=None, intermediate_tensors=None, input_embeds=[c[None]], starting_sequence_start_ids=[0],
                    is_profiling_run=False, is_logit_only=torch.tensor([False]))
    x = torch.cat([c, gpt.mel_embed([1024] + toks, list(range(n_tok + 1)))], dim=0)
    h, _ = gpt.forward_rows(x, None)
    assert h.shape == ref.shape and (h - ref).abs().max().item() < 3e-5
