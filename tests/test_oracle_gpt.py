"""CPU: cross-checks of the GPT restatement (oracle is 'parity unpinned' for the GPT: the arithmetic lives in the
un-vendored vllm==0.6.4.post1; see oracle/xtts_oracle.py header)."""
import numpy as np
import pytest
import torch

from auralis_amd.checkpoint import make_synthetic_text_ids
from oracle import xtts_oracle as O


def _hf_gpt2(gpt_sd, n_layer, activation="gelu_new"):
    from transformers import GPT2Config, GPT2Model
    cfg = GPT2Config(vocab_size=8, n_positions=1100, n_embd=1024, n_layer=n_layer, n_head=16, n_inner=4096,
                     activation_function=activation, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                     layer_norm_epsilon=1e-5)
    m = GPT2Model(cfg).eval()
    sd = m.state_dict()
    for k in list(sd.keys()):
        src = "gpt." + k
        if k.startswith("h.") or k.startswith("ln_f."):
            if src in gpt_sd:
                sd[k] = gpt_sd[src].clone()
    sd["wpe.weight"] = torch.zeros_like(sd["wpe.weight"])
    m.load_state_dict(sd, strict=False)
    return m


def test_block_stack_matches_hf_gpt2(gpt_sd_small, xtts_sd, dims, conditioning):
    gpt = O.GPTOracle(gpt_sd_small, xtts_sd)
    ids = make_synthetic_text_ids(dims, n_text=20)
    cond = gpt.build_cond(conditioning[0], ids)
    x = torch.cat([cond, gpt.mel_embed([1024], [0])], dim=0)
    h, _ = gpt.forward_rows(x, None)
    hf = _hf_gpt2(gpt_sd_small, 3)
    with torch.no_grad():
        ref = hf(inputs_embeds=x[None]).last_hidden_state[0]
    assert (h - ref).abs().max().item() < 2e-4


def test_erf_gelu_block_stack_matches_hf_gpt2(gpt_sd_small, xtts_sd, dims, conditioning):
    """config.json "activation_function": "gelu" (the XTTSGPTConfig class default, xttsv2_gpt_config.py:184) = the erf form:
    the oracle's variant against transformers' GPT2Model built with activation_function="gelu", and it must differ from the
    tanh form by more than the comparison tolerance (else the test could not tell them apart)."""
    gpt = O.GPTOracle(gpt_sd_small, xtts_sd, activation="gelu")
    ids = make_synthetic_text_ids(dims, n_text=20)
    cond = gpt.build_cond(conditioning[0], ids)
    x = torch.cat([cond, gpt.mel_embed([1024], [0])], dim=0)
    h, _ = gpt.forward_rows(x, None)
    with torch.no_grad():
        ref = _hf_gpt2(gpt_sd_small, 3, "gelu")(inputs_embeds=x[None]).last_hidden_state[0]
    assert (h - ref).abs().max().item() < 2e-4
    t = torch.linspace(-4, 4, 1001)
    assert (O.gelu_erf(t) - O.gelu_new(t)).abs().max().item() > 1e-4


def test_incremental_decode_equals_full_prefill(gpt_sd_small, xtts_sd, dims, conditioning):
    gpt = O.GPTOracle(gpt_sd_small, xtts_sd)
    ids = make_synthetic_text_ids(dims, n_text=12)
    cond = gpt.build_cond(conditioning[0], ids)
    out = gpt.generate(cond, O.SamplingCfg(temperature=0.0, max_tokens=10, ignore_stop=True))
    toks = out["tokens"]
    assert len(toks) == 10
    # decode-time ln_f rows == second-pass rows [n_cond:-5]
    lat2 = gpt.second_pass_latents(cond, toks)
    lat1 = gpt.latents_from_decode_rows(out["decode_rows"], len(toks))
    assert lat1.shape == lat2.shape == (1, 10, 1024)
    assert (lat1 - lat2).abs().max().item() < 5e-4


def test_repetition_penalty_example():
    # SURVEY A7'(v): logits [2,-1,4,0.5], ids {0,0,1,1,1,3}, p=5 -> [0.4,-5,4,0.1]
    z = O.apply_repetition_penalty(torch.tensor([2.0, -1.0, 4.0, 0.5]), [0, 0, 1, 1, 1, 3], 5.0)
    assert torch.allclose(z, torch.tensor([0.4, -5.0, 4.0, 0.1]))


def test_sampler_topk_topp_semantics():
    z = torch.log(torch.tensor([0.5, 0.2, 0.15, 0.1, 0.05]))
    ones = np.ones(5, dtype=np.float32)
    # uniform noise => argmax of probs
    assert O.sample_token(z, 1.0, 0, 1.0, ones) == 0
    # top_k = 2 keeps {0,1}; noise making index 2 attractive must not matter
    n = np.array([1, 1, 1e-6, 1, 1], dtype=np.float32)
    assert O.sample_token(z, 1.0, 2, 1.0, n) in (0, 1)
    # top_p = 0.6: ascending cumsum .05,.15,.30,.50,1.0 ; mask <= 0.4 -> keeps {0,1}
    n = np.array([1, 1, 1e-6, 1e-6, 1e-6], dtype=np.float32)
    assert O.sample_token(z, 1.0, 0, 0.6, n) in (0, 1)
    # greedy ignores noise
    assert O.sample_token(z, 0.0, 50, 0.85, None) == 0


def test_exp_noise_is_deterministic_and_exponential():
    e1 = O.exp_noise(7, 3, 1026)
    e2 = O.exp_noise(7, 3, 1026)
    assert np.array_equal(e1, e2)
    assert not np.array_equal(e1, O.exp_noise(7, 4, 1026))
    big = np.concatenate([O.exp_noise(s, 0, 1026) for s in range(200)])
    assert abs(big.mean() - 1.0) < 0.02 and abs(big.var() - 1.0) < 0.06 and big.min() >= 0


def test_stop_token_ends_generation(gpt_sd_small, xtts_sd, dims, conditioning):
    sd = {k: v.clone() for k, v in gpt_sd_small.items()}
    sd["mel_head.bias"][1025] = 100.0     # force the stop id
    gpt = O.GPTOracle(sd, xtts_sd)
    cond = gpt.build_cond(conditioning[0], make_synthetic_text_ids(dims, n_text=8))
    out = gpt.generate(cond, O.SamplingCfg(temperature=0.0, max_tokens=20))
    assert out["tokens"] == [1025]        # stop id kept in the output (XTTSv2.py:737)
