"""CPU: the oracle's sampler leg (oracle.sample_token, vLLM 0.6.4 `Sampler` semantics) against two
independent restatements, so that row a9 does not rest on a single unverified transcription:

 1. transformers' own logits warpers — TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper (vLLM's
    `_apply_top_k_top_p` is the fused form of these two: ascending sort, top-k by threshold with ties kept, top-p mask
    `cumsum <= 1 - p` that never drops the largest) — then softmax and the exponential race argmax(probs / e);
 2. a numpy implementation written from Appendix A4 with a lexicographic (value, id) sort.

The reference's call site is XttsGPT.sample (vllm_mm_gpt.py:691-712) -> vLLM Sampler; vLLM itself is not importable here.
Tie groups: vLLM leaves the order of equal logits to torch.sort; the oracle and the HIP sampler define it as ascending
(value, id) — the cases below pin that definition so a top-p cut inside a tie group is reproducible.
"""
import numpy as np
import pytest
import torch
from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

from oracle import xtts_oracle as O

V = 1026


def hf_filtered(logits: torch.Tensor, T: float, k: int, p: float) -> torch.Tensor:
    z = logits.float()[None]
    z = TemperatureLogitsWarper(T)(None, z)
    if 0 < k < z.shape[-1]:
        z = TopKLogitsWarper(top_k=k)(None, z)
    if p < 1.0:
        z = TopPLogitsWarper(top_p=p)(None, z)
    return z[0]


def np_filtered(logits: np.ndarray, T: float, k: int, p: float) -> np.ndarray:
    z = (logits.astype(np.float32) / np.float32(T)).astype(np.float32)
    n = z.shape[0]
    order = np.lexsort((np.arange(n), z))            # ascending by value, then id
    zs = z[order].copy()
    if 0 < k < n:
        zs[zs < zs[n - k]] = -np.inf
    if p < 1.0:
        e = np.exp(zs - zs[-1], dtype=np.float32)
        pr = e / e.sum(dtype=np.float32)
        cs = np.cumsum(pr, dtype=np.float32)
        m = cs <= np.float32(1.0 - p)
        m[-1] = False
        zs[m] = -np.inf
    out = np.empty_like(zs)
    out[order] = zs
    return out


def race(z_filtered: np.ndarray, noise: np.ndarray) -> int:
    z = z_filtered.astype(np.float64)
    pr = np.exp(z - z.max())
    pr /= pr.sum()
    return int(np.argmax(pr / noise.astype(np.float64)))


def oracle_survivors(logits: torch.Tensor, T: float, k: int, p: float) -> set:
    """ids that can still be drawn: sample under many noise vectors is costly, so read the survivor set off a probe —
    an id survives iff a noise vector that is tiny at that id makes it win."""
    surv = set()
    base = np.ones(V, dtype=np.float32)
    for v in range(V):
        n = base.copy()
        n[v] = 1e-30
        if O.sample_token(logits, T, k, p, n) == v:
            surv.add(v)
    return surv


@pytest.mark.parametrize("seed,T,k,p", [(0, 0.75, 50, 0.85), (1, 0.75, 50, 0.85), (2, 1.0, 50, 0.5), (3, 0.3, 5, 0.95),
                                        (4, 1.3, 0, 0.85), (5, 0.75, 50, 1.0), (6, 0.75, 1, 0.85), (7, 2.0, 1025, 0.999)])
def test_survivor_sets_and_tokens_agree(seed, T, k, p):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(V, generator=g) * 3.0
    hf = hf_filtered(logits, T, k, p).numpy()
    npf = np_filtered(logits.numpy(), T, k, p)
    s_hf, s_np = set(np.flatnonzero(np.isfinite(hf)).tolist()), set(np.flatnonzero(np.isfinite(npf)).tolist())
    assert s_hf == s_np
    if seed < 3:                                       # the probe costs V sampler calls: a few configurations are enough
        assert oracle_survivors(logits, T, k, p) == s_hf
    for step in range(16):
        noise = O.exp_noise(seed * 977 + 13, step, V)
        tok = O.sample_token(logits, T, k, p, noise)
        assert tok == race(hf, noise) == race(npf, noise), (seed, step)
        assert tok in s_hf


def test_tie_groups_are_cut_in_value_then_id_order():
    """A top-p cut that lands inside a group of equal logits drops the SMALLER ids first (ascending (value, id) order).
    6 equal runners-up below one leader: with p chosen so that three of the six fall under 1 - p, ids 10, 20, 30 go and
    40, 50, 60 stay."""
    z = torch.full((V,), -30.0)
    z[7] = 2.0
    for i in (10, 20, 30, 40, 50, 60):
        z[i] = 1.0
    e = np.exp(np.array([1.0] * 6 + [2.0]))
    pr = e / e.sum()
    lim = float(pr[:3].sum() + 0.5 * pr[3])            # cumulative mass of three tied ids, half-way to the fourth
    p = 1.0 - lim
    want = {40, 50, 60, 7}
    assert set(np.flatnonzero(np.isfinite(np_filtered(z.numpy(), 1.0, 0, p))).tolist()) == want
    base = np.ones(V, dtype=np.float32)
    got = set()
    for v in (7, 10, 20, 30, 40, 50, 60):
        n = base.copy()
        n[v] = 1e-30
        if O.sample_token(z, 1.0, 0, p, n) == v:
            got.add(v)
    assert got == want


def test_top_k_keeps_ties_at_the_threshold():
    z = torch.full((V,), -5.0)
    z[3], z[4], z[5], z[6] = 3.0, 2.0, 2.0, 2.0        # k = 2: the k-th value is 2.0 and all three 2.0s survive
    f = np_filtered(z.numpy(), 1.0, 2, 1.0)
    assert set(np.flatnonzero(np.isfinite(f)).tolist()) == {3, 4, 5, 6}
    assert set(np.flatnonzero(np.isfinite(hf_filtered(z, 1.0, 2, 1.0).numpy())).tolist()) == {3, 4, 5, 6}
    base = np.ones(V, dtype=np.float32)
    for v in (3, 4, 5, 6):
        n = base.copy()
        n[v] = 1e-30
        assert O.sample_token(z, 1.0, 2, 1.0, n) == v
    n = base.copy()
    n[100] = 1e-30
    assert O.sample_token(z, 1.0, 2, 1.0, n) in (3, 4, 5, 6)


def test_greedy_ignores_top_k_top_p_and_noise():
    g = torch.Generator().manual_seed(9)
    z = torch.randn(V, generator=g)
    assert O.sample_token(z, 0.0, 1, 0.01, None) == int(torch.argmax(z))
