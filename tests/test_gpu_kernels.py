"""GPU: each HIP kernel family against plain torch-CPU fp32 through the C-ABI debug entry points."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from auralis_amd import weights as Wt
from tests.gpu_util import make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e, *_ = make_engine(1, max_seqs=8)
    yield e
    e.close()


# the prefill-regime LDS-tiled kernel (128 x 128 / 64 x 64 tiles, one slab) that every prompt-row GEMM runs on
@pytest.mark.parametrize("M,N,K", [(103, 3072, 1024), (71, 1024, 1024), (4544, 1024, 4096), (300, 4096, 1024), (1, 1024, 1024),
                                   (129, 1024, 1024), (1000, 3072, 1024)])
def test_gemm_tile(eng, M, N, K):
    g = torch.Generator().manual_seed(M * 5 + N + K)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) * 0.05
    ref = (X.double() @ W.double()).float().numpy()
    got = eng.dbg_gemm(X.numpy(), W.numpy())
    err = np.abs(got - ref).max()
    assert err < 2e-4 * max(1.0, np.abs(ref).max()), err


def test_gemm_tile_split_arithmetic_matches_exact_f32(eng):
    """Default engines run the prompt-row GEMMs in the three-way bf16 split arithmetic (gemm_tile_split_kernel); with
    aur_config.gemm_f32_exact they run on exact-f32 MFMA.  Both against float64."""
    e32, *_ = make_engine(1, max_seqs=8, gemm_f32_exact=True)
    try:
        g = torch.Generator().manual_seed(77)
        for M, N, K in [(300, 3072, 1024), (200, 1024, 4096)]:
            X = torch.randn(M, K, generator=g)
            W = torch.randn(K, N, generator=g) * 0.05
            ref = (X.double() @ W.double()).numpy()
            a, b = eng.dbg_gemm(X.numpy(), W.numpy()), e32.dbg_gemm(X.numpy(), W.numpy())
            ea, eb = np.abs(a - ref).max(), np.abs(b - ref).max()
            print(f"M={M} N={N} K={K}: split max err {ea:.3e}, exact-f32 max err {eb:.3e}, max|ref| {np.abs(ref).max():.2f}")
            assert ea < 4.0 * eb + 1e-6, (ea, eb)
    finally:
        e32.close()


def test_gemm_tile_is_batch_invariant_bitwise(eng):
    """A prompt's rows must not depend on how many other rows were admitted in the same prefill step."""
    g = torch.Generator().manual_seed(9)
    X = torch.randn(500, 1024, generator=g)
    W = torch.randn(1024, 3072, generator=g) * 0.05
    big = eng.dbg_gemm(X.numpy(), W.numpy())
    for m in (1, 71, 128, 129, 300):
        assert np.array_equal(big[:m], eng.dbg_gemm(X[:m].numpy(), W.numpy())), m
    # narrow GEMMs pick their tile shape by the number of rows (128 x 128 / 128 x 64 / 64 x 64): same bits on every shape
    X2 = torch.randn(3300, 1024, generator=g)
    W2 = torch.randn(1024, 1024, generator=g) * 0.05
    big2 = eng.dbg_gemm(X2.numpy(), W2.numpy())
    for m in (64, 300, 2000):
        assert np.array_equal(big2[:m], eng.dbg_gemm(X2[:m].numpy(), W2.numpy())), m


def test_gemm_tile_presplit_weights_equal_split_on_the_fly(eng, monkeypatch):
    """Prompt-row GEMMs read their weights as the three bf16 planes of the exact split, packed at load time (launch_pack_wsplit);
    AUR_GEMM_PRESPLIT=0 selects the kernel that splits the fp32 weights again per tile.  Same split, same MFMA order: equal bit for
    bit, on every tile shape (128 x 128, 128 x 64, 64 x 64) and for every K."""
    monkeypatch.setenv("AUR_GEMM_PRESPLIT", "0")
    e_reg, *_ = make_engine(1, max_seqs=8)
    try:
        g = torch.Generator().manual_seed(21)
        for M, N, K in [(71, 3072, 1024), (300, 1024, 1024), (2000, 1024, 1024), (4544, 1024, 1024), (513, 4096, 1024), (150, 1024, 4096), (1, 1024, 1024)]:
            X = torch.randn(M, K, generator=g)
            W = torch.randn(K, N, generator=g) * 0.05
            a, b = eng.dbg_gemm(X.numpy(), W.numpy()), e_reg.dbg_gemm(X.numpy(), W.numpy())
            assert np.array_equal(a, b), (M, N, K, np.abs(a - b).max())
    finally:
        e_reg.close()


def _gelu_new(x):
    return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


# decode-regime GEMM (gemm_rows_kernel): M picks the rows-per-workgroup variant (16 / 32 / 64 rows), N the tile map
# (N / 16 multiple of 8 or not), K = 4096 the two-buffer chunk loop; epi 0 bias, 1 bias + gelu, 2 residual
@pytest.mark.parametrize("M,N,K,epi,ln", [(64, 3072, 1024, 0, True), (64, 4096, 1024, 1, True), (64, 1024, 1024, 2, False),
                                          (64, 1024, 4096, 2, False), (64, 1088, 1024, 0, False), (1, 4096, 1024, 1, True),
                                          (17, 1024, 4096, 2, False), (37, 3072, 1024, 0, True), (5, 1088, 1024, 0, False),
                                          (130, 4096, 1024, 1, True), (200, 1024, 4096, 0, False), (48, 1024, 1024, 2, False)])
def test_gemm_rows(eng, M, N, K, epi, ln):
    g = torch.Generator().manual_seed(M * 11 + N + K + epi)
    X = torch.randn(M, K, generator=g) * (3.0 if ln else 1.0) + (0.5 if ln else 0.0)
    W = torch.randn(K, N, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    gamma = torch.randn(K, generator=g) if ln else None
    beta = torch.randn(K, generator=g) if ln else None
    res = torch.randn(M, N, generator=g) if epi == 2 else None
    A = F.layer_norm(X, (K,), gamma, beta, 1e-5) if ln else X
    ref = (A.double() @ W.double() + bias.double()).numpy()
    if epi == 1:
        ref = _gelu_new(ref)
    if epi == 2:
        ref = ref + res.double().numpy()
    got = eng.dbg_gemm_rows(X.numpy(), W.numpy(), bias.numpy(), None if gamma is None else gamma.numpy(),
                            None if beta is None else beta.numpy(), None if res is None else res.numpy(), epi)
    err = np.abs(got - ref).max()
    assert err < 2e-4 * max(1.0, np.abs(ref).max()), err


def test_gemm_rows_is_batch_invariant_bitwise(eng):
    """A row's result does not depend on how many other rows are live (16-, 32- and 64-row workgroup variants, row groups)."""
    g = torch.Generator().manual_seed(5)
    for K, N, ln in ((1024, 4096, True), (4096, 1024, False), (1024, 1024, False)):
        X = torch.randn(150, K, generator=g)
        W = torch.randn(K, N, generator=g) * 0.05
        gamma = torch.randn(K, generator=g).numpy() if ln else None
        beta = torch.randn(K, generator=g).numpy() if ln else None
        full = eng.dbg_gemm_rows(X.numpy(), W.numpy(), None, gamma, beta, None, 0)
        for m in (1, 16, 17, 33, 64, 100):
            part = eng.dbg_gemm_rows(X[:m].numpy(), W.numpy(), None, gamma, beta, None, 0)
            assert np.array_equal(full[:m], part), (K, N, m)


@pytest.mark.parametrize("M", [1, 16, 32])
def test_gemm_rows_ksplit_stress(eng, M):
    """The K-split projection (four workgroups per output tile, partial tiles handed over through write-through stores, a device
    ticket and sc0 sc1 loads; the decode path of every step at <= 32 live sequences): 3 000 back-to-back launches, each compared
    bitwise on the device with the unsplit kernel.  A hand-off that is not visible in time would show as rare differing words."""
    assert eng.dbg_gemm_rows_ksplit_stress(M, 3000) == 0


def test_cross_lane_helpers_equal_the_shuffles_they_replace(eng):
    """lane_xor<1..32>, wave_sum and wave_max (csrc/common.h: DPP permutations and gfx950's v_permlane16/32_swap) against __shfl_xor
    and the six-step shuffle butterflies, bitwise, on 1 024 waves of pseudo-random words: LayerNorm statistics, the sampler's argmax,
    its 64-key sort and its softmax denominator all run on them, and every bit-exact test downstream assumes they permute exactly."""
    assert eng.dbg_lane_xor_selftest(256) == 0


def test_layernorm(eng):
    g = torch.Generator().manual_seed(0)
    h = torch.randn(37, 1024, generator=g) * 3 + 0.5
    gamma = torch.randn(1024, generator=g)
    beta = torch.randn(1024, generator=g)
    ref = F.layer_norm(h, (1024,), gamma, beta, 1e-5).numpy()
    got = eng.dbg_layernorm(h.numpy(), gamma.numpy(), beta.numpy())
    assert np.abs(got - ref).max() < 2e-5


CONV_CASES = [(64, 32, 3, 1), (64, 32, 3, 3), (64, 32, 3, 5), (128, 16, 7, 1), (64, 16, 7, 3), (64, 16, 7, 5),
              (64, 16, 11, 1), (64, 16, 11, 3), (64, 16, 11, 5), (32, 32, 3, 1), (32, 32, 7, 5), (32, 32, 11, 5),
              (32, 32, 11, 1)]


@pytest.mark.parametrize("cout,cin,k,d", CONV_CASES)
def test_conv1d_mfma(eng, cout, cin, k, d):
    g = torch.Generator().manual_seed(cout + cin * k + d)
    B, L = 2, 700
    lens = [700, 389]                                    # ragged batch: per-utterance zero padding
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    pad = (k - 1) // 2 * d
    got = eng.dbg_conv1d(x.numpy(), Wt.pack_conv(w).numpy(), b.numpy(), res.numpy(), lens, k, d, pad, 0.1, cout)
    for bi, ln in enumerate(lens):
        ref = F.conv1d(F.leaky_relu(x[bi:bi + 1, :, :ln], 0.1), w, b, dilation=d, padding=pad)[0] + res[bi, :, :ln]
        err = (torch.from_numpy(got[bi, :, :ln]) - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (bi, err)
        assert np.all(got[bi, :, ln:] == 0.0)            # nothing written past the utterance


@pytest.mark.parametrize("cout,cin,k,d", [(64, 32, 3, 1), (64, 32, 7, 3), (128, 64, 11, 5), (32, 32, 3, 5), (32, 32, 11, 1),
                                          (64, 16, 7, 1)])
def test_conv1d_mfma_f16(eng, cout, cin, k, d):
    """fp16-input / fp32-accumulate variant: exact against a reference that rounds inputs and weights to fp16."""
    g = torch.Generator().manual_seed(cout + cin * k + d)
    B, L = 2, 900
    lens = [900, 411]
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    pad = (k - 1) // 2 * d
    wp16 = Wt.pack_conv_f16(w)
    got = eng.dbg_conv1d_f16(x.numpy(), wp16.numpy(), b.numpy(), res.numpy(), lens, k, d, pad, 0.1, cout, cout)
    w16 = w.to(torch.float16).float()
    for bi, ln in enumerate(lens):
        xa = F.leaky_relu(x[bi:bi + 1, :, :ln], 0.1).to(torch.float16).float()
        ref = F.conv1d(xa.double(), w16.double(), b.double(), dilation=d, padding=pad)[0].float() + res[bi, :, :ln]
        err = (torch.from_numpy(got[bi, :, :ln]) - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (bi, err)
        full = F.conv1d(F.leaky_relu(x[bi:bi + 1, :, :ln], 0.1), w, b, dilation=d, padding=pad)[0] + res[bi, :, :ln]
        assert (torch.from_numpy(got[bi, :, :ln]) - full).abs().max().item() < 2e-2      # fp16 rounding only
        assert np.all(got[bi, :, ln:] == 0.0)


def test_conv_transpose_polyphase_f16(eng):
    g = torch.Generator().manual_seed(21)
    cin, cout, s, k = 64, 32, 8, 16
    B, L = 2, 300
    lens = [300, 77]
    w = torch.randn(cin, cout, k, generator=g) / (cin * k / s) ** 0.5
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    wv = Wt.polyphase_convT(w, s)
    got = eng.dbg_conv1d_f16(x.numpy(), Wt.pack_conv_f16(wv).numpy(), b.numpy(), None, lens, 2, 1, 1, 0.1, cout, cout * s,
                             ups_s=s, ups_p=(k - s) // 2)
    for bi, ln in enumerate(lens):
        xa = F.leaky_relu(x[bi:bi + 1, :, :ln], 0.1).to(torch.float16).float()
        ref = F.conv_transpose1d(xa, w.to(torch.float16).float(), b, stride=s, padding=(k - s) // 2)[0]
        err = (torch.from_numpy(got[bi, :, :ln * s]) - ref).abs().max().item()
        assert err < 5e-5 * max(1.0, ref.abs().max().item()), (bi, err)


@pytest.mark.parametrize("cin,cout,s,k", [(64, 32, 8, 16), (32, 32, 2, 4), (512, 256, 8, 16)])
def test_conv_transpose_polyphase(eng, cin, cout, s, k):
    g = torch.Generator().manual_seed(cin + s)
    B, L = 2, 300
    lens = [300, 77]
    w = torch.randn(cin, cout, k, generator=g) / (cin * k / s) ** 0.5
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    wp = Wt.pack_conv(Wt.polyphase_convT(w, s))
    got = eng.dbg_conv1d(x.numpy(), wp.numpy(), b.numpy(), None, lens, 2, 1, 1, 0.1, cout, ups_s=s, ups_p=(k - s) // 2)
    for bi, ln in enumerate(lens):
        ref = F.conv_transpose1d(F.leaky_relu(x[bi:bi + 1, :, :ln], 0.1), w, b, stride=s, padding=(k - s) // 2)[0]
        err = (torch.from_numpy(got[bi, :, :ln * s]) - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (bi, err)


def test_sampler_greedy_and_penalty(eng):
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(6, 1026, generator=g)
    toks = eng.dbg_sample(logits.numpy(), 0.0, 0.85, 50)
    assert list(toks) == logits.argmax(dim=1).tolist()
    # repetition penalty applies in greedy mode too (hijack.py:49-88)
    seen = np.zeros((6, 1026), dtype=np.uint8)
    for r in range(6):
        seen[r, int(logits[r].argmax())] = 1
    toks2 = eng.dbg_sample(logits.numpy(), 0.0, 0.85, 50, repetition_penalty=5.0, seen=seen)
    from oracle import xtts_oracle as O
    for r in range(6):
        z = O.apply_repetition_penalty(logits[r], np.nonzero(seen[r])[0].tolist(), 5.0)
        assert toks2[r] == int(z.argmax())


@pytest.mark.parametrize("T,top_k,top_p", [(0.75, 50, 0.85), (1.0, -1, 0.5), (0.3, 5, 1.0), (1.3, 1026, 0.99)])
def test_sampler_matches_oracle_with_shared_noise(eng, T, top_k, top_p):
    from oracle import xtts_oracle as O
    g = torch.Generator().manual_seed(int(T * 100) + top_k)
    B = 8
    logits = torch.randn(B, 1026, generator=g) * 2.0
    seed, step = 1000, 3
    toks = eng.dbg_sample(logits.numpy(), T, top_p, top_k, seed=seed, step=step)
    agree = 0
    for r in range(B):
        noise = O.exp_noise(seed + r, step, 1026)
        agree += int(toks[r] == O.sample_token(logits[r], T, top_k, top_p, noise))
    assert agree == B, (agree, toks)


def test_sampler_tie_groups_match_oracle(eng):
    """Heavily tied logits: the top-k threshold keeps every tie at the k-th value and a top-p cut inside a tie group drops
    the smaller ids first (ascending (value, id) order, the oracle's stated tie rule; CPU cross-check of that rule against
    transformers' warpers and a numpy restatement: tests/test_oracle_sampler.py)."""
    from oracle import xtts_oracle as O
    g = torch.Generator().manual_seed(123)
    B = 8
    tied = torch.round(torch.randn(B, 1026, generator=g) * 2.0) * 0.5          # ~10 distinct values
    coarse = torch.round(torch.randn(B, 1026, generator=g) * 40.0) / 8.0      # a few ties around the k-th value
    for name, lg in (("tied", tied), ("coarse", coarse)):
        for T, k, p in ((0.75, 50, 0.85), (1.0, 64, 0.5), (1.5, 7, 0.3), (0.9, -1, 0.6)):
            for step in (0, 5):
                toks = eng.dbg_sample(lg.numpy(), T, p, k, seed=900, step=step)
                want = [O.sample_token(lg[r], T, k, p, O.exp_noise(900 + r, step, 1026)) for r in range(B)]
                assert list(toks) == want, (name, T, k, p, step)


def test_sampler_distribution_chi2(eng):
    """Exponential-race sampling reproduces softmax(top-k) frequencies (chi-square over 4000 draws)."""
    z = np.full(1026, -30.0, dtype=np.float32)
    p = np.array([0.4, 0.3, 0.2, 0.1])
    z[:4] = np.log(p)
    counts = np.zeros(4)
    B = 8
    for it in range(500):
        toks = eng.dbg_sample(np.tile(z, (B, 1)), 1.0, 1.0, 4, seed=it * 64, step=it)
        for t in toks:
            assert t < 4
            counts[t] += 1
    n = counts.sum()
    chi2 = float(((counts - n * p) ** 2 / (n * p)).sum())
    assert chi2 < 16.3, (chi2, counts)     # 3 dof, p = 0.001


def test_sampler_topk_fast_path_equals_full_sort(eng, monkeypatch):
    """0 < top_k <= 64 takes the bisection + one-wave sort path; AUR_SAMPLER_FULL_SORT=1 forces the 2048-key bitonic sort.
    Same tokens on smooth logits, on heavily tied logits (ties at the k-th value are kept, > 128 survivors fall back to
    the full sort) and with -inf / duplicated maxima.  (The oracle comparisons above cover k = 50 and k = 5 on smooth logits;
    inside a tie group the top-p cut depends on the sort's tie order, which torch.sort does not pin down.)"""
    monkeypatch.setenv("AUR_SAMPLER_FULL_SORT", "1")
    ref_eng, *_ = make_engine(1, max_seqs=8)
    try:
        g = torch.Generator().manual_seed(77)
        B = 8
        smooth = torch.randn(B, 1026, generator=g) * 3.0
        tied = torch.round(torch.randn(B, 1026, generator=g) * 2.0) * 0.5          # ~10 distinct values: massive ties
        coarse = torch.round(torch.randn(B, 1026, generator=g) * 40.0) / 8.0      # a few ties around the k-th value
        holes = smooth.clone()
        holes[:, ::3] = -float("inf")
        holes[:, 5] = holes[:, 7] = 9.0                                           # duplicated maximum
        for name, lg in (("smooth", smooth), ("tied", tied), ("coarse", coarse), ("holes", holes)):
            for T, k, p in ((0.75, 50, 0.85), (1.0, 64, 1.0), (0.4, 1, 0.9), (1.5, 7, 0.3), (0.9, 33, 0.6)):
                for step in (0, 11):
                    a = eng.dbg_sample(lg.numpy(), T, p, k, seed=4242, step=step)
                    b = ref_eng.dbg_sample(lg.numpy(), T, p, k, seed=4242, step=step)
                    assert list(a) == list(b), (name, T, k, p, step, a, b)
    finally:
        ref_eng.close()


def test_sampler_top_p_cut_is_the_serial_cumulative_sum(eng):
    """the top-p cumulative sum of the k <= 64 path was a parallel suffix scan in double -- the reference's additions in
    another order, equal to the serial form except in the 53rd bit.  It is now serial in the reference's order on every path.  The test
    drives the same distributions through BOTH code paths of the kernel: with top_k = 50 the candidates are the 50 survivors (one wave,
    registers), with top_k disabled they are all 1026 ids sorted by the full bitonic network and cut by thread 0's serial loop -- the
    other 976 ids carry probability exactly 0 (their logits sit 1e4 below), which adds nothing to either sum.  Same noise, so the same
    token on every row, for top_p values that put the cut at a boundary (cumulative sums of dyadic probabilities) and in a wide
    dynamic range."""
    g = torch.Generator().manual_seed(91)
    B = 8
    base = torch.full((B, 1026), -1.0e4) - torch.arange(1026, dtype=torch.float32)[None, :] * 1e-2   # distinct, far below: probability 0
    wide = base.clone()
    dyadic = base.clone()
    for r in range(B):
        ids = torch.randperm(1026, generator=g)[:50]
        wide[r, ids] = torch.randn(50, generator=g) * 6.0                        # probabilities over ~15 orders of magnitude
        # probabilities 2^-1, 2^-2, ..., 2^-9 and forty-one of 2^-9 / 41: cumulative sums land ON the dyadic boundaries 1 - p
        lp = torch.cat([-(torch.arange(1, 10, dtype=torch.float32)) * 0.6931471805599453,
                        torch.full((41,), -9 * 0.6931471805599453 - float(np.log(41.0)))])
        dyadic[r, ids] = lp
    for name, lg in (("wide", wide), ("dyadic", dyadic)):
        for T, p in ((1.0, 0.5), (1.0, 0.75), (1.0, 0.875), (1.0, 0.9375), (0.75, 0.85), (1.3, 0.6)):
            for step in (0, 5, 9):
                a = eng.dbg_sample(lg.numpy(), T, p, 50, seed=808, step=step)     # survivors in one wave
                b = eng.dbg_sample(lg.numpy(), T, p, 0, seed=808, step=step)      # all ids, serial cut
                assert list(a) == list(b), (name, T, p, step, list(a), list(b))


def _attention_f64(q, k, v, ctx):
    """float64 restatement of GPT2Attention on a cached context (vllm_mm_gpt.py:757-761 -> vLLM paged attention): per head,
    softmax(q . K^T / sqrt(64)) V over the row's first ctx[m] tokens."""
    M = q.shape[0]
    out = np.zeros((M, 1024), np.float64)
    for m in range(M):
        n = int(ctx[m])
        qq = q[m].astype(np.float64).reshape(16, 64)
        kk = k[m, :n].astype(np.float64).reshape(n, 16, 64)
        vv = v[m, :n].astype(np.float64).reshape(n, 16, 64)
        s = np.einsum("hd,nhd->hn", qq, kk) / 8.0
        s -= s.max(axis=1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=1, keepdims=True)
        out[m] = np.einsum("hn,nhd->hd", p, vv).reshape(1024)
    return out


def test_paged_attention_kernel_against_float64_ragged_shared_prefix_and_batch_invariance(eng):
    from auralis_amd._lib import AurError
    engine = eng
    """The decode attention kernel alone (31 % of the GPU time of the bench; until round 6 it was held only through the end-to-end
    goldens).  Fifteen rows whose contexts sit on every edge of its tiling -- 1 token, either side of a 16-token block, of the 64-token
    iteration of a workgroup, the bench's 244, the 605-token budget + prompt, the table's full 1 056 -- with keys scaled so that the
    softmax is peaked on some rows and flat on others:
      * against a float64 restatement: <= 2e-6 of the output scale;
      * a row alone == the row in the batch, bit for bit (batch invariance);
      * the first 32 tokens in blocks shared by every row (the speaker prefix) == the same tokens in the rows' own blocks, bit for bit;
      * the fp16 pool against float64 on the rounded keys / values: same tolerance."""
    rng = np.random.default_rng(5)
    ctx = np.array([1, 2, 15, 16, 17, 32, 33, 63, 64, 65, 128, 244, 640, 708, 1056], np.int32)
    M, cmax = len(ctx), int(ctx.max())
    q = rng.standard_normal((M, 1024)).astype(np.float32)
    k = rng.standard_normal((M, cmax, 1024)).astype(np.float32)
    v = rng.standard_normal((M, cmax, 1024)).astype(np.float32)
    k[::2] *= 0.25           # every other row: nearly flat softmax; the others peaked (|q.k| / 8 ~ N(0, 1) * 8 / 8)
    got = engine.dbg_paged_attention(q, k, v, ctx)
    ref = _attention_f64(q, k, v, ctx)
    err = np.abs(got - ref).max()
    assert err <= 2e-6 * max(1.0, np.abs(ref).max()), err
    for m in (0, 3, 7, 11, 14):
        solo = engine.dbg_paged_attention(q[m:m + 1], k[m:m + 1, :ctx[m]], v[m:m + 1, :ctx[m]], ctx[m:m + 1])
        assert np.array_equal(solo[0], got[m]), m
    # shared prefix: rows with >= 32 tokens, the first 32 tokens of every row replaced by row 0's (what sharing means), once in
    # private blocks and once in two blocks every table points to
    rows = np.nonzero(ctx >= 32)[0]
    qs, ks, vs, cs = q[rows], k[rows].copy(), v[rows].copy(), ctx[rows]
    ks[:, :32] = ks[0, :32]
    vs[:, :32] = vs[0, :32]
    private = engine.dbg_paged_attention(qs, ks, vs, cs, shared=0)
    shared = engine.dbg_paged_attention(qs, ks, vs, cs, shared=32)
    assert np.array_equal(private, shared)
    assert np.abs(shared - _attention_f64(qs, ks, vs, cs)).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    # fp16 pool
    got_h = engine.dbg_paged_attention(q, k, v, ctx, kv_half=True)
    ref_h = _attention_f64(q, k.astype(np.float16).astype(np.float32), v.astype(np.float16).astype(np.float32), ctx)
    assert np.abs(got_h - ref_h).max() <= 2e-6 * max(1.0, np.abs(ref_h).max())
    with pytest.raises(AurError):
        engine.dbg_paged_attention(q[:1], k[:1], v[:1], np.array([0], np.int32))


def test_prompt_attention_kernel_against_float64_query_blocks_shared_prefix_and_batch_invariance(eng):
    """The prefill attention kernel alone (exact-f32 MFMA tiles over query blocks of up to 32 consecutive rows of one sequence): a full
    prompt from position 0 (103 rows = blocks of 32, 32, 32, 7), a prompt whose first 32 positions are somebody else's (rows start at
    position 32, as with the shared speaker prefix), one lone row deep in a context, a 41-row prompt; causal mask per element.
      * against a float64 restatement: <= 2e-6 of the output scale;
      * a sequence's rows alone == the same rows in the batch, bit for bit;
      * the first 32 tokens in shared blocks == the same tokens in private blocks, bit for bit;
      * the fp16 pool against float64 on the rounded keys / values."""
    rng = np.random.default_rng(9)
    n_seq, cmax = 4, 200
    k = rng.standard_normal((n_seq, cmax, 1024)).astype(np.float32)
    v = rng.standard_normal((n_seq, cmax, 1024)).astype(np.float32)
    k[1] *= 0.25
    spans = [(0, 0, 103), (1, 32, 111), (2, 150, 151), (3, 0, 41)]          # (sequence, first position, one past the last)
    row_seq = np.concatenate([np.full(b - a, s_, np.int32) for s_, a, b in spans])
    row_pos = np.concatenate([np.arange(a, b, dtype=np.int32) for _, a, b in spans])
    M = len(row_seq)
    q = rng.standard_normal((M, 1024)).astype(np.float32)

    def ref64(q_, k_, v_, rs, rp):
        kk = np.stack([k_[s_] for s_ in rs])
        vv = np.stack([v_[s_] for s_ in rs])
        return _attention_f64(q_, kk, vv, rp + 1)
    got = eng.dbg_prompt_attention(q, k, v, row_seq, row_pos)
    ref = ref64(q, k, v, row_seq, row_pos)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-6 * scale, np.abs(got - ref).max()
    sel = row_seq == 1                                                        # sequence 1 alone, as sequence 0 of its own launch
    solo = eng.dbg_prompt_attention(q[sel], k[1:2], v[1:2], np.zeros(sel.sum(), np.int32), row_pos[sel])
    assert np.array_equal(solo, got[sel])
    ks, vs = k.copy(), v.copy()
    ks[:, :32] = ks[0, :32]
    vs[:, :32] = vs[0, :32]
    private = eng.dbg_prompt_attention(q, ks, vs, row_seq, row_pos, shared=0)
    shared = eng.dbg_prompt_attention(q, ks, vs, row_seq, row_pos, shared=32)
    assert np.array_equal(private, shared)
    assert np.abs(shared - ref64(q, ks, vs, row_seq, row_pos)).max() <= 2e-6 * scale
    got_h = eng.dbg_prompt_attention(q, k, v, row_seq, row_pos, kv_half=True)
    ref_h = ref64(q, k.astype(np.float16).astype(np.float32), v.astype(np.float16).astype(np.float32), row_seq, row_pos)
    assert np.abs(got_h - ref_h).max() <= 2e-6 * max(1.0, np.abs(ref_h).max())
