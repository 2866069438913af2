"""CPU fp32 oracle for the XTTSv2 generate_speech() hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU fp32, the arithmetic of the reference path
(astramind-ai/Auralis v0.2.8.post2).  It is imported ONLY by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, always as the checker; the product path (auralis_amd/) never
imports it and fails loudly when the HIP extension is missing.

Parity pin status
-----------------
* Vocoder (hifigan_forward / hifi_decoder_forward): PINNED against the reference's own
  `HifiDecoder` class imported unmodified in the build container (oracle/make_golden.py →
  tests/golden/vocoder_*.npz, and live in tests/test_oracle_vocoder.py when /root/reference exists).
* GPT glue (GPTOracle.build_cond / mel_embed / forward_rows ordering / ln_f, apply_repetition_penalty): PINNED against
  the reference's own `GPT2Model.forward`, `LearnedPositionEmbeddings` (vllm_mm_gpt.py) and `LogitsRepetitionPenalizer`
  (vllm/hijack.py), executed unmodified in the build container through import stubs (oracle/ref_gpt_import.py,
  oracle/make_golden_gpt.py → tests/golden/gpt_glue_L2.npz, tests/test_reference_gpt_glue.py).
* Sampler cross-check (not a pin): tests/test_oracle_sampler.py drives the same logits through two independent
  restatements — transformers' TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper (the code vLLM's
  `_apply_top_k_top_p` was derived from) followed by the exponential race, and a numpy re-implementation written from
  SURVEY Appendix A4 — and requires identical survivor sets and tokens, including explicit tie-group cases.
* GPT block arithmetic and the sampler: the reference delegates these to the un-vendored third-party dependency
  vllm==0.6.4.post1 (setup.py:63): `GPT2Block` and `Sampler` are not under /root/reference and the reference's tests hold
  no golden vectors for them.  In the fixture above vllm's block is replaced by the textbook GPT-2 block
  built from transformers' Conv1D / gelu_new modules; the restatement is additionally cross-checked against
  transformers.GPT2Model (tests/test_oracle_gpt.py) and the sampler follows vLLM 0.6.4's published order.
  For these two pieces the oracle is "parity unpinned" in the strict sense: no reference-produced vector can exist.

Reference citations are relative to /root/reference/src/auralis/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# small ops
# ----------------------------------------------------------------------------------------------
def gelu_new(x: Tensor) -> Tensor:
    """tanh-form GELU ("gelu_new", models/xttsv2/utils/checkpoint_converter.py:197)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def gelu_erf(x: Tensor) -> Tensor:
    """erf-form GELU (config.json "activation_function": "gelu", the XTTSGPTConfig class default,
    src/auralis/models/xttsv2/config/xttsv2_gpt_config.py:184; transformers ACT2FN["gelu"])."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ----------------------------------------------------------------------------------------------
# counter-based noise shared by oracle and HIP sampler (new surface: the reference has no seed)
# ----------------------------------------------------------------------------------------------
def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def exp_noise(seed: int, step: int, vocab: int) -> np.ndarray:
    """Exp(1) noise e[v] for sampling step `step` of a sequence seeded with `seed`.

    u = (hash >> 8 + 1) / 2^24 in (0, 1];  e = max(-log(u), 2^-25)   (float32).  Mirrors exp_noise in csrc/gpt_kernels.hip.
    The floor only acts on u == 1 (one draw in 2^24), where -log(u) is -0.0: vLLM's `probs.div_(q).argmax()` would turn a
    masked-out id (p = 0) into 0 / 0 = NaN there and select it; a strictly positive e keeps q = p / e finite and zero for every
    masked id (found by the all-64-seeds fixture: seed 50, step 274)."""
    with np.errstate(over="ignore"):
        v = np.arange(vocab, dtype=np.uint32)
        a = _lowbias32(v * np.uint32(0x9E3779B1) + np.uint32(seed & 0xFFFFFFFF))
        b = _lowbias32(a ^ (np.uint32(step & 0xFFFFFFFF) * np.uint32(0x85EBCA77) + np.uint32(0x165667B1)))
    u = ((b >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    return np.maximum(-np.log(u.astype(np.float32)), np.float32(2.0 ** -25)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# sampling (models/xttsv2/components/vllm/hijack.py:49-88 + vLLM 0.6.4 Sampler order, Appendix A4)
# ----------------------------------------------------------------------------------------------
def apply_repetition_penalty(logits: Tensor, seen_ids: Sequence[int], penalty: float) -> Tensor:
    """hijack.py:67-88: gather → where(>0, /p, *p) → scatter; once per unique id."""
    if penalty == 1.0 or len(seen_ids) == 0:
        return logits
    idx = torch.tensor(list(seen_ids), dtype=torch.long)
    rep = logits[idx]
    rep = torch.where(rep > 0, rep / penalty, rep * penalty)
    logits = logits.clone()
    logits[idx] = rep
    return logits


def sample_token(logits: Tensor, temperature: float, top_k: int, top_p: float,
                 noise: Optional[np.ndarray]) -> int:
    """vLLM 0.6.4.post1 Sampler semantics on one row of (already penalised) fp32 logits.

    greedy when temperature < 1e-5 (vLLM _SAMPLING_EPS) → argmax, top-k/top-p ignored;
    otherwise /T → top-k (ties at the k-th value kept) → top-p on the ascending-sorted softmax
    (mask cumsum <= 1-p, never the largest) → softmax → argmax(probs / Exp(1)).
    """
    z = logits.to(torch.float32)
    if temperature < 1e-5:
        return int(torch.argmax(z).item())
    z = z / temperature
    V = z.shape[0]
    # tie order made explicit (vLLM's torch.sort leaves it to the backend): ascending by (value, id), i.e. a stable sort;
    # a top-p cut that falls inside a group of equal logits therefore drops the smaller ids first.  The HIP sampler sorts
    # the packed (value, id) keys and produces the same order (tests/test_oracle_sampler.py, tests/test_gpu_kernels.py)
    z_sort, z_idx = z.sort(dim=-1, descending=False, stable=True)
    if top_k is not None and 0 < top_k < V:
        thr = z_sort[V - top_k]
        z_sort = z_sort.masked_fill(z_sort < thr, float("-inf"))
    if top_p is not None and top_p < 1.0:
        probs_sort = z_sort.softmax(dim=-1)
        probs_sum = probs_sort.cumsum(dim=-1)
        mask = probs_sum <= (1.0 - top_p)
        mask[-1] = False
        z_sort = z_sort.masked_fill(mask, float("-inf"))
    z = torch.empty_like(z_sort).scatter_(0, z_idx, z_sort)
    probs = torch.softmax(z, dim=-1)
    assert noise is not None, "sampling needs Exp(1) noise"
    q = probs / torch.from_numpy(noise.astype(np.float32))
    return int(torch.argmax(q).item())


# ----------------------------------------------------------------------------------------------
# GPT
# ----------------------------------------------------------------------------------------------
@dataclass
class SamplingCfg:
    temperature: float = 0.0
    top_k: int = 50
    top_p: float = 0.85
    repetition_penalty: float = 5.0
    max_tokens: int = 605
    ignore_stop: bool = False       # fixed-length mode used for timing
    seed: int = 0


class GPTOracle:
    """XttsGPT restated (models/xttsv2/components/vllm_mm_gpt.py:341-849; Appendix A1-A6).

    `gpt_sd` uses the on-disk key names of gpt2_model.safetensors (Appendix B); `xtts_sd` supplies
    text_embedding / text_pos_embedding (xtts-v2.safetensors).
    """

    def __init__(self, gpt_sd: Dict[str, Tensor], xtts_sd: Optional[Dict[str, Tensor]] = None,
                 n_head: int = 16, eps: float = 1e-5, start_token: int = 1024, stop_token: int = 1025,
                 activation: str = "gelu_new"):
        assert activation in ("gelu_new", "gelu"), activation
        self.act = gelu_new if activation == "gelu_new" else gelu_erf
        self.w = {k: v.to(torch.float32) for k, v in gpt_sd.items()}
        self.x = None if xtts_sd is None else xtts_sd
        self.n_layer = 1 + max(int(k.split(".")[2]) for k in self.w if k.startswith("gpt.h."))
        self.hidden = self.w["gpt.wte.weight"].shape[1]
        self.n_head = n_head
        self.head_dim = self.hidden // n_head
        self.eps = eps
        self.start_token = start_token
        self.stop_token = stop_token

    # -- A1 prompt build (XTTSv2.py:345,528; vllm_mm_gpt.py:804-813,777-783) ------------------
    def text_embed(self, text_ids: Sequence[int]) -> Tensor:
        ids = torch.tensor(list(text_ids), dtype=torch.long)
        te = self.x["text_embedding.weight"].float()[ids]
        tp = self.x["text_pos_embedding.emb.weight"].float()[: len(ids)]
        return te + tp

    def build_cond(self, gpt_cond_latent: Tensor, text_ids: Sequence[int]) -> Tensor:
        return torch.cat([gpt_cond_latent.reshape(-1, self.hidden).float(), self.text_embed(text_ids)], dim=0)

    def mel_embed(self, ids: Sequence[int], positions: Sequence[int]) -> Tensor:
        ids_t = torch.tensor(list(ids), dtype=torch.long)
        pos_t = torch.tensor(list(positions), dtype=torch.long)
        return self.w["gpt.wte.weight"][ids_t] + self.w["gpt.wpe.emb.weight"][pos_t]

    # -- A2 block ---------------------------------------------------------------------------
    def _block(self, i: int, x: Tensor, kv: Optional[Tuple[Tensor, Tensor]]):
        """Pre-LN GPT-2 block over rows x [T,H] appended after cached kv; returns (x, (K,V))."""
        w = self.w
        p = f"gpt.h.{i}."
        T = x.shape[0]
        a = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], self.eps)
        qkv = a @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
        q, k, v = qkv.split(self.hidden, dim=-1)
        if kv is not None:
            k = torch.cat([kv[0], k], dim=0)
            v = torch.cat([kv[1], v], dim=0)
        S = k.shape[0]
        qh = q.view(T, self.n_head, self.head_dim).transpose(0, 1)          # [h,T,d]
        kh = k.view(S, self.n_head, self.head_dim).transpose(0, 1)
        vh = v.view(S, self.n_head, self.head_dim).transpose(0, 1)
        scores = (qh @ kh.transpose(1, 2)) * (1.0 / math.sqrt(self.head_dim))
        # causal: query t (absolute position S-T+t) sees keys <= S-T+t
        qpos = torch.arange(S - T, S).unsqueeze(1)
        kpos = torch.arange(S).unsqueeze(0)
        scores = scores.masked_fill((kpos > qpos).unsqueeze(0), float("-inf"))
        att = torch.softmax(scores, dim=-1) @ vh                             # [h,T,d]
        att = att.transpose(0, 1).reshape(T, self.hidden)
        x = x + att @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
        m = layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], self.eps)
        f = self.act(m @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
        x = x + f @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
        return x, (k, v)

    def forward_rows(self, x: Tensor, cache: Optional[List[Tuple[Tensor, Tensor]]]):
        """Run all blocks + ln_f (vllm_mm_gpt.py:839-848). Returns (ln_f(h), new cache)."""
        new_cache = []
        for i in range(self.n_layer):
            x, kv = self._block(i, x, None if cache is None else cache[i])
            new_cache.append(kv)
        h = layer_norm(x, self.w["gpt.ln_f.weight"], self.w["gpt.ln_f.bias"], self.eps)
        return h, new_cache

    # -- A3 logits --------------------------------------------------------------------------
    def final_norm(self, h: Tensor) -> Tensor:
        return layer_norm(h, self.w["final_norm.weight"], self.w["final_norm.bias"], self.eps)

    def logits(self, h_last: Tensor) -> Tensor:
        """compute_logits (vllm_mm_gpt.py:664-688): final_norm → mel_head (+bias)."""
        y = self.final_norm(h_last)
        return y @ self.w["mel_head.weight"].t() + self.w["mel_head.bias"]

    # -- hot loop 1: autoregressive generation (Appendix A1,A3-A5) ----------------------------
    @torch.no_grad()
    def generate(self, cond: Tensor, cfg: SamplingCfg, return_debug: bool = False):
        """cond [n_cond,H] → list of mel token ids (stop id kept, XTTSv2.py:737).

        Also returns the decode-time ln_f rows (start, tok_1 … tok_{N-1}) used by the
        latent-stash equivalence test.
        """
        n_cond = cond.shape[0]
        start = self.mel_embed([self.start_token], [0])
        x = torch.cat([cond.float(), start], dim=0)
        h, cache = self.forward_rows(x, None)
        rows = [h[-1]]
        # prompt ids the penaliser sees: [1]*n_cond + [1024] (vllm_mm_gpt.py:325; hijack.py:49)
        seen = {1, self.start_token}
        toks: List[int] = []
        margins: List[float] = []
        h_last = h[-1]
        prefill_logits = None
        for step in range(cfg.max_tokens):
            z = self.logits(h_last)
            if step == 0:
                prefill_logits = z.clone()
            z = apply_repetition_penalty(z, sorted(seen), cfg.repetition_penalty)
            if return_debug:
                top2 = torch.topk(z, 2).values
                margins.append(float(top2[0] - top2[1]))
            noise = None if cfg.temperature < 1e-5 else exp_noise(cfg.seed, step, z.shape[0])
            tok = sample_token(z, cfg.temperature, cfg.top_k, cfg.top_p, noise)
            toks.append(tok)
            seen.add(tok)
            if (tok == self.stop_token and not cfg.ignore_stop) or len(toks) >= cfg.max_tokens:
                break
            # A5: k-th generated token (k = len(toks)) enters at mel position k
            e = self.mel_embed([tok], [len(toks)])
            h, cache = self.forward_rows(e, cache)
            h_last = h[-1]
            rows.append(h_last)
        out = {"tokens": toks, "decode_rows": torch.stack(rows, dim=0)}
        if return_debug:
            out["margins"] = margins
            out["prefill_logits"] = prefill_logits
            out["prefill_h"] = rows[0]
        return out

    # -- hot step 2: literal second pass (XTTSv2.py:617-687; Appendix A6) ----------------------
    @torch.no_grad()
    def second_pass_latents(self, cond: Tensor, tokens: Sequence[int]) -> Tensor:
        ids = [self.start_token] + list(tokens) + [self.stop_token] * 4
        emb = self.mel_embed(ids, list(range(len(ids))))
        x = torch.cat([cond.float(), emb], dim=0)
        h, _ = self.forward_rows(x, None)
        hs = self.final_norm(h)                      # collector sees final_norm(ln_f(h)) (vllm_mm_gpt.py:671-682)
        n_cond = cond.shape[0]
        return self.final_norm(hs[n_cond:-5]).unsqueeze(0)   # second final_norm (XTTSv2.py:685-687)

    def latents_from_decode_rows(self, decode_rows: Tensor, n_tokens: int) -> Tensor:
        """Latent-stash form: final_norm∘final_norm of the decode-time ln_f rows."""
        return self.final_norm(self.final_norm(decode_rows[:n_tokens])).unsqueeze(0)


# ----------------------------------------------------------------------------------------------
# Vocoder (models/xttsv2/components/tts/layers/xtts/hifigan_decoder.py; Appendix A7/A7')
# ----------------------------------------------------------------------------------------------
def fold_weight_norm(g: Tensor, v: Tensor) -> Tensor:
    """w = g * v / ||v||, norm over all dims but 0 (torch parametrizations.weight_norm, dim=0)."""
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * (v / n)


def interp_linear(x: Tensor, scale: float) -> Tensor:
    """Closed form of F.interpolate(x[...,L], scale_factor=s, mode='linear', align_corners=False).

    Lo = floor(L*s); r = float32(1/s); src = max(r*(j+0.5)-0.5, 0); i0=floor(src); i1=min(i0+1,L-1).
    (hifigan_decoder.py:787-800; verified against torch in tests/test_oracle_vocoder.py)
    """
    L = x.shape[-1]
    Lo = int(math.floor(L * scale))
    r = np.float32(1.0 / scale)
    j = np.arange(Lo, dtype=np.float32)
    src = np.maximum(r * (j + np.float32(0.5)) - np.float32(0.5), np.float32(0.0)).astype(np.float32)
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, L - 1)
    lam = torch.from_numpy((src - i0.astype(np.float32)).astype(np.float32))
    i0t, i1t = torch.from_numpy(i0), torch.from_numpy(i1)
    return (1.0 - lam) * x[..., i0t] + lam * x[..., i1t]


def vocoder_effective_weights(sd: Dict[str, Tensor], prefix: str = "hifigan_decoder.waveform_decoder.") -> Dict[str, Tensor]:
    """Fold weight-norm parametrisations into plain conv weights (A7'(iii)); keys without prefix."""
    out: Dict[str, Tensor] = {}
    keys = [k for k in sd if k.startswith(prefix)]
    for k in keys:
        name = k[len(prefix):]
        if name.endswith("parametrizations.weight.original0"):
            base = name[: -len("parametrizations.weight.original0")]
            g = sd[k].float()
            v = sd[prefix + base + "parametrizations.weight.original1"].float()
            out[base + "weight"] = fold_weight_norm(g, v)
        elif name.endswith("parametrizations.weight.original1"):
            continue
        else:
            out[name] = sd[k].float()
    return out


@torch.no_grad()
def hifigan_forward(w: Dict[str, Tensor], z: Tensor, g: Tensor,
                    ups=((8, 16), (8, 16), (2, 4), (2, 4)), rb_kernels=(3, 7, 11), rb_dils=(1, 3, 5),
                    collect: Optional[dict] = None) -> Tensor:
    """HifiganGenerator.forward (hifigan_decoder.py:228-260) on z [1024,T'] and g [1,512,1].

    `w` = vocoder_effective_weights(...). Returns [1,1,256*T'] fp32.
    """
    x = F.conv1d(z.unsqueeze(0), w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, w["cond_layer.weight"], w["cond_layer.bias"])
    if collect is not None:
        collect["conv_pre"] = x.clone()
    nk = len(rb_kernels)
    for i, (s, k) in enumerate(ups):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=s, padding=(k - s) // 2)
        x = x + F.conv1d(g, w[f"conds.{i}.weight"], w[f"conds.{i}.bias"])
        if collect is not None:
            collect[f"ups.{i}"] = x.clone()
        z_sum = None
        for j, rk in enumerate(rb_kernels):
            r = x
            p = f"resblocks.{i * nk + j}."
            for c, d in enumerate(rb_dils):
                xt = F.leaky_relu(r, 0.1)
                xt = F.conv1d(xt, w[p + f"convs1.{c}.weight"], w[p + f"convs1.{c}.bias"], dilation=d,
                              padding=(rk * d - d) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, w[p + f"convs2.{c}.weight"], w[p + f"convs2.{c}.bias"], padding=(rk - 1) // 2)
                r = xt + r
            z_sum = r if z_sum is None else z_sum + r
        x = z_sum / nk
        if collect is not None:
            collect[f"mrf.{i}"] = x.clone()
    x = F.leaky_relu(x)                       # default slope 0.01 (hifigan_decoder.py:257)
    x = F.conv1d(x, w["conv_post.weight"], None, padding=3)
    return torch.tanh(x)


@torch.no_grad()
def hifi_decoder_forward(w: Dict[str, Tensor], latents: Tensor, g: Tensor, collect: Optional[dict] = None) -> Tensor:
    """HifiDecoder.forward (hifigan_decoder.py:776-802): latents [1,T,1024] → wav [1,1,256*T']."""
    z = interp_linear(latents.float().transpose(1, 2), 1024 / 256).squeeze(0)
    z = interp_linear(z, 24000 / 22050)
    if collect is not None:
        collect["z"] = z.clone()
    return hifigan_forward(w, z, g, collect=collect)
