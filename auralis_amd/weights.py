"""Load-time transformation of the reference's on-disk tensors into the layouts the HIP kernels stream.

Input: the two state dicts of the reference format (written by
src/auralis/models/xttsv2/utils/checkpoint_converter.py:225-284).  Output: {packed name -> fp32 ndarray}
for aur_load_weights.  Runs once per model load on the host (not part of the hot path):

* GPT linears stay in the HF-Conv1D [in, out] layout of the file (the reference's vLLM loader transposes
  them, vllm_mm_gpt.py:723-725; the split-K kernel streams [K][N] rows directly);
* mel_head [1026,1024] -> transposed [1024,1088] (columns zero-padded to a multiple of 64);
* vocoder: weight-norm folded (w = g*v/||v||, hifigan_decoder.py weight_norm parametrisations; norm over all
  dims but 0, which is Cin for ConvTranspose1d), then Conv1d [Cout,Cin,k] -> MFMA A-operand tiles
  [Cout/MT][Cin][k][MT]; ConvTranspose1d [Cin,Cout,k] (stride s, k = 2s) -> 2-tap polyphase form over virtual
  channels v = co*s + r:  Wv[v][ci][j'] = W[ci][co][r + s*(1-j')]  (tap j' reads x[q + j' - 1]).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

Tensor = torch.Tensor
VOC_PREFIX = "hifigan_decoder.waveform_decoder."
HEAD_PAD = 1088


def fold_weight_norm(g: Tensor, v: Tensor) -> Tensor:
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * (v / n)


def pack_conv(w: Tensor) -> Tensor:
    """Conv1d weight [Cout, Cin, k] -> [Cout/MT, Cin, k, MT] with MT = 64 (32 when Cout % 64 != 0)."""
    cout, cin, k = w.shape
    mt = 64 if cout % 64 == 0 else 32
    assert cout % mt == 0
    return w.reshape(cout // mt, mt, cin, k).permute(0, 2, 3, 1).contiguous()


def pack_conv_f16(w: Tensor) -> Tensor:
    """Conv1d weight [Cout, Cin, k] -> fp16 [Cout/MT, Cin/16, k, MT, 16] (A fragments of v_mfma_f32_32x32x16_f16:
    16 input channels of one tap per MFMA)."""
    cout, cin, k = w.shape
    mt = 64 if cout % 64 == 0 else 32
    assert cout % mt == 0 and cin % 16 == 0
    return w.reshape(cout // mt, mt, cin // 16, 16, k).permute(0, 2, 4, 1, 3).contiguous().to(torch.float16)


def f16_as_f32_words(t: Tensor) -> np.ndarray:
    """Reinterpret an fp16 tensor's bytes as float32 words (aur_load_weights moves opaque 4-byte words)."""
    a = np.ascontiguousarray(t.numpy())
    assert a.dtype == np.float16 and a.size % 2 == 0
    return a.view(np.float32).reshape(-1)


def polyphase_convT(w: Tensor, stride: int) -> Tensor:
    """ConvTranspose1d weight [Cin, Cout, k=2s] -> virtual Conv1d weight [Cout*s, Cin, 2]."""
    cin, cout, k = w.shape
    assert k == 2 * stride
    # W[ci][co][r + s*m] with m in {0,1}; tap j' = 1 - m
    wv = w.reshape(cin, cout, 2, stride)              # [ci][co][m][r]
    wv = wv.permute(1, 3, 0, 2)                       # [co][r][ci][m]
    wv = wv.flip(-1)                                  # j' = 1 - m
    return wv.reshape(cout * stride, cin, 2).contiguous()


def _effective(sd: Dict[str, Tensor], base: str) -> Tensor:
    k0 = base + "parametrizations.weight.original0"
    if k0 in sd:
        return fold_weight_norm(sd[k0].float(), sd[base + "parametrizations.weight.original1"].float())
    if base + "weight_g" in sd:   # legacy torch.nn.utils.weight_norm naming
        return fold_weight_norm(sd[base + "weight_g"].float(), sd[base + "weight_v"].float())
    return sd[base + "weight"].float()


def pack_vocoder(xtts_sd: Dict[str, Tensor], upsample_rates=(8, 8, 2, 2), fp16: bool = True) -> Dict[str, np.ndarray]:
    p = VOC_PREFIX
    out: Dict[str, Tensor] = {}
    out["voc.conv_pre.wp"] = pack_conv(_effective(xtts_sd, p + "conv_pre."))
    out["voc.conv_pre.bias"] = xtts_sd[p + "conv_pre.bias"].float()
    out["voc.cond_layer.w"] = xtts_sd[p + "cond_layer.weight"].float().squeeze(-1)
    out["voc.cond_layer.b"] = xtts_sd[p + "cond_layer.bias"].float()
    for i, s in enumerate(upsample_rates):
        w = _effective(xtts_sd, p + f"ups.{i}.")
        out[f"voc.ups.{i}.wp"] = pack_conv(polyphase_convT(w, s))
        out[f"voc.ups.{i}.bias"] = xtts_sd[p + f"ups.{i}.bias"].float()
        out[f"voc.conds.{i}.w"] = xtts_sd[p + f"conds.{i}.weight"].float().squeeze(-1)
        out[f"voc.conds.{i}.b"] = xtts_sd[p + f"conds.{i}.bias"].float()
    n_rb = 3 * len(upsample_rates)
    for n in range(n_rb):
        for grp, tag in (("convs1", "c1"), ("convs2", "c2")):
            for q in range(3):
                base = p + f"resblocks.{n}.{grp}.{q}."
                out[f"voc.rb.{n}.{tag}.{q}.wp"] = pack_conv(_effective(xtts_sd, base))
                out[f"voc.rb.{n}.{tag}.{q}.bias"] = xtts_sd[base + "bias"].float()
    out["voc.conv_post.w"] = _effective(xtts_sd, p + "conv_post.").reshape(-1, 7)
    res = {k: np.ascontiguousarray(v.numpy(), dtype=np.float32) for k, v in out.items()}
    if fp16:
        # fp16-input MFMA variant of every conv ("voc16.<layer>.wp"); biases/conditioning stay fp32
        res["voc16.conv_pre.wp"] = f16_as_f32_words(pack_conv_f16(_effective(xtts_sd, p + "conv_pre.")))
        for i, s in enumerate(upsample_rates):
            res[f"voc16.ups.{i}.wp"] = f16_as_f32_words(pack_conv_f16(polyphase_convT(_effective(xtts_sd, p + f"ups.{i}."), s)))
        for n in range(n_rb):
            for grp, tag in (("convs1", "c1"), ("convs2", "c2")):
                for q in range(3):
                    res[f"voc16.rb.{n}.{tag}.{q}.wp"] = f16_as_f32_words(
                        pack_conv_f16(_effective(xtts_sd, p + f"resblocks.{n}.{grp}.{q}.")))
    return res


def pack_gpt(gpt_sd: Dict[str, Tensor], xtts_sd: Dict[str, Tensor]) -> Dict[str, np.ndarray]:
    out: Dict[str, Tensor] = {}
    out["gpt.wte"] = gpt_sd["gpt.wte.weight"]
    out["gpt.wpe"] = gpt_sd["gpt.wpe.emb.weight"]
    n_layer = 1 + max(int(k.split(".")[2]) for k in gpt_sd if k.startswith("gpt.h."))
    for i in range(n_layer):
        src, dst = f"gpt.h.{i}.", f"gpt.h.{i}."
        for ln in ("ln_1", "ln_2"):
            out[dst + ln + ".w"] = gpt_sd[src + ln + ".weight"]
            out[dst + ln + ".b"] = gpt_sd[src + ln + ".bias"]
        for lin in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
            out[dst + lin + ".w"] = gpt_sd[src + lin + ".weight"]      # [in, out] as on disk
            out[dst + lin + ".b"] = gpt_sd[src + lin + ".bias"]
    out["gpt.ln_f.w"] = gpt_sd["gpt.ln_f.weight"]
    out["gpt.ln_f.b"] = gpt_sd["gpt.ln_f.bias"]
    out["final_norm.w"] = gpt_sd["final_norm.weight"]
    out["final_norm.b"] = gpt_sd["final_norm.bias"]
    head = gpt_sd["mel_head.weight"].float()                           # [V, H]
    V, H = head.shape
    headT = torch.zeros(H, HEAD_PAD, dtype=torch.float32)
    headT[:, :V] = head.t()
    hb = torch.zeros(HEAD_PAD, dtype=torch.float32)
    hb[:V] = gpt_sd["mel_head.bias"].float()
    out["mel_head.wT"] = headT
    out["mel_head.b"] = hb
    out["text_emb"] = xtts_sd["text_embedding.weight"]
    out["text_pos"] = xtts_sd["text_pos_embedding.emb.weight"]
    return {k: np.ascontiguousarray(v.float().numpy(), dtype=np.float32) for k, v in out.items()}


def pack_all(gpt_sd, xtts_sd) -> Dict[str, np.ndarray]:
    d = pack_vocoder(xtts_sd)
    d.update(pack_gpt(gpt_sd, xtts_sd))
    return d


# ---- CPU emulation of what conv1d_mfma_kernel computes from packed weights (host-logic tests only) -------
def emulate_packed_conv(x: Tensor, wp: Tensor, bias, ks: int, dil: int, padl: int, slope: float,
                        ups_s: int = 0, ups_p: int = 0) -> Tensor:
    """x [Cin, L] -> y per the kernel's definition (vocoder_kernels.h); used to validate the packing."""
    mtiles, cin, k, mt = wp.shape
    assert k == ks
    w = wp.permute(0, 3, 1, 2).reshape(mtiles * mt, cin, k)          # [Mtot][Cin][k]
    L = x.shape[-1]
    xa = torch.where(x > 0, x, x * slope)
    n_q = L + 1 if ups_s else L
    halo = (ks - 1) * dil
    xp = torch.zeros(cin, n_q + halo + padl + 8)
    xp[:, padl:padl + L] = xa
    y = torch.zeros(w.shape[0], n_q)
    for j in range(ks):
        y += w[:, :, j] @ xp[:, j * dil: j * dil + n_q]
    if not ups_s:
        return y + (0 if bias is None else bias[:, None])
    cout = w.shape[0] // ups_s
    out = torch.zeros(cout, L * ups_s)
    for v in range(w.shape[0]):
        co, r = divmod(v, ups_s)
        t = torch.arange(n_q) * ups_s + r - ups_p
        ok = (t >= 0) & (t < L * ups_s)
        out[co, t[ok]] = y[v, ok]
    return out + (0 if bias is None else bias[:, None])
