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

import math

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


# ---- speaker conditioning (cond_net.h / cond_kernels.h) ----------------------------------------------------------------------
def _pad2(a: np.ndarray, rows: int, cols: int) -> np.ndarray:
    out = np.zeros((rows, cols), np.float32)
    out[: a.shape[0], : a.shape[1]] = a
    return out


def _pad1(a: np.ndarray, n: int, fill: float = 0.0) -> np.ndarray:
    out = np.full((n,), fill, np.float32)
    out[: a.shape[0]] = a
    return out


def _ceil(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _dft_matrix(n_fft: int) -> np.ndarray:
    """[n_fft][pad128(2 * bins)]: column 2k = cos(2 pi k n / N), 2k + 1 = -sin (float64 -> float32)."""
    bins = n_fft // 2 + 1
    k = np.arange(bins, dtype=np.float64)[None, :]
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * ((k * n) % n_fft) / n_fft
    m = np.zeros((n_fft, _ceil(2 * bins, 128)), np.float32)
    m[:, 0: 2 * bins: 2] = np.cos(ang)
    m[:, 1: 2 * bins: 2] = -np.sin(ang)
    return m


def _window(kind: str, win_length: int, n_fft: int) -> np.ndarray:
    w = (torch.hann_window if kind == "hann" else torch.hamming_window)(win_length, periodic=True, dtype=torch.float64).numpy()
    out = np.zeros(n_fft, np.float32)
    left = (n_fft - win_length) // 2          # torch.stft centres a short window inside n_fft
    out[left: left + win_length] = w
    return out


def _resample_table(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """[new][2 * width + orig] kernel of torchaudio.functional.resample (same formulas as conditioning.resample)."""
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * (base / orig)
    return kern.astype(np.float32)


def pack_conditioning(xtts_sd) -> Dict[str, np.ndarray]:
    """Once-per-speaker modules of xtts-v2.safetensors -> the "cond.*" tensors the HIP conditioning path reads: every linear
    map as [K][N] (K padded to 16, N to 64 / 128 with zeros), BatchNorm folded to (scale, shift), the ResNet's 32-channel
    stage padded to 64 channels, plus the fixed tables of the mel front-ends (windows, DFT matrices, mel filterbanks, the
    22 050 -> 16 000 Hz resampling kernel)."""
    from . import conditioning as Cn

    def f(name):
        return xtts_sd[name].detach().to(torch.float32).cpu().numpy()
    d: Dict[str, np.ndarray] = {}
    d["cond.mel_stats"] = f("mel_stats").reshape(80) if "mel_stats" in xtts_sd else np.ones(80, np.float32)
    d["cond.win_gpt"] = _window("hann", 1024, 2048)
    d["cond.win_spk"] = _window("hamming", 400, 512)
    d["cond.dft_gpt"] = _dft_matrix(2048)
    d["cond.dft_spk"] = _dft_matrix(512)
    d["cond.fb_gpt"] = _pad2(Cn.mel_filterbank(1025, 0.0, 8000.0, 80, 22050, True).numpy(), _ceil(1025, 16), 128)
    d["cond.fb_spk"] = _pad2(Cn.mel_filterbank(257, 0.0, 8000.0, 64, 16000, False).numpy(), _ceil(257, 16), 128)
    d["cond.rs_22050_16000"] = _resample_table(22050, 16000)
    # ConditioningEncoder
    e = "conditioning_encoder."
    d["cond.enc.init.w"] = np.ascontiguousarray(f(e + "init.weight")[:, :, 0].T)
    d["cond.enc.init.b"] = f(e + "init.bias")
    i = 0
    while e + f"attn.{i}.norm.weight" in xtts_sd:
        p, q = e + f"attn.{i}.", f"cond.enc.{i}."
        d[q + "gn.w"], d[q + "gn.b"] = f(p + "norm.weight"), f(p + "norm.bias")
        d[q + "qkv.w"] = np.ascontiguousarray(f(p + "qkv.weight")[:, :, 0].T)
        d[q + "qkv.b"] = f(p + "qkv.bias")
        d[q + "proj.w"] = np.ascontiguousarray(f(p + "proj_out.weight")[:, :, 0].T)
        d[q + "proj.b"] = f(p + "proj_out.bias")
        i += 1
    # PerceiverResampler
    e = "conditioning_perceiver."
    d["cond.per.latents"] = f(e + "latents")
    d["cond.per.norm.g"] = f(e + "norm.gamma")
    i = 0
    while e + f"layers.{i}.0.to_q.weight" in xtts_sd:
        p, q = e + f"layers.{i}.", f"cond.per.{i}."
        d[q + "q.w"] = np.ascontiguousarray(f(p + "0.to_q.weight").T)
        d[q + "kv.w"] = np.ascontiguousarray(f(p + "0.to_kv.weight").T)
        d[q + "out.w"] = np.ascontiguousarray(f(p + "0.to_out.weight").T)
        w1, w2 = f(p + "1.0.weight"), f(p + "1.2.weight")           # [2*inner][1024], [1024][inner]
        inner = w2.shape[1]
        d[q + "ff1.w"] = _pad2(w1.T, 1024, _ceil(2 * inner, 128))
        d[q + "ff1.b"] = _pad1(f(p + "1.0.bias"), _ceil(2 * inner, 128))
        d[q + "ff2.w"] = _pad2(w2.T, _ceil(inner, 16), 1024)
        d[q + "ff2.b"] = f(p + "1.2.bias")
        i += 1
    # ResNet-SE speaker encoder (NHWC, im2col row = (ky, kx, cin))
    s, o = "hifigan_decoder.speaker_encoder.", "cond.spk."

    def conv3(name, cin_pad, cout_pad):
        w = f(name)                                  # [Cout][Cin][3][3]
        co, ci = w.shape[:2]
        m = np.zeros((9, cin_pad, cout_pad), np.float32)
        m[:, :ci, :co] = w.transpose(2, 3, 1, 0).reshape(9, ci, co)
        return _pad2(m.reshape(9 * cin_pad, cout_pad), _ceil(9 * cin_pad, 16), cout_pad)

    def bn(prefix, out_prefix, cpad):
        sc = f(prefix + "weight") / np.sqrt(f(prefix + "running_var") + 1e-5)
        d[out_prefix + "scale"] = _pad1(sc, cpad)
        d[out_prefix + "shift"] = _pad1(f(prefix + "bias") - f(prefix + "running_mean") * sc, cpad)
    d[o + "conv1.w"] = conv3(s + "conv1.weight", 1, 64)
    d[o + "conv1.b"] = _pad1(f(s + "conv1.bias"), 64)
    bn(s + "bn1.", o + "bn1.", 64)
    cin = 64
    for li, (planes, blocks) in enumerate(zip((32, 64, 128, 256), (3, 4, 6, 3)), start=1):
        cp = max(planes, 64)
        for bi in range(blocks):
            p, q = s + f"layer{li}.{bi}.", o + f"layer{li}.{bi}."
            d[q + "conv1.w"] = conv3(p + "conv1.weight", cin, cp)
            bn(p + "bn1.", q + "bn1.", cp)
            d[q + "conv2.w"] = conv3(p + "conv2.weight", cp, cp)
            bn(p + "bn2.", q + "bn2.", cp)
            w1, w2 = f(p + "se.fc.0.weight"), f(p + "se.fc.2.weight")    # [Cr][C], [C][Cr]
            d[q + "se1.w"] = _pad2(w1, w1.shape[0], cp)
            d[q + "se1.b"] = f(p + "se.fc.0.bias")
            d[q + "se2.w"] = _pad2(w2, cp, w2.shape[1])
            d[q + "se2.b"] = _pad1(f(p + "se.fc.2.bias"), cp)
            if p + "downsample.0.weight" in xtts_sd:
                wd = f(p + "downsample.0.weight")[:, :, 0, 0]            # [Cout][Cin]
                d[q + "down.w"] = _pad2(wd.T, cin, cp)
                bn(p + "downsample.1.", q + "down.", cp)
            cin = cp
    d[o + "att0.w"] = np.ascontiguousarray(f(s + "attention.0.weight")[:, :, 0].T)     # [2048][128]
    d[o + "att0.b"] = f(s + "attention.0.bias")
    bn(s + "attention.2.", o + "att2.", 128)
    d[o + "att3.w"] = np.ascontiguousarray(f(s + "attention.3.weight")[:, :, 0].T)     # [128][2048]
    d[o + "att3.b"] = f(s + "attention.3.bias")
    d[o + "fc.w"] = np.ascontiguousarray(f(s + "fc.weight").T)                         # [4096][512]
    d[o + "fc.b"] = f(s + "fc.bias")
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in d.items()}


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
