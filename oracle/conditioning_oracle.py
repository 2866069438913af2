"""CPU / PyTorch restatement of the once-per-speaker conditioning networks: reference audio -> (gpt_cond_latent [1,32,1024],
speaker_embedding [1,512,1]).

TEST INFRASTRUCTURE ONLY: the checker the HIP path (aur_compute_conditioning) is compared with, and bench.py's CPU timing of
the same stage.  Nothing under auralis_amd/ imports it.

Reference: XTTSv2Engine.get_conditioning_latents / get_gpt_cond_latents / _get_speaker_embedding
(src/auralis/models/xttsv2/XTTSv2.py:312-328, 349-468), wav_to_mel_cloning (src/auralis/common/utilities.py:9-71),
ConditioningEncoder (components/tts/layers/xtts/latent_encoder.py:134-253), PerceiverResampler
(.../perceiver_encoder.py:363-485) and ResNetSpeakerEncoder (.../hifigan_decoder.py:485-689), written as plain functions over
the checkpoint's state-dict tensors (xtts-v2.safetensors keys).  The three network bodies are pinned
against the reference's own module classes (tests/test_conditioning.py, tests/golden/cond_female_6s.npz); the mel front-end
restates torchaudio's published algorithm (torchaudio is absent offline) and is cross-checked against scipy only.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from auralis_amd.conditioning import load_audio, mel_filterbank, read_wav, resample  # noqa: F401  (host loader, shared)

Tensor = torch.Tensor


def mel_spectrogram(wav: Tensor, sample_rate: int, n_fft: int, win_length: int, hop_length: int, n_mels: int,
                    f_min: float = 0.0, f_max: float = None, window: str = "hann", slaney_norm: bool = False) -> Tensor:
    """torchaudio.transforms.MelSpectrogram (power 2, center, reflect padding, one-sided) -> [B, n_mels, frames]."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    win = (torch.hann_window if window == "hann" else torch.hamming_window)(win_length, periodic=True,
                                                                             device=wav.device, dtype=wav.dtype)
    spec = torch.stft(wav, n_fft, hop_length=hop_length, win_length=win_length, window=win, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = mel_filterbank(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, slaney_norm).to(wav.device)
    return torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)


# ----------------------------------------------------------------------------------------------- ConditioningEncoder
def _group_norm32(x: Tensor, w: Tensor, b: Tensor, groups: int = 32) -> Tensor:
    return F.group_norm(x.float(), groups, w.float(), b.float(), 1e-5).to(x.dtype)


def conditioning_encoder(sd: Dict[str, Tensor], mel: Tensor, n_heads: int = 16, prefix: str = "conditioning_encoder.") -> Tensor:
    """mel [B,80,T] -> [B,1024,T] (latent_encoder.py:209-253; AttentionBlock 134-206, QKVAttention 95-131).

    Quirks kept: the block returns x_proj(norm(x)) + proj_out(attn) — the residual is the NORMALISED x; qkv channels are
    head-major (q|k|v per head) and q, k are each scaled by ch^-1/4."""
    h = F.conv1d(mel, sd[prefix + "init.weight"], sd[prefix + "init.bias"])
    i = 0
    while prefix + f"attn.{i}.norm.weight" in sd:
        p = prefix + f"attn.{i}."
        x = _group_norm32(h, sd[p + "norm.weight"], sd[p + "norm.bias"])
        qkv = F.conv1d(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
        bs, width, length = qkv.shape
        ch = width // (3 * n_heads)
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
        scale = 1.0 / math.sqrt(math.sqrt(ch))
        wgt = torch.einsum("bct,bcs->bts", q * scale, k * scale)
        wgt = torch.softmax(wgt.float(), dim=-1).type(wgt.dtype)
        a = torch.einsum("bts,bcs->bct", wgt, v).reshape(bs, -1, length)
        h = x + F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
        i += 1
    return h


# ----------------------------------------------------------------------------------------------- PerceiverResampler
def perceiver_resampler(sd: Dict[str, Tensor], x: Tensor, heads: int = 8, prefix: str = "conditioning_perceiver.") -> Tensor:
    """x [B,T,1024] -> [B,32,1024] (perceiver_encoder.py:363-442).  Cross-attention keys/values are
    concat(latents, context) (:476-477); GEGLU = gelu(gate) * x with the gate in the second half (:334-335);
    final RMSNorm = normalize(x) * sqrt(dim) * gamma (:275-276)."""
    B = x.shape[0]
    lat = sd[prefix + "latents"].unsqueeze(0).expand(B, -1, -1)
    li = 0
    while prefix + f"layers.{li}.0.to_q.weight" in sd:
        p = prefix + f"layers.{li}."
        ctx = torch.cat((lat, x), dim=-2)
        q = F.linear(lat, sd[p + "0.to_q.weight"])
        k, v = F.linear(ctx, sd[p + "0.to_kv.weight"]).chunk(2, dim=-1)

        def split(t):
            return t.reshape(B, t.shape[1], heads, -1).transpose(1, 2)
        q, k, v = split(q), split(k), split(v)
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * (q.shape[-1] ** -0.5)
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
        out = out.transpose(1, 2).reshape(B, lat.shape[1], -1)
        lat = F.linear(out, sd[p + "0.to_out.weight"]) + lat
        hgate = F.linear(lat, sd[p + "1.0.weight"], sd[p + "1.0.bias"])
        a, gate = hgate.chunk(2, dim=-1)
        lat = F.linear(F.gelu(gate) * a, sd[p + "1.2.weight"], sd[p + "1.2.bias"]) + lat
        li += 1
    return F.normalize(lat, dim=-1) * (lat.shape[-1] ** 0.5) * sd[prefix + "norm.gamma"]


# ----------------------------------------------------------------------------------------------- ResNetSpeakerEncoder
def _bn(sd, p, x, dims2d=True):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _se_block(sd: Dict[str, Tensor], p: str, x: Tensor, stride: int) -> Tensor:
    r = x
    y = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
    y = _bn(sd, p + "bn1.", F.relu(y))                     # conv -> relu -> bn (hifigan_decoder.py:386-389)
    y = _bn(sd, p + "bn2.", F.conv2d(y, sd[p + "conv2.weight"], None, padding=1))
    s = y.mean(dim=(2, 3))
    s = torch.sigmoid(F.linear(F.relu(F.linear(s, sd[p + "se.fc.0.weight"], sd[p + "se.fc.0.bias"])),
                               sd[p + "se.fc.2.weight"], sd[p + "se.fc.2.bias"]))
    y = y * s[:, :, None, None]
    if p + "downsample.0.weight" in sd:
        r = _bn(sd, p + "downsample.1.", F.conv2d(r, sd[p + "downsample.0.weight"], None, stride=stride))
    return F.relu(y + r)


def speaker_encoder_from_mel(sd: Dict[str, Tensor], mel: Tensor, prefix: str = "hifigan_decoder.speaker_encoder.",
                             l2_norm: bool = True) -> Tensor:
    """mel power spectrogram [B,64,T] -> [B,512] (ResNetSpeakerEncoder.forward after torch_spec, :602-646)."""
    x = torch.log(mel + 1e-6)
    x = F.instance_norm(x).unsqueeze(1)
    x = _bn(sd, prefix + "bn1.", F.relu(F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1)))
    for li, (n_blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(n_blocks):
            x = _se_block(sd, prefix + f"layer{li}.{bi}.", x, stride if bi == 0 else 1)
    x = x.reshape(x.size(0), -1, x.size(-1))
    a = F.conv1d(x, sd[prefix + "attention.0.weight"], sd[prefix + "attention.0.bias"])
    a = _bn(sd, prefix + "attention.2.", F.relu(a))
    w = torch.softmax(F.conv1d(a, sd[prefix + "attention.3.weight"], sd[prefix + "attention.3.bias"]), dim=2)
    mu = torch.sum(x * w, dim=2)
    sg = torch.sqrt((torch.sum((x ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-5))
    e = F.linear(torch.cat((mu, sg), 1), sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    return F.normalize(e, p=2, dim=1) if l2_norm else e


def speaker_embedding(sd: Dict[str, Tensor], audio: Tensor, sr: int) -> Tensor:
    """XTTSv2.py:312-328: resample to 16 kHz -> pre-emphasis 0.97 -> 64-mel (n_fft 512, win 400, hop 160, hamming) ->
    ResNet-SE -> L2 norm -> [1,512,1]."""
    a16 = resample(audio, sr, 16000)
    x = F.pad(a16.unsqueeze(1), (1, 0), "reflect")
    x = F.conv1d(x, torch.tensor([-0.97, 1.0], dtype=a16.dtype, device=a16.device).view(1, 1, -1)).squeeze(1)
    mel = mel_spectrogram(x, 16000, 512, 400, 160, 64, window="hamming", slaney_norm=False)
    return speaker_encoder_from_mel(sd, mel).unsqueeze(-1)


# ----------------------------------------------------------------------------------------------- top level
def gpt_cond_latents(sd: Dict[str, Tensor], audio: Tensor, sr: int, length: int = 30, chunk_length: int = 6,
                     n_heads: int = 16) -> Tensor:
    """XTTSv2.py:349-407: <=length s of audio in chunk_length-s chunks (shorter than 0.33 s skipped) -> mel (n_fft 2048,
    hop 256, win 1024, 80 mels, 0-8 kHz, slaney) -> log(clamp 1e-5) / mel_stats -> encoder -> perceiver -> mean -> [1,32,1024]."""
    if sr != 22050:
        audio = resample(audio, sr, 22050)
    if length > 0:
        audio = audio[:, : 22050 * length]
    embs: List[Tensor] = []
    for i in range(0, audio.shape[1], 22050 * chunk_length):
        chunk = audio[:, i: i + 22050 * chunk_length]
        if chunk.size(-1) < 22050 * 0.33:
            continue
        mel = mel_spectrogram(chunk, 22050, 2048, 1024, 256, 80, 0.0, 8000.0, "hann", slaney_norm=True)
        mel = torch.log(torch.clamp(mel, min=1e-5)) / sd["mel_stats"].to(mel.device)[None, :, None]
        conds = conditioning_encoder(sd, mel, n_heads)
        embs.append(perceiver_resampler(sd, conds.permute(0, 2, 1)))
    if not embs:
        raise ValueError("reference audio shorter than 0.33 s")
    return torch.stack(embs).mean(dim=0)


@torch.no_grad()
def get_conditioning_latents(sd: Dict[str, Tensor], audio_reference: Union[str, bytes, Sequence], max_ref_length: int = 30,
                             gpt_cond_len: int = 6, gpt_cond_chunk_len: int = 6, sound_norm_refs: bool = False,
                             load_sr: int = 22050, device: Union[str, torch.device] = "cpu") -> Tuple[Tensor, Tensor]:
    """XTTSv2.py:409-468 -> (gpt_cond_latent [1,32,1024], speaker_embedding [1,512,1]) on `device`."""
    refs = list(audio_reference) if isinstance(audio_reference, (list, tuple)) else [audio_reference]
    sd = {k: v.to(device=device, dtype=torch.float32) for k, v in sd.items()
          if k.startswith(("conditioning_", "hifigan_decoder.speaker_encoder.", "mel_stats"))}
    embs, audios = [], []
    for ref in refs:
        a = load_audio(ref, load_sr)[:, : load_sr * max_ref_length].to(device)
        if sound_norm_refs:
            a = (a / torch.abs(a).max()) * 0.75
        embs.append(speaker_embedding(sd, a, load_sr))
        audios.append(a)
    cond = gpt_cond_latents(sd, torch.cat(audios, dim=-1), load_sr, length=gpt_cond_len, chunk_length=gpt_cond_chunk_len)
    return cond, torch.stack(embs).mean(dim=0)
