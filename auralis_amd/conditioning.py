"""Reference-audio loader in front of the speaker-conditioning kernels: decode (wav / FLAC), mono, clip, resample
(src/auralis/common/utilities.py:74-98 load_audio), plus the mel filter-bank tables weights.py packs for the HIP mel
front-end (utilities.py:9-71 wav_to_mel_cloning; hifigan_decoder.py:537-548).  Host-side plumbing only: the conditioning
networks themselves run in HIP (aur_compute_conditioning, csrc/cond_net.h).  Resampling and the filter banks restate
torchaudio's published algorithms because torchaudio is not available offline ("parity unpinned" for that part); the PyTorch
restatement of the networks that the tests compare the HIP path with lives in oracle/conditioning_oracle.py.
"""
from __future__ import annotations

import io
import math
import wave
from typing import Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------- audio front-end
def read_wav(src: Union[str, bytes]) -> Tuple[Tensor, int]:
    """Reference audio (path or bytes) -> mono float32 [1, n], sample rate.  RIFF/WAVE (8/16/24/32-bit int, 32-bit float) and
    FLAC are decoded natively; other containers through torchaudio / ffmpeg when present (api/codecs.py) - the reference uses
    torchaudio.load (utilities.py:74-98)."""
    from .api import codecs
    a, sr = codecs.decode(src)
    return torch.from_numpy(a).unsqueeze(0), sr


def resample(x: Tensor, orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tensor:
    """Band-limited sinc interpolation with a Hann window (the algorithm of torchaudio.functional.resample)."""
    if orig == new:
        return x
    g = math.gcd(int(orig), int(new))
    orig, new = int(orig) // g, int(new) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64, device=x.device)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64, device=x.device)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    kern = kern.to(x.dtype)
    n = x.shape[-1]
    y = F.conv1d(F.pad(x.reshape(-1, n), (width, width + orig))[:, None], kern, stride=orig)
    y = y.transpose(1, 2).reshape(x.shape[:-1] + (-1,))
    return y[..., : math.ceil(new * n / orig)]


def load_audio(src: Union[str, bytes], sampling_rate: int) -> Tensor:
    """utilities.py:74-98: load, mono, resample, clip to [-1, 1]."""
    a, sr = read_wav(src)
    return resample(a, sr, sampling_rate).clamp_(-1.0, 1.0)


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int, slaney_norm: bool) -> Tensor:
    """torchaudio.functional.melscale_fbanks with mel_scale='htk' -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    if slaney_norm:
        fb = fb * (2.0 / (f_pts[2: n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb.float()
