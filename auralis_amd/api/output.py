"""TTSOutput — container for synthesized audio, API-compatible with the reference on the path's surface
(src/auralis/common/definitions/output.py:16-329: array, sample_rate, start_time, token_length, combine_outputs,
to_tensor, to_bytes, save, resample, get_info, from_tensor, from_file, change_speed).  Codec back-ends differ:
the reference goes through torchaudio/librosa/sounddevice (absent offline); wav, raw PCM and FLAC are written and read
natively (api/codecs.py, api/flac.py), mp3 / opus / aac go to torchaudio or an ffmpeg executable when one is present."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np


@dataclass
class TTSOutput:
    array: Union[np.ndarray, bytes]
    sample_rate: int = 24000
    bit_depth: int = 32
    bit_rate: int = 192
    compression: int = 10
    channel: int = 1
    start_time: Optional[float] = None
    end_time: Optional[float] = None
    token_length: Optional[int] = None

    def __post_init__(self):
        if isinstance(self.array, np.ndarray):
            self.array = np.asarray(self.array, dtype=np.float32).reshape(-1)

    # -- construction ---------------------------------------------------------------------------------
    @staticmethod
    def combine_outputs(outputs: List["TTSOutput"]) -> "TTSOutput":
        """Concatenate chunk outputs in order; sample rate of the first (output.py:95-111)."""
        if not outputs:
            raise ValueError("combine_outputs needs at least one output")
        if len(outputs) == 1:   # a one-chunk request (every text up to the character limit): the chunk's own array, not a 1.25 MB copy of it
            o = outputs[0]
            return TTSOutput(array=o.array, sample_rate=o.sample_rate, token_length=o.token_length or None, start_time=o.start_time)
        return TTSOutput(array=np.concatenate([o.array for o in outputs]), sample_rate=outputs[0].sample_rate,
                         token_length=sum(o.token_length or 0 for o in outputs) or None,
                         start_time=outputs[0].start_time)

    @classmethod
    def from_tensor(cls, tensor, sample_rate: int = 24000) -> "TTSOutput":
        arr = tensor.detach().cpu().numpy() if hasattr(tensor, "detach") else np.asarray(tensor)
        return cls(array=arr.squeeze(), sample_rate=sample_rate)

    @classmethod
    def from_file(cls, filename: Union[str, Path]) -> "TTSOutput":
        from . import codecs
        a, sr = codecs.decode(str(filename))
        return cls(array=a, sample_rate=sr)

    # -- views ----------------------------------------------------------------------------------------
    def to_tensor(self):
        import torch
        return torch.from_numpy(np.asarray(self.array))

    def get_info(self) -> Tuple[int, int, float]:
        n = len(self.array)
        return n, self.sample_rate, n / float(self.sample_rate)

    def to_bytes(self, format: str = "wav", sample_width: int = 2) -> bytes:
        """output.py:119-187: 'mp3', 'opus', 'aac', 'flac', 'wav', 'pcm'; samples clamped to [-1, 1]."""
        from . import codecs
        return codecs.encode(np.asarray(self.array, dtype=np.float32), self.sample_rate, format, sample_width,
                             bit_rate=self.bit_rate, compression=self.compression)

    def save(self, filename: Union[str, Path], sample_rate: Optional[int] = None, format: Optional[str] = None) -> None:
        out = self if sample_rate in (None, self.sample_rate) else self.resample(sample_rate)
        fmt = format or (Path(str(filename)).suffix.lstrip(".") or "wav")
        Path(str(filename)).write_bytes(out.to_bytes(format=fmt))

    # -- transforms -----------------------------------------------------------------------------------
    def resample(self, new_sample_rate: int) -> "TTSOutput":
        if new_sample_rate == self.sample_rate:
            return self
        from math import gcd

        from scipy.signal import resample_poly
        g = gcd(int(new_sample_rate), int(self.sample_rate))
        y = resample_poly(np.asarray(self.array, dtype=np.float64), new_sample_rate // g, self.sample_rate // g)
        return TTSOutput(array=y.astype(np.float32), sample_rate=new_sample_rate, token_length=self.token_length)

    def change_speed(self, speed_factor: float) -> "TTSOutput":
        """Time-stretch with an STFT phase vocoder (n_fft 2048, hop 512) and peak-normalise, as the reference does
        through librosa (output.py:40-92)."""
        if speed_factor <= 0:
            raise ValueError("Speed factor must be positive")
        if speed_factor == 1.0:
            return self
        from scipy.signal import istft, stft
        n_fft, hop = 2048, 512
        x = np.asarray(self.array, dtype=np.float32)
        _, _, D = stft(x, nperseg=n_fft, noverlap=n_fft - hop, window="hann", boundary="zeros", padded=True)
        n_frames = D.shape[1]
        steps = np.arange(0, n_frames - 1, speed_factor)
        phase_adv = np.linspace(0, np.pi * hop, D.shape[0])
        phase = np.angle(D[:, 0])
        out = np.zeros((D.shape[0], len(steps)), dtype=np.complex128)
        Dp = np.pad(D, ((0, 0), (0, 2)))
        for i, st in enumerate(steps):
            k = int(st)
            a = st - k
            mag = (1 - a) * np.abs(Dp[:, k]) + a * np.abs(Dp[:, k + 1])
            out[:, i] = mag * np.exp(1j * phase)
            dphi = np.angle(Dp[:, k + 1]) - np.angle(Dp[:, k]) - phase_adv
            dphi -= 2 * np.pi * np.round(dphi / (2 * np.pi))
            phase += phase_adv + dphi
        _, y = istft(out, nperseg=n_fft, noverlap=n_fft - hop, window="hann")
        peak = np.max(np.abs(y)) if y.size else 0.0
        if peak > 0:
            y = y / peak
        return TTSOutput(array=y.astype(np.float32), sample_rate=self.sample_rate)

    def play(self) -> None:
        """Blocking playback on the default sound device (reference: output.py:287-303, sounddevice).  The package is optional."""
        try:
            import sounddevice as sd
        except ImportError as e:
            raise RuntimeError("audio playback needs the optional package sounddevice (not part of the synthesis path)") from e
        sd.play(np.clip(np.asarray(self.array, np.float32), -1.0, 1.0), self.sample_rate, blocksize=2048)
        sd.wait()

    def display(self):
        """Notebook audio widget (reference: output.py:305-319): returns the IPython Audio object, or None (with a hint) when
        IPython is not there or the widget cannot be built."""
        try:
            from IPython.display import Audio, display
            widget = Audio(self.to_bytes(format="wav"), rate=self.sample_rate, autoplay=False)
            display(widget)
            return widget
        except Exception as e:
            print(f"Could not display audio widget: {e}\nTry using .play() method instead")
            return None

    def preview(self) -> None:
        """Widget in a notebook, sound device otherwise (reference: output.py:321-329); never raises."""
        try:
            if self.display() is None:
                self.play()
        except Exception as e:
            print(f"Error playing audio: {e}")
