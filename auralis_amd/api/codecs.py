"""Audio containers for TTSOutput.to_bytes / from_file and for reference audio.

The reference delegates every format to torchaudio (-> ffmpeg): output.py:119-187 (`torchaudio.save(format=...)`) and
torchaudio.load for speaker files.  Here wav, raw pcm and flac are native (no dependency); mp3 / opus / aac — lossy codecs
whose encoders are far outside this path — go to the same external back-ends the reference needs, if one is present at run
time: the `torchaudio` module, else an `ffmpeg` executable on PATH.  Without either the call fails with a message that says so
(never a silent substitution of another format)."""
from __future__ import annotations

import io
import shutil
import subprocess
import wave
from typing import Optional, Tuple, Union

import numpy as np

from . import flac

NATIVE_FORMATS = ("wav", "pcm", "flac")
EXTERNAL_FORMATS = ("mp3", "opus", "aac")
_FFMPEG_ARGS = {"mp3": ["-f", "mp3", "-codec:a", "libmp3lame"], "opus": ["-f", "ogg", "-codec:a", "libopus"],
                "aac": ["-f", "adts", "-codec:a", "aac"]}


def external_backend() -> Optional[str]:
    try:
        import torchaudio  # noqa: F401
        return "torchaudio"
    except Exception:
        pass
    return "ffmpeg" if shutil.which("ffmpeg") else None


def _to_int(pcm: np.ndarray, sample_width: int) -> np.ndarray:
    x = np.clip(np.asarray(pcm, dtype=np.float32).reshape(-1), -1.0, 1.0)
    if sample_width == 2:
        return (x * 32767.0).astype("<i2")
    if sample_width == 3:
        return np.rint(x.astype(np.float64) * 8388607.0).astype(np.int32)
    if sample_width == 4:
        return (x.astype(np.float64) * 2147483647.0).astype("<i4")
    if sample_width == 1:
        return (x * 127.0).astype(np.int8)
    raise ValueError("sample_width must be 1, 2, 3 or 4 bytes")


def wav_bytes(pcm: np.ndarray, sample_rate: int, sample_width: int = 2) -> bytes:
    if sample_width not in (2, 4):
        raise ValueError("wav: sample_width must be 2 or 4")
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(sample_width)
        w.setframerate(sample_rate)
        w.writeframes(_to_int(pcm, sample_width).tobytes())
    return buf.getvalue()


def encode(pcm: np.ndarray, sample_rate: int, fmt: str = "wav", sample_width: int = 2, bit_rate: int = 192,
           compression: int = 10) -> bytes:
    """float mono PCM in [-1, 1] -> bytes in `fmt` (the reference's format list: mp3, opus, aac, flac, wav, pcm)."""
    fmt = fmt.lower()
    if fmt in ("pcm", "raw"):
        x = _to_int(pcm, sample_width)
        if sample_width == 3:   # 24-bit samples travel as 3 little-endian bytes each (the int32 carrier has 4)
            return x.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
        return x.tobytes()
    if fmt == "wav":
        return wav_bytes(pcm, sample_rate, sample_width)
    if fmt == "flac":   # lossless: 16 bit for sample_width 2 (the reference's default), 24 bit above
        bps = 16 if sample_width <= 2 else 24
        return flac.encode(_to_int(pcm, 2 if bps == 16 else 3), sample_rate, bps)
    if fmt not in EXTERNAL_FORMATS:
        raise ValueError(f"Unsupported format: {fmt}. Supported formats are: mp3, opus, aac, flac, wav, pcm")
    backend = external_backend()
    if backend == "torchaudio":   # the reference's own call (output.py:150-176)
        import torch
        import torchaudio
        from torchaudio.io import CodecConfig
        t = torch.from_numpy(np.clip(np.asarray(pcm, np.float32).reshape(1, -1), -1.0, 1.0))
        buf = io.BytesIO()
        cfg = CodecConfig(compression_level=compression) if fmt == "opus" else CodecConfig(bit_rate=bit_rate)
        torchaudio.save(buf, t, sample_rate, format="adts" if fmt == "aac" else fmt, compression=cfg)
        return buf.getvalue()
    if backend == "ffmpeg":
        cmd = ["ffmpeg", "-hide_banner", "-loglevel", "error", "-f", "s16le", "-ar", str(sample_rate), "-ac", "1", "-i", "pipe:0",
               *_FFMPEG_ARGS[fmt], "-b:a", f"{bit_rate}k", "pipe:1"]
        r = subprocess.run(cmd, input=_to_int(pcm, 2).tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError(f"ffmpeg failed encoding {fmt}: {r.stderr.decode(errors='replace')[:500]}")
        return r.stdout
    raise RuntimeError(f"format '{fmt}' is a lossy codec that needs an external encoder (torchaudio or an ffmpeg executable, "
                       f"the reference's own dependency); neither is available here - wav, pcm and flac are built in")


def _read_wav(data: bytes) -> Tuple[np.ndarray, int]:
    fmt_tag = int.from_bytes(data[20:22], "little")
    with wave.open(io.BytesIO(data if fmt_tag == 1 else data[:20] + (1).to_bytes(2, "little") + data[22:]), "rb") as w:
        n, sw, ch, sr = w.getnframes(), w.getsampwidth(), w.getnchannels(), w.getframerate()
        raw = w.readframes(n)
    if fmt_tag == 3 and sw == 4:
        a = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    elif sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / 8388608.0
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported wav sample width {sw}")
    a = a.reshape(-1, ch).mean(axis=1) if ch > 1 else a
    return np.ascontiguousarray(a, dtype=np.float32), sr


def decode(src: Union[str, bytes, bytearray]) -> Tuple[np.ndarray, int]:
    """Audio file (path or bytes) -> (mono float32 samples, sample rate).  RIFF/WAVE and FLAC natively, anything else through
    the external back-end."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        data = bytes(src)
    else:
        with open(src, "rb") as f:
            data = f.read()
    if len(data) >= 44 and data[:4] == b"RIFF" and data[8:12] == b"WAVE":
        return _read_wav(data)
    if data[:4] == b"fLaC":
        x, sr, bps = flac.decode(data)
        return (x.astype(np.float32).mean(axis=1) / float(1 << (bps - 1))).astype(np.float32), sr
    backend = external_backend()
    if backend == "torchaudio":
        import torchaudio
        t, sr = torchaudio.load(io.BytesIO(data))
        return t.mean(dim=0).numpy().astype(np.float32), int(sr)
    if backend == "ffmpeg":
        r = subprocess.run(["ffmpeg", "-hide_banner", "-loglevel", "error", "-i", "pipe:0", "-f", "wav", "-acodec", "pcm_s16le", "pipe:1"],
                           input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise ValueError(f"ffmpeg could not decode the audio: {r.stderr.decode(errors='replace')[:500]}")
        return _read_wav(r.stdout)
    raise ValueError("audio must be RIFF/WAVE or FLAC (built in); other containers (mp3, ogg, m4a, ...) need torchaudio or an "
                     "ffmpeg executable, the reference's own dependency, and neither is available here")
