"""CPU: audio containers of TTSOutput / reference audio (api/codecs.py, api/flac.py) — the reference's format list is
mp3, opus, aac, flac, wav, pcm (src/auralis/common/definitions/output.py:119-187)."""
import hashlib
import io
import struct
import wave

import numpy as np
import pytest

from auralis_amd import TTSOutput
from auralis_amd.api import codecs, flac

# RFC 9639 (FLAC), appendix D.1 "Decoding example 1": a complete two-channel, one-sample stream.  Known answer for the reader:
# both CRCs, the VERBATIM subframe with one wasted bit, and the STREAMINFO MD5 of the decoded audio.
RFC9639_EXAMPLE_1 = bytes.fromhex("664c614380000022100010000000" "0f00000f0ac442f000000001" "3e84b41807dc690307586a3dad1a2e0f"
                                  "fff869180000bf" "0358fd03128b" "aa9a")


def _speechlike(n=24000 * 2, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    x = 0.3 * np.sin(2 * np.pi * 180 * t) * np.exp(-1.5 * t) + 0.05 * np.sin(2 * np.pi * 2300 * t) + 0.01 * rng.standard_normal(n)
    x[n // 3: n // 3 + 6000] = 0.0          # digital silence between sentences -> CONSTANT blocks
    return x.astype(np.float32)


def test_flac_reader_known_answer_from_the_rfc():
    x, sr, bps = flac.decode(RFC9639_EXAMPLE_1)
    assert (sr, bps, x.shape) == (44100, 16, (1, 2))
    assert hashlib.md5(struct.pack("<hh", *x[0])).hexdigest() == "3e84b41807dc690307586a3dad1a2e0f"
    bad = bytearray(RFC9639_EXAMPLE_1)
    bad[-3] ^= 1                                                   # flip a bit of the last sample: CRC-16 must notice
    with pytest.raises(ValueError, match="CRC"):
        flac.decode(bytes(bad))


@pytest.mark.parametrize("n", [1, 5, 100, 4096, 4097, 10000])
def test_flac_lossless_roundtrip_lengths(n):
    z = (np.random.default_rng(n).standard_normal(n) * 3000).astype(np.int16)
    y, sr, bps = flac.decode(flac.encode(z, 22050, 16))
    assert sr == 22050 and bps == 16 and np.array_equal(y[:, 0], z)


def test_flac_block_kinds_and_long_streams():
    rng = np.random.default_rng(1)
    noise = rng.integers(-32768, 32767, 5000).astype(np.int16)     # incompressible -> VERBATIM
    assert np.array_equal(flac.decode(flac.encode(noise, 8000, 16))[0][:, 0], noise)
    const = np.full(9000, -7, np.int16)
    enc = flac.encode(const, 8000, 16)
    assert len(enc) < 100 and np.array_equal(flac.decode(enc)[0][:, 0], const)
    long = (rng.standard_normal(4096 * 130 + 17) * 500).astype(np.int16)   # > 127 frames: two-byte frame numbers
    assert np.array_equal(flac.decode(flac.encode(long, 24000, 16))[0][:, 0], long)
    with pytest.raises(ValueError):
        flac.decode(b"RIFFxxxx")


def test_ttsoutput_formats(tmp_path):
    x = _speechlike()
    out = TTSOutput(array=x)
    ref16 = (np.clip(x, -1, 1) * 32767.0).astype("<i2")
    assert out.to_bytes("pcm") == ref16.tobytes()
    with wave.open(io.BytesIO(out.to_bytes("wav")), "rb") as w:
        assert (w.getframerate(), w.getsampwidth(), w.getnchannels()) == (24000, 2, 1)
        assert w.readframes(w.getnframes()) == ref16.tobytes()
    fl = out.to_bytes("flac")
    y, sr, bps = flac.decode(fl)
    assert sr == 24000 and bps == 16 and np.array_equal(y[:, 0], ref16)          # lossless
    assert len(fl) < 0.75 * len(out.to_bytes("wav"))                              # and actually compressed
    y24, _, bps24 = flac.decode(out.to_bytes("flac", sample_width=4))
    assert bps24 == 24 and np.abs(y24[:, 0] / 8388607.0 - np.clip(x, -1, 1)).max() < 1e-6
    p = tmp_path / "a.flac"
    out.save(p)
    back = TTSOutput.from_file(p)
    assert back.sample_rate == 24000 and np.abs(back.array - x).max() < 1e-4
    with pytest.raises(ValueError, match="Unsupported format"):
        out.to_bytes("ogg-vorbis")


@pytest.mark.parametrize("fmt", ["mp3", "opus", "aac"])
def test_lossy_formats_use_the_references_backends_or_say_so(fmt, monkeypatch):
    """mp3 / opus / aac are produced by torchaudio or an ffmpeg executable (what the reference itself needs); without them the
    call names the missing back-end instead of substituting another format."""
    out = TTSOutput(array=_speechlike(4000))
    if codecs.external_backend() is None:
        with pytest.raises(RuntimeError, match="torchaudio or an ffmpeg"):
            out.to_bytes(fmt)
    else:
        assert len(out.to_bytes(fmt)) > 0
    # the ffmpeg leg, with the executable replaced by a recorder
    calls = {}

    class R:
        returncode, stdout, stderr = 0, b"ENCODED", b""

    def fake_run(cmd, input=None, **kw):
        calls["cmd"], calls["n"] = cmd, len(input)
        return R()

    monkeypatch.setattr(codecs, "external_backend", lambda: "ffmpeg")
    monkeypatch.setattr(codecs.subprocess, "run", fake_run)
    assert out.to_bytes(fmt) == b"ENCODED"
    assert calls["cmd"][0] == "ffmpeg" and "s16le" in calls["cmd"] and "24000" in calls["cmd"] and calls["n"] == 2 * 4000
    assert {"mp3": "libmp3lame", "opus": "libopus", "aac": "aac"}[fmt] in calls["cmd"]


def test_reference_audio_may_be_flac():
    """Speaker files: RIFF/WAVE and FLAC are decoded natively (conditioning.read_wav -> codecs.decode)."""
    from auralis_amd import conditioning as Cn
    x = _speechlike(22050)
    fl = flac.encode((x * 32767.0).astype(np.int16), 22050, 16)
    wv = codecs.wav_bytes(x, 22050)
    a, sr = Cn.read_wav(fl)
    b, sr2 = Cn.read_wav(wv)
    assert sr == sr2 == 22050 and a.shape == b.shape == (1, 22050)
    assert float((a - b).abs().max()) < 1e-4
    if codecs.external_backend() is None:
        with pytest.raises(ValueError, match="torchaudio or an"):
            Cn.read_wav(b"ID3\x04" + bytes(100))


def test_pcm_24_bit_is_three_little_endian_bytes_per_sample():
    """encode(fmt="pcm", sample_width=3) emitted the int32 carrier (4 bytes per sample)."""
    from auralis_amd.api import codecs
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1e-3], np.float32)
    raw = codecs.encode(x, 24000, "pcm", sample_width=3)
    assert len(raw) == 3 * len(x)
    back = [int.from_bytes(raw[3 * i: 3 * i + 3], "little", signed=True) for i in range(len(x))]
    assert back == [int(np.rint(float(v) * 8388607.0)) for v in x]
    assert len(codecs.encode(x, 24000, "pcm", sample_width=2)) == 2 * len(x)
    assert len(codecs.encode(x, 24000, "pcm", sample_width=4)) == 4 * len(x)
