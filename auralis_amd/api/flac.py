"""Native FLAC (free lossless audio codec) writer and reader for TTSOutput / reference audio.

The reference encodes through torchaudio -> ffmpeg (src/auralis/common/definitions/output.py:119-187) and decodes reference
audio through torchaudio.load (models/xttsv2/components/tts/layers/xtts/... load_audio); neither exists offline, and FLAC is the
one compressed format of the response_format list whose bitstream is simple enough to state here in full:

  stream   = "fLaC" + STREAMINFO metadata block + frames
  frame    = header (sync 0x3FFE, fixed block size, sample size, frame number, CRC-8) + one subframe per channel + padding + CRC-16
  subframe = CONSTANT | VERBATIM | FIXED (polynomial predictor of order 0..4) | LPC, residual in partitioned Rice codes

Writer: mono, 16 or 24 bit, 4096-sample blocks, per block the best fixed predictor (orders 0..4 by sum |residual|), Rice partition
order 0..4 with the exact-cost parameter per partition, CONSTANT blocks for digital silence; MD5 of the PCM in STREAMINFO.
Reader: every subframe type, 1-2 channels with left/side, right/side and mid/side decorrelation, 4..32 bits, both Rice
methods with escape partitions, variable block sizes; CRC-8 / CRC-16 / MD5 are verified."""
from __future__ import annotations

import hashlib
import struct
from typing import List, Tuple

import numpy as np

_BLOCK = 4096


def _crc_table(poly: int, bits: int) -> List[int]:
    top, mask, tab = 1 << (bits - 1), (1 << bits) - 1, []
    for i in range(256):
        c = i << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        tab.append(c)
    return tab


_CRC8 = _crc_table(0x07, 8)
_CRC16 = _crc_table(0x8005, 16)


def _crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def _crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ b]
    return c


def _utf8_number(n: int) -> bytes:
    """The frame number in the "UTF-8"-like coding of the frame header (up to 36 bits)."""
    if n < 0x80:
        return bytes([n])
    out, lead, limit = [], 0xC0, 0x20
    while True:
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        if n < limit:
            out.append(lead | n)
            return bytes(reversed(out))
        lead = (lead >> 1) | 0x80
        limit >>= 1


# ------------------------------------------------------------------------------------------------ writer
def _rice_cost(u: np.ndarray, k: int) -> int:
    return int(u.size) * (k + 1) + int((u >> k).sum())


def _best_partitioning(u: np.ndarray, order: int, block: int) -> Tuple[int, List[int], int]:
    """-> (partition order, Rice parameter per partition, total bits) for the zigzagged residual `u` (length block - order)."""
    best = None
    for po in range(0, 5):
        parts = 1 << po
        if block % parts or (block >> po) <= order:
            break
        n0 = (block >> po) - order
        bounds = [0, n0] + [n0 + (block >> po) * i for i in range(1, parts)]
        ks, bits = [], 2 + 4
        for a, b in zip(bounds[:-1], bounds[1:]):
            seg = u[a:b]
            mean = float(seg.mean()) if seg.size else 0.0
            k0 = min(14, max(0, int(np.floor(np.log2(mean + 1.0)))))
            cands = sorted({max(0, k0 - 1), k0, min(14, k0 + 1)})
            k = min(cands, key=lambda kk: _rice_cost(seg, kk))
            ks.append(k)
            bits += 4 + _rice_cost(seg, k)
        if best is None or bits < best[2]:
            best = (po, ks, bits)
    return best


def _put(bits: np.ndarray, pos: int, value: int, n: int) -> int:
    for i in range(n):
        bits[pos + i] = (value >> (n - 1 - i)) & 1
    return pos + n


def _encode_block(x: np.ndarray, frame_no: int, bps: int) -> bytes:
    n = int(x.size)
    x = x.astype(np.int64)
    head = bytearray()
    bs_code = 0b1100 if n == 4096 else 0b0111
    ss_code = {8: 0b001, 12: 0b010, 16: 0b100, 20: 0b101, 24: 0b110}[bps]
    head += bytes([0xFF, 0xF8, (bs_code << 4) | 0b0000, (0b0000 << 4) | (ss_code << 1)])
    head += _utf8_number(frame_no)
    if bs_code == 0b0111:
        head += struct.pack(">H", n - 1)
    head.append(_crc8(bytes(head)))

    if np.all(x == x[0]):   # CONSTANT (digital silence between sentences)
        body = np.zeros(8 + bps, np.uint8)
        p = _put(body, 0, 0b00000000, 8)
        _put(body, p, int(x[0]) & ((1 << bps) - 1), bps)
    else:
        res, order, e = [x], 0, x
        for _ in range(4):
            e = np.diff(e)
            res.append(e)
        costs = [int(np.abs(r[max(0, 4 - o):]).sum()) for o, r in enumerate(res)]   # compared over the same samples
        order = int(np.argmin(costs)) if n > 4 else 0
        e = res[order]   # residual of samples order..n-1
        u = np.where(e >= 0, e << 1, ((-e) << 1) - 1).astype(np.int64)
        po, ks, rbits = _best_partitioning(u, order, n)
        if rbits >= (n - order) * bps:   # incompressible: VERBATIM
            body = np.zeros(8 + n * bps, np.uint8)
            p = _put(body, 0, 0b00000010, 8)
            vals = (x & ((1 << bps) - 1)).astype(np.int64)
            for j in range(bps):
                body[p + j:p + n * bps:bps] = (vals >> (bps - 1 - j)) & 1
        else:
            body = np.zeros(8 + order * bps + rbits, np.uint8)
            p = _put(body, 0, (0b001000 | order) << 1, 8)
            for i in range(order):
                p = _put(body, p, int(x[i]) & ((1 << bps) - 1), bps)
            p = _put(body, p, 0b00, 2)
            p = _put(body, p, po, 4)
            parts = 1 << po
            n0 = (n >> po) - order
            bounds = [0, n0] + [n0 + (n >> po) * i for i in range(1, parts)]
            for (a, b), k in zip(zip(bounds[:-1], bounds[1:]), ks):
                p = _put(body, p, k, 4)
                seg = u[a:b]
                q = seg >> k
                lens = q + 1 + k
                starts = p + np.concatenate(([0], np.cumsum(lens)[:-1]))
                body[starts + q] = 1   # unary: q zeros, then a one
                rem = seg & ((1 << k) - 1)
                for j in range(k):
                    body[starts + q + 1 + j] = (rem >> (k - 1 - j)) & 1
                p += int(lens.sum())
            assert p == body.size
    frame = bytes(head) + np.packbits(body).tobytes()   # packbits pads the last byte with zeros = the frame's alignment padding
    return frame + struct.pack(">H", _crc16(frame))


def encode(pcm: np.ndarray, sample_rate: int, bits_per_sample: int = 16) -> bytes:
    """Mono integer PCM (int16 / int32 holding `bits_per_sample`-bit values) -> FLAC stream."""
    if bits_per_sample not in (16, 24):
        raise ValueError("FLAC writer: 16 or 24 bits per sample")
    x = np.asarray(pcm).reshape(-1).astype(np.int64)
    lim = 1 << (bits_per_sample - 1)
    if x.size and (x.min() < -lim or x.max() >= lim):
        raise ValueError("FLAC writer: sample out of range for the bit depth")
    frames = [_encode_block(x[i:i + _BLOCK], i // _BLOCK, bits_per_sample) for i in range(0, x.size, _BLOCK)]
    bytes_per = bits_per_sample // 8
    raw = x.astype("<i2").tobytes() if bytes_per == 2 else b"".join(int(v).to_bytes(3, "little", signed=True) for v in x)
    sizes = [len(f) for f in frames] or [0]
    blocks = [min(_BLOCK, x.size - i) for i in range(0, x.size, _BLOCK)] or [_BLOCK]
    si = struct.pack(">HH", max(16, min(blocks[:-1] or blocks)), max(16, max(blocks)))
    si += min(sizes).to_bytes(3, "big") + max(sizes).to_bytes(3, "big")
    si += ((sample_rate << 44) | (0 << 41) | ((bits_per_sample - 1) << 36) | x.size).to_bytes(8, "big")
    si += hashlib.md5(raw).digest()
    return b"fLaC" + bytes([0x80]) + len(si).to_bytes(3, "big") + si + b"".join(frames)


# ------------------------------------------------------------------------------------------------ reader
class _Bits:
    def __init__(self, data: bytes, pos: int):
        self.d, self.p = data, pos * 8

    def read(self, n: int) -> int:
        v = 0
        while n > 0:
            byte, off = self.d[self.p >> 3], self.p & 7
            take = min(n, 8 - off)
            v = (v << take) | ((byte >> (8 - off - take)) & ((1 << take) - 1))
            self.p += take
            n -= take
        return v

    def signed(self, n: int) -> int:
        v = self.read(n)
        return v - (1 << n) if v >> (n - 1) else v

    def unary(self) -> int:
        q = 0
        while True:
            byte, off = self.d[self.p >> 3], self.p & 7
            rest = byte & ((1 << (8 - off)) - 1)
            if rest:
                lead = (8 - off) - rest.bit_length()
                self.p += lead + 1
                return q + lead
            q += 8 - off
            self.p += 8 - off

    def align(self):
        self.p = (self.p + 7) & ~7


def _read_residual(br: _Bits, n: int, order: int) -> List[int]:
    method = br.read(2)
    if method > 1:
        raise ValueError("FLAC: reserved residual coding method")
    kbits, esc = (4, 15) if method == 0 else (5, 31)
    po = br.read(4)
    out = []
    for part in range(1 << po):
        cnt = (n >> po) - (order if part == 0 else 0)
        k = br.read(kbits)
        if k == esc:
            w = br.read(5)
            out.extend(br.signed(w) if w else 0 for _ in range(cnt))
        else:
            for _ in range(cnt):
                u = (br.unary() << k) | (br.read(k) if k else 0)
                out.append((u >> 1) ^ -(u & 1))
    return out


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _read_subframe(br: _Bits, n: int, bps: int) -> List[int]:
    if br.read(1):
        raise ValueError("FLAC: subframe padding bit set")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    if typ == 0:
        out = [br.signed(bps)] * n
    elif typ == 1:
        out = [br.signed(bps) for _ in range(n)]
    elif 8 <= typ <= 12 or typ >= 32:
        if typ >= 32:
            order = (typ & 31) + 1
            out = [br.signed(bps) for _ in range(order)]
            prec = br.read(4) + 1
            shift = br.signed(5)
            coef = [br.signed(prec) for _ in range(order)]
        else:
            order = typ - 8
            out = [br.signed(bps) for _ in range(order)]
            shift, coef = 0, _FIXED[order]
        res = _read_residual(br, n, order)
        for e in res:
            s = 0
            for i, c in enumerate(coef):
                s += c * out[-1 - i]
            out.append(e + (s >> shift))
    else:
        raise ValueError("FLAC: reserved subframe type")
    return [v << wasted for v in out] if wasted else out


def decode(data: bytes) -> Tuple[np.ndarray, int, int]:
    """FLAC stream -> (int32 samples [frames][channels], sample rate, bits per sample)."""
    if data[:4] != b"fLaC":
        raise ValueError("not a FLAC stream")
    pos, sr, ch, bps, total, md5 = 4, 0, 0, 0, 0, b""
    while True:
        last, typ = data[pos] >> 7, data[pos] & 0x7F
        ln = int.from_bytes(data[pos + 1:pos + 4], "big")
        if typ == 0:
            v = int.from_bytes(data[pos + 4 + 10:pos + 4 + 18], "big")
            sr, ch, bps, total = v >> 44, ((v >> 41) & 7) + 1, ((v >> 36) & 31) + 1, v & ((1 << 36) - 1)
            md5 = data[pos + 4 + 18:pos + 4 + 34]
        pos += 4 + ln
        if last:
            break
    if not sr:
        raise ValueError("FLAC: no STREAMINFO")
    chans: List[List[int]] = [[] for _ in range(ch)]
    while pos + 2 <= len(data):
        if data[pos] != 0xFF or (data[pos + 1] & 0xFE) != 0xF8:
            raise ValueError("FLAC: lost frame sync")
        start = pos
        br = _Bits(data, pos)
        br.read(16)
        bs_code, sr_code = br.read(4), br.read(4)
        ca, ss_code = br.read(4), br.read(3)
        br.read(1)
        lead = br.read(8)   # frame / sample number, UTF-8 style
        extra = 0
        while lead & (0x80 >> extra):
            extra += 1
        for _ in range(max(0, extra - 1)):
            br.read(8)
        if bs_code == 1:
            n = 192
        elif 2 <= bs_code <= 5:
            n = 576 << (bs_code - 2)
        elif bs_code == 6:
            n = br.read(8) + 1
        elif bs_code == 7:
            n = br.read(16) + 1
        else:
            n = 256 << (bs_code - 8)
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        fbps = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24}.get(ss_code)
        if fbps is None:
            raise ValueError("FLAC: reserved sample size")
        hdr_end = br.p >> 3
        if _crc8(data[start:hdr_end]) != data[hdr_end]:
            raise ValueError("FLAC: frame header CRC mismatch")
        br.read(8)
        if ca < 8:
            subs = [_read_subframe(br, n, fbps) for _ in range(ca + 1)]
        elif ca == 8:    # left / side
            l, s = _read_subframe(br, n, fbps), _read_subframe(br, n, fbps + 1)
            subs = [l, [a - b for a, b in zip(l, s)]]
        elif ca == 9:    # side / right
            s, r = _read_subframe(br, n, fbps + 1), _read_subframe(br, n, fbps)
            subs = [[a + b for a, b in zip(s, r)], r]
        elif ca == 10:   # mid / side
            m, s = _read_subframe(br, n, fbps), _read_subframe(br, n, fbps + 1)
            subs = [[((2 * a + (b & 1)) + b) >> 1 for a, b in zip(m, s)], [((2 * a + (b & 1)) - b) >> 1 for a, b in zip(m, s)]]
        else:
            raise ValueError("FLAC: reserved channel assignment")
        br.align()
        end = br.p >> 3
        if _crc16(data[start:end]) != int.from_bytes(data[end:end + 2], "big"):
            raise ValueError("FLAC: frame CRC mismatch")
        pos = end + 2
        for c in range(ch):
            chans[c].extend(subs[c])
    out = np.asarray(chans, dtype=np.int64).T
    if total and out.shape[0] != total:
        raise ValueError("FLAC: sample count differs from STREAMINFO")
    if md5 != bytes(16):
        nb = (bps + 7) // 8
        raw = out.astype("<i2").tobytes() if nb == 2 else b"".join(int(v).to_bytes(nb, "little", signed=True) for v in out.reshape(-1))
        if hashlib.md5(raw).digest() != md5:
            raise ValueError("FLAC: MD5 of the decoded audio differs from STREAMINFO")
    return out.astype(np.int32), sr, bps
