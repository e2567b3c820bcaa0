"""FLAC reader for machines without ``soundfile`` / ``torchaudio`` (the reference globs ``*.flac`` next to ``*.wav``:
enhancement.py:39-43, and loads both through torchaudio, :61).  Pure Python + numpy, decode only, written from the format
specification (RFC 9639): STREAMINFO, frames with CRC-8 / CRC-16, CONSTANT / VERBATIM / FIXED / LPC subframes, Rice-coded residuals
(both parameter widths, escaped partitions), wasted bits, the three stereo decorrelation modes, 4 ... 32 bits per sample.
Every frame's checksums are verified and, when the stream carries one, the MD5 of the decoded samples.

    samples, sample_rate, bits = read_flac(path)      # int32 [frames, channels]
    frames, sample_rate, channels, bits = flac_info(path)   # from the header only
"""
from __future__ import annotations

import hashlib
from typing import Tuple

import numpy as np

__all__ = ["read_flac", "flac_info", "is_flac", "FlacError"]


class FlacError(ValueError):
    pass


def _crc_table(poly: int, bits: int):
    top, mask = 1 << (bits - 1), (1 << bits) - 1
    tab = []
    for b in range(256):
        c = b << (bits - 8)
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


_LZ8 = [8] + [7 - (b.bit_length() - 1) for b in range(1, 256)]      # leading zeros of a byte


class _Bits:
    """MSB-first bit reader over a bytes object."""

    def __init__(self, data: bytes, byte_pos: int = 0):
        self.d, self.p = data, byte_pos * 8

    def read(self, n: int) -> int:
        if n == 0:
            return 0
        p = self.p
        b0, b1 = p >> 3, (p + n + 7) >> 3
        if b1 > len(self.d):
            raise FlacError("truncated stream")
        v = int.from_bytes(self.d[b0:b1], "big")
        self.p = p + n
        return (v >> ((b1 << 3) - p - n)) & ((1 << n) - 1)

    def read_signed(self, n: int) -> int:
        v = self.read(n)
        return v - (1 << n) if n and (v >> (n - 1)) else v

    def unary(self) -> int:
        """Number of 0 bits before the next 1 bit (which is consumed)."""
        d, p = self.d, self.p
        i, off = p >> 3, p & 7
        if i >= len(d):
            raise FlacError("truncated stream")
        cur = (d[i] << off) & 0xFF
        if cur:
            z = _LZ8[cur]
            self.p = p + z + 1
            return z
        z = 8 - off
        i += 1
        n = len(d)
        while i < n and d[i] == 0:
            z += 8
            i += 1
        if i >= n:
            raise FlacError("truncated stream")
        lz = _LZ8[d[i]]
        self.p = (i << 3) + lz + 1
        return z + lz

    def align(self):
        self.p = (self.p + 7) & ~7

    @property
    def byte_pos(self) -> int:
        return self.p >> 3


def is_flac(path: str) -> bool:
    with open(path, "rb") as fh:
        head = fh.read(4)
    return head == b"fLaC" or head[:3] == b"ID3"


def _streaminfo(data: bytes):
    """-> (info dict, offset of the first frame)"""
    pos = 0
    if data[:3] == b"ID3":                                   # an ID3v2 tag in front of the stream
        if len(data) < 10:
            raise FlacError("truncated ID3 tag")
        size = ((data[6] & 0x7F) << 21) | ((data[7] & 0x7F) << 14) | ((data[8] & 0x7F) << 7) | (data[9] & 0x7F)
        pos = 10 + size
    if data[pos:pos + 4] != b"fLaC":
        raise FlacError("not a FLAC stream")
    pos += 4
    info = None
    while True:
        if pos + 4 > len(data):
            raise FlacError("truncated metadata")
        last, btype = data[pos] >> 7, data[pos] & 0x7F
        length = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + length]
        if btype == 0:
            if length < 34:
                raise FlacError("short STREAMINFO")
            v = int.from_bytes(body[10:18], "big")
            info = {"min_block": int.from_bytes(body[0:2], "big"), "max_block": int.from_bytes(body[2:4], "big"),
                    "sample_rate": v >> 44, "channels": ((v >> 41) & 7) + 1, "bits": ((v >> 36) & 31) + 1,
                    "total": v & ((1 << 36) - 1), "md5": bytes(body[18:34])}
        pos += 4 + length
        if last:
            break
    if info is None:
        raise FlacError("no STREAMINFO block")
    return info, pos


def flac_info(path: str) -> Tuple[int, int, int, int]:
    """(frames, sample rate, channels, bits per sample) from the header; a stream that does not state its length is decoded."""
    with open(path, "rb") as fh:
        data = fh.read()
    info, _ = _streaminfo(data)
    if info["total"] == 0:
        x, sr, bits = read_flac(path)
        return int(x.shape[0]), sr, int(x.shape[1]), bits
    return info["total"], info["sample_rate"], info["channels"], info["bits"]


_FIXED = ((), (1,), (2, -1), (3, -3, 1), (4, -6, 4, -1))


def _residual(br: _Bits, blocksize: int, order: int) -> np.ndarray:
    method = br.read(2)
    if method > 1:
        raise FlacError("reserved residual coding method")
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    po = br.read(4)
    nparts = 1 << po
    if (blocksize >> po) << po != blocksize and po:
        raise FlacError("partition order does not divide the block size")
    out = np.empty(blocksize - order, dtype=np.int64)
    o = 0
    for part in range(nparts):
        n = (blocksize >> po) - (order if part == 0 else 0)
        if n < 0:
            raise FlacError("partition shorter than the predictor order")
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for i in range(n):
                out[o + i] = br.read_signed(nb)
        else:
            read, unary = br.read, br.unary
            vals = [0] * n
            for i in range(n):
                q = unary()
                u = (q << k) | read(k) if k else q
                vals[i] = (u >> 1) ^ -(u & 1)
            out[o:o + n] = vals
        o += n
    return out


def _subframe(br: _Bits, blocksize: int, bps: int) -> np.ndarray:
    if br.read(1):
        raise FlacError("subframe padding bit set")
    stype = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
        if bps <= 0:
            raise FlacError("wasted bits exceed the sample size")
    if stype == 0:                                           # CONSTANT
        s = np.full(blocksize, br.read_signed(bps), dtype=np.int64)
    elif stype == 1:                                         # VERBATIM
        s = np.array([br.read_signed(bps) for _ in range(blocksize)], dtype=np.int64)
    elif 8 <= stype <= 12:                                   # FIXED predictor of order 0 .. 4: an order-fold running sum
        order = stype - 8
        if order > blocksize:
            raise FlacError("predictor order exceeds the block size")
        warm = [br.read_signed(bps) for _ in range(order)]
        res = _residual(br, blocksize, order)
        s = np.empty(blocksize, dtype=np.int64)
        s[:order] = warm
        if order == 0:
            s[:] = res
        else:
            # d_0 = the signal, d_j = the j-th difference; the residual is d_order.  Integrate order times, each time with the
            # initial value taken from the warm-up samples' own differences
            diffs = [np.array(warm, dtype=np.int64)]
            for j in range(1, order):
                diffs.append(np.diff(diffs[-1]))
            cur = res
            for j in range(order - 1, -1, -1):
                init = diffs[j][-1]                          # d_j at index order - 1
                cur = init + np.cumsum(cur)                   # d_j at indices order .. blocksize - 1
            s[order:] = cur
    elif stype >= 32:                                        # LPC
        order = (stype & 31) + 1
        if order > blocksize:
            raise FlacError("predictor order exceeds the block size")
        warm = [br.read_signed(bps) for _ in range(order)]
        prec = br.read(4) + 1
        if prec == 16:
            raise FlacError("invalid LPC precision")
        shift = br.read_signed(5)
        if shift < 0:
            raise FlacError("negative LPC shift")
        coefs = [br.read_signed(prec) for _ in range(order)]
        res = _residual(br, blocksize, order).tolist()
        hist = list(warm)
        rc = coefs[::-1]                                      # aligned with hist[-order:]
        for r in res:
            acc = 0
            tail = hist[-order:]
            for c, h in zip(rc, tail):
                acc += c * h
            hist.append(r + (acc >> shift))
        s = np.array(hist, dtype=np.int64)
    else:
        raise FlacError("reserved subframe type")
    if wasted:
        s <<= wasted
    return s


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192, 14: 16384, 15: 32768}
_RATE = {1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100, 10: 48000, 11: 96000}
_BITS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}


def read_flac(path: str, verify_md5: bool = True) -> Tuple[np.ndarray, int, int]:
    """Decode a FLAC file: (int32 samples [frames, channels], sample rate, bits per sample)."""
    with open(path, "rb") as fh:
        data = fh.read()
    info, pos = _streaminfo(data)
    nch, sbits = info["channels"], info["bits"]
    blocks = []
    n = len(data)
    while pos < n:
        if pos + 2 > n or data[pos] != 0xFF or (data[pos + 1] & 0xFE) != 0xF8:
            if info["total"] and sum(b.shape[0] for b in blocks) >= info["total"]:
                break                                        # trailing bytes behind the last frame (tags)
            raise FlacError(f"lost frame synchronisation at byte {pos}")
        start = pos
        br = _Bits(data, pos + 2)
        bs_code, sr_code = br.read(4), br.read(4)
        ch_code, bits_code = br.read(4), br.read(3)
        if br.read(1):
            raise FlacError("reserved frame header bit set")
        first = br.read(8)                                   # UTF-8 style coded frame / sample number: skip its continuation bytes
        if first & 0x80:
            extra = _LZ8[(~first) & 0xFF] - 1
            if extra < 1 or extra > 6:
                raise FlacError("invalid coded frame number")
            for _ in range(extra):
                if (br.read(8) & 0xC0) != 0x80:
                    raise FlacError("invalid coded frame number")
        if bs_code == 0:
            raise FlacError("reserved block size code")
        blocksize = br.read(8) + 1 if bs_code == 6 else br.read(16) + 1 if bs_code == 7 else _BLOCK[bs_code]
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        elif sr_code == 15:
            raise FlacError("invalid sample rate code")
        hdr_end = br.byte_pos
        if _crc8(data[start:hdr_end]) != br.read(8):
            raise FlacError(f"frame header checksum mismatch at byte {start}")
        if bits_code == 3:
            raise FlacError("reserved sample size code")
        bps = sbits if bits_code == 0 else _BITS[bits_code]
        if ch_code < 8:
            if ch_code + 1 != nch:
                raise FlacError("channel count changes inside the stream")
            chans = [_subframe(br, blocksize, bps) for _ in range(nch)]
        elif ch_code <= 10:
            if nch != 2:
                raise FlacError("stereo decorrelation in a stream that is not stereo")
            if ch_code == 8:                                 # left, side
                a = _subframe(br, blocksize, bps); s = _subframe(br, blocksize, bps + 1)
                chans = [a, a - s]
            elif ch_code == 9:                               # side, right
                s = _subframe(br, blocksize, bps + 1); b = _subframe(br, blocksize, bps)
                chans = [b + s, b]
            else:                                            # mid, side
                m = _subframe(br, blocksize, bps); s = _subframe(br, blocksize, bps + 1)
                m = (m << 1) | (s & 1)
                chans = [(m + s) >> 1, (m - s) >> 1]
        else:
            raise FlacError("reserved channel assignment")
        br.align()
        end = br.byte_pos
        if end + 2 > n or _crc16(data[start:end]) != int.from_bytes(data[end:end + 2], "big"):
            raise FlacError(f"frame checksum mismatch in the frame at byte {start}")
        pos = end + 2
        blocks.append(np.stack(chans, axis=1))
    x = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, nch), dtype=np.int64)
    if info["total"] and x.shape[0] != info["total"]:
        raise FlacError(f"decoded {x.shape[0]} samples, the header announces {info['total']}")
    if verify_md5 and info["md5"] != bytes(16):
        nbytes = (sbits + 7) // 8
        raw = x.astype("<i8").view(np.uint8).reshape(x.shape[0], nch, 8)[:, :, :nbytes]      # little-endian, sign-extended to whole bytes
        if hashlib.md5(np.ascontiguousarray(raw).tobytes()).digest() != info["md5"]:
            raise FlacError("MD5 of the decoded samples does not match the stream's")
    return x.astype(np.int32), info["sample_rate"], sbits
