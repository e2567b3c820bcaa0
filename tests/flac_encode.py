"""A small FLAC ENCODER for the tests of sgmse_amd/util/flac.py (test infrastructure: this image has no FLAC files and no tool that
writes them).  Written independently of the decoder from the format specification (RFC 9639); it can emit every construct the
decoder has a branch for: CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC subframes, both Rice parameter widths, partition orders,
escaped partitions, wasted bits, the three stereo decorrelation modes, explicit 8- / 16-bit block sizes and an ID3v2 prefix."""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence

import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value: int, bits: int):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (value & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def put_signed(self, value: int, bits: int):
        self.put(value & ((1 << bits) - 1), bits)

    def unary(self, zeros: int):
        while zeros >= 32:
            self.put(0, 32)
            zeros -= 32
        self.put(1, zeros + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.out)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= (1 << (5 * n + 1)):          # payload bits of an n-byte code: (7 - n) + 6 (n - 1) = 5 n + 1
        n += 1
    out = [((0xFF << (8 - n)) & 0xFF) | (v >> (6 * (n - 1)))]
    for i in range(n - 2, -1, -1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    return bytes(out)


_FIXED = ((), (1,), (2, -1), (3, -3, 1), (4, -6, 4, -1))


def _zigzag(r: int) -> int:
    return (r << 1) if r >= 0 else ((-r) << 1) - 1


def write_residual(bw: BitWriter, res: Sequence[int], blocksize: int, order: int, partition_order: int, rice2: bool, escape_first: bool):
    bw.put(1 if rice2 else 0, 2)
    bw.put(partition_order, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    o = 0
    for part in range(1 << partition_order):
        n = (blocksize >> partition_order) - (order if part == 0 else 0)
        chunk = [int(v) for v in res[o:o + n]]
        o += n
        if escape_first and part == 0:
            nb = max([1] + [(abs(v) if v >= 0 else abs(v + 1)).bit_length() + 1 for v in chunk])
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for v in chunk:
                bw.put_signed(v, nb)
            continue
        us = [_zigzag(v) for v in chunk]
        best_k, best = 0, None
        for k in range(0, esc):
            cost = sum((u >> k) + 1 + k for u in us)
            if best is None or cost < best:
                best_k, best = k, cost
        bw.put(best_k, pbits)
        for u in us:
            bw.unary(u >> best_k)
            bw.put(u & ((1 << best_k) - 1), best_k)
    assert o == len(res)


def write_subframe(bw: BitWriter, s: np.ndarray, bps: int, kind: str, order: int = 0, partition_order: int = 0, rice2: bool = False,
                   escape_first: bool = False, wasted: int = 0, lpc: Optional[dict] = None):
    s = [int(v) for v in s]
    n = len(s)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in s)
        s = [v >> wasted for v in s]
        bps -= wasted
    code = {"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 32 + (order - 1)}[kind]
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert all(v == s[0] for v in s)
        bw.put_signed(s[0], bps)
    elif kind == "verbatim":
        for v in s:
            bw.put_signed(v, bps)
    elif kind == "fixed":
        for v in s[:order]:
            bw.put_signed(v, bps)
        c = _FIXED[order]
        res = [s[i] - sum(c[j] * s[i - 1 - j] for j in range(order)) for i in range(order, n)]
        write_residual(bw, res, n, order, partition_order, rice2, escape_first)
    else:
        coefs, shift, prec = lpc["coefs"], lpc["shift"], lpc["precision"]
        assert len(coefs) == order
        for v in s[:order]:
            bw.put_signed(v, bps)
        bw.put(prec - 1, 4)
        bw.put_signed(shift, 5)
        for cf in coefs:
            bw.put_signed(cf, prec)
        res = [s[i] - (sum(coefs[j] * s[i - 1 - j] for j in range(order)) >> shift) for i in range(order, n)]
        write_residual(bw, res, n, order, partition_order, rice2, escape_first)


_BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
_BITS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode_flac(x: np.ndarray, sample_rate: int, bits: int, blocksize: int = 1024, plans: Optional[List[dict]] = None,
                id3_prefix: bool = False, with_md5: bool = True, announce_total: bool = True) -> bytes:
    """x: integer samples [frames, channels].  plans[i] describes frame i (cycled): {"stereo": "independent" | "left_side" |
    "right_side" | "mid_side", "sub": [subframe keyword dicts, one per coded channel]}."""
    x = np.asarray(x, dtype=np.int64)
    frames, nch = x.shape
    plans = plans or [{"stereo": "independent", "sub": [dict(kind="fixed", order=2)] * nch}]
    body = bytearray()
    fno = 0
    for start in range(0, frames, blocksize):
        blk = x[start:start + blocksize]
        n = blk.shape[0]
        plan = plans[fno % len(plans)]
        mode = plan["stereo"]
        hb = BitWriter()
        hb.put(0b11111111111110, 14); hb.put(0, 1); hb.put(0, 1)       # sync, reserved, fixed block size stream
        if n in _BLOCK_CODES:
            bs_code = _BLOCK_CODES[n]
        else:
            bs_code = 6 if n <= 256 else 7
        hb.put(bs_code, 4)
        hb.put(0, 4)                                                    # sample rate: from STREAMINFO
        hb.put({"independent": nch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[mode], 4)
        hb.put(_BITS_CODES.get(bits, 0) if plan.get("explicit_bits", True) else 0, 3)
        hb.put(0, 1)
        for b in utf8_number(fno):
            hb.put(b, 8)
        if bs_code == 6:
            hb.put(n - 1, 8)
        elif bs_code == 7:
            hb.put(n - 1, 16)
        header = hb.bytes()
        fb = BitWriter()
        for b in header:
            fb.put(b, 8)
        fb.put(crc8(header), 8)
        if mode == "independent":
            coded = [(blk[:, c], bits) for c in range(nch)]
        else:
            L, R = blk[:, 0], blk[:, 1]
            side = L - R
            if mode == "left_side":
                coded = [(L, bits), (side, bits + 1)]
            elif mode == "right_side":
                coded = [(side, bits + 1), (R, bits)]
            else:
                coded = [((L + R) >> 1, bits), (side, bits + 1)]
        for (sig, bps), sub in zip(coded, plan["sub"]):
            write_subframe(fb, sig, bps, **sub)
        fb.align()
        fr = fb.bytes()
        body += fr + crc16(fr).to_bytes(2, "big")
        fno += 1
    nbytes = (bits + 7) // 8
    raw = x.astype("<i8").view(np.uint8).reshape(frames, nch, 8)[:, :, :nbytes]
    md5 = hashlib.md5(np.ascontiguousarray(raw).tobytes()).digest() if with_md5 else bytes(16)
    si = BitWriter()
    si.put(blocksize, 16); si.put(blocksize, 16); si.put(0, 24); si.put(0, 24)
    si.put(sample_rate, 20); si.put(nch - 1, 3); si.put(bits - 1, 5); si.put(frames if announce_total else 0, 36)
    for b in md5:
        si.put(b, 8)
    sib = si.bytes()
    pad = bytes(10)                                                     # a second (padding) metadata block: the decoder has to walk the list
    out = b"fLaC" + bytes([0x00]) + len(sib).to_bytes(3, "big") + sib + bytes([0x80 | 1]) + len(pad).to_bytes(3, "big") + pad + bytes(body)
    if id3_prefix:
        tag = b"hello-id3"
        size = len(tag)
        out = b"ID3\x03\x00\x00" + bytes([(size >> 21) & 0x7F, (size >> 14) & 0x7F, (size >> 7) & 0x7F, size & 0x7F]) + tag + out
    return out
