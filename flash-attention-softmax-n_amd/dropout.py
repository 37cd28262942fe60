"""Host mirror of the kernels' dropout bits (csrc/fasn_common.h: drop_seed / drop_row_base / drop_mix / drop_pair_word / DropThr).

Stream definition 2 (round 6). The keep/drop decision of attention weight (b, h, row i, key j):
  y    = drop_mix(row_base(seed, offset, b*H + h, i), seed_hi, j >> 4)      one 32-bit state per (row, group of 16 keys), built from 24-bit
                                                                           multiplies, rotates, adds and xors (full-rate VALU operations only)
  p    = (j & 15) >> 1 = 4 q + 2 h + c                                      the key's pair inside the group
  word = mul24(rotl(y, 16 h + 4 q + 5 c), MUL[2 q + c])                     one more 24-bit multiply per PAIR of keys
  f    = (word >> 16 if j & 1 else word & 0xffff) ^ 0x8000                  the odd key's field is the high half of the product, the even key's the low half
  kept iff f >= thr, thr = round(65536 p) clipped to [1, 65535] (p honoured to 1.5e-5).
(Round 5 used one state per key quad and one multiply per key; the kernels of round 6 test a packed pair of weights with three packed
16-bit instructions.) Used by the tests to build the explicit mask for the oracle, and by anyone who needs to reproduce a run's dropout pattern.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def threshold(p: float) -> int:
    if p <= 0:
        return 0
    return int(min(65535, max(1, round(float(np.float32(p) * np.float32(65536.0))))))


def effective_p(p: float) -> float:
    return threshold(p) / 65536.0


def keep_mask(seed: int, offset: int, B: int, H: int, L: int, S: int, p: float) -> np.ndarray:
    """Boolean [B, H, L, S]: True where the attention weight is kept."""
    thr = threshold(p)
    if thr == 0:
        return np.ones((B, H, L, S), dtype=bool)
    seed, offset = seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF
    seed_lo = np.uint64((seed & 0xFFFFFFFF) ^ (((offset & 0xFFFFFFFF) * 0x9E3779B1) & 0xFFFFFFFF))
    seed_hi = np.uint64(((seed >> 32) + (offset >> 32)) & 0xFFFFFFFF)
    bh = np.arange(B * H, dtype=np.uint64).reshape(B * H, 1, 1)
    row = np.arange(L, dtype=np.uint64).reshape(1, L, 1)
    key = np.arange(S, dtype=np.uint64).reshape(1, 1, S)
    rb = ((seed_lo ^ ((bh * np.uint64(0x9E3779B1)) & _M32)) + row * np.uint64(0x85EBCA77)) & _M32
    m24 = np.uint64(0xFFFFFF)

    def mul24(u, c):   # v_mul_u32_u24: low 24 bits of both operands, low 32 bits of the product
        return ((u & m24) * np.uint64(c)) & _M32

    def rotl(u, r):
        r = np.uint64(r)
        return ((u << r) | (u >> (np.uint64(32) - r))) & _M32

    x = ((rb + mul24(key >> np.uint64(4), 0x9E3779)) & _M32) ^ seed_hi
    y = (mul24(x, 0xC2B2AF) + rotl(mul24(rotl(x, 20), 0x85EBCB), 13)) & _M32
    y ^= y >> np.uint64(15)
    y = (y + rotl(y, 9)) & _M32
    pr = ((key & np.uint64(15)) >> np.uint64(1)).astype(np.int64)            # pair p = 4 q + 2 h + c
    q, hh, c = pr >> 2, (pr >> 1) & 1, pr & 1
    rot = (16 * hh + 4 * q + 5 * c).astype(np.uint64)
    mul = np.choose(2 * q + c, [np.uint64(0x2C1B3D), np.uint64(0x297A2D), np.uint64(0x1B56C5), np.uint64(0x7ED55D)])
    win = np.where(rot == np.uint64(0), y, ((y << rot) | (y >> (np.uint64(32) - np.where(rot == 0, np.uint64(1), rot)))) & _M32)
    word = ((win & m24) * mul) & _M32
    field = np.where((key & np.uint64(1)) == np.uint64(1), word >> np.uint64(16), word & np.uint64(0xFFFF)) ^ np.uint64(0x8000)
    return (field >= np.uint64(thr)).reshape(B, H, L, S)
