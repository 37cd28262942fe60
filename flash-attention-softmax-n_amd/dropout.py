"""Host mirror of the kernels' dropout bits (csrc/fasn_common.h: drop_seed / drop_row_base / drop_hash / drop_keep).

The keep/drop decision of attention weight (b, h, row i, key j) is the 16-bit field (j & 3) of a 64-bit hash of
(seed, offset, b*H + h, i, j >> 2); kept iff field >= thr, thr = round(65536 p) clipped to [1, 65535] (p honoured to 1.5e-5).
Used by the tests to build the explicit mask for the oracle, and by anyone who needs to reproduce a run's dropout pattern.
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
    x = rb ^ ((((key >> np.uint64(2)) * np.uint64(0xC2B2AE3D)) + seed_hi) & _M32)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    lo = (x * np.uint64(0x846CA68B)) & _M32
    hi = (lo * np.uint64(0x9E3779B1)) & _M32
    hi ^= hi >> np.uint64(15)
    lo ^= lo >> np.uint64(16)
    e = key & np.uint64(3)
    word = np.where(e >= np.uint64(2), hi, lo)
    field = (word >> (np.uint64(16) * (e & np.uint64(1)))) & np.uint64(0xFFFF)
    return (field >= np.uint64(thr)).reshape(B, H, L, S)
