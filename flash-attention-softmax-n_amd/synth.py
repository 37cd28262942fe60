"""Counter-based synthetic input generator (bench.py, tests, golden fixtures).

Element i of a stream is a pure function of (seed, i): 64-bit integer hashing (splitmix64) -> sixteen 16-bit uniforms
-> their sum (Irwin-Hall, ~normal) -> one float64 affine map -> rounding to the target dtype. Only integer arithmetic
and one IEEE multiply are involved, so CPU and GPU, here and on the GPU box, produce identical bits (a checksum in the
golden fixtures proves it), and any slice of a big tensor can be generated without the rest.
Inputs follow the reference tests' distribution N(0, 0.5^2) (reference tests/common.py:18-20).
"""
import math

import torch

_M1 = 0xBF58476D1CE4E5B9 - (1 << 64)
_M2 = 0x94D049BB133111EB - (1 << 64)
_GOLD = 0x9E3779B97F4A7C15 - (1 << 64)
_SIGMA16 = math.sqrt(16.0 * (65536.0 ** 2 - 1.0) / 12.0)


def _lsr(z, s):
    return (z >> s) & ((1 << (64 - s)) - 1)


def _mix(z):
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _seed_offset(seed: int) -> int:
    v = (seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & ((1 << 64) - 1)
    return v - (1 << 64) if v >= (1 << 63) else v


def counter_normal(shape, seed: int, std: float = 0.5, dtype=torch.bfloat16, device="cpu", start: int = 0, chunk: int = 1 << 24) -> torch.Tensor:
    """Tensor of `shape` whose flattened element i is stream element start+i, ~N(0, std^2), rounded to `dtype`."""
    n = 1
    for d in shape:
        n *= d
    out = torch.empty(n, dtype=dtype, device=device)
    so = _seed_offset(seed)
    for c0 in range(0, n, chunk):
        c = min(chunk, n - c0)
        idx = torch.arange(start + c0, start + c0 + c, dtype=torch.int64, device=device)
        total = torch.zeros(c, dtype=torch.int64, device=device)
        base = idx * 4 + so
        for s in range(4):
            z = _mix(base + s)
            for f in range(4):
                total += _lsr(z, 16 * f) & 0xFFFF
        x = (total.to(torch.float64) - 16 * 32767.5) * (std / _SIGMA16)
        out[c0:c0 + c] = x.to(torch.float32).to(dtype)
    return out.reshape(shape)


def exact16(x: torch.Tensor) -> torch.Tensor:
    """float32 values exactly representable in BOTH bf16 and fp16 (|x| < 2^-14 flushed to 0)."""
    y = x.to(torch.float32).to(torch.bfloat16).to(torch.float32)
    return torch.where(y.abs() < 2.0 ** -14, torch.zeros_like(y), y)


def checksum(t: torch.Tensor) -> int:
    """Order-independent bit checksum of a 16/32-bit tensor (sum of its integer bit patterns)."""
    it = {2: torch.int16, 4: torch.int32}[t.element_size()]
    return int(t.contiguous().view(it).to(torch.int64).sum().item())


def alibi_slopes(H: int) -> torch.Tensor:
    """slope_h = 2^(-8(h+1)/H) built from exact IEEE operations only (sqrt, multiply) when H is a power of two."""
    assert H & (H - 1) == 0, "H must be a power of two"
    r = 2.0 ** -8  # 2^(-8/H) = (2^-8)^(1/H): take log2(H) square roots
    k = H
    while k > 1:
        r = math.sqrt(r)
        k //= 2
    out, cur = [], 1.0
    for _ in range(H):
        cur *= r
        out.append(cur)
    return torch.tensor(out, dtype=torch.float64)


def alibi_bias_rows(H: int, L: int, S: int, heads, rows, dtype) -> torch.Tensor:
    """bias[h, i, j] = -slope_h * |i + (S - L) - j| for the given heads/rows -> [len(heads), len(rows), S]."""
    sl = alibi_slopes(H)[torch.as_tensor(heads)].view(-1, 1, 1)
    i = torch.as_tensor(rows, dtype=torch.float64).view(1, -1, 1) + (S - L)
    j = torch.arange(S, dtype=torch.float64).view(1, 1, -1)
    return (-(sl * (i - j).abs())).to(torch.float32).to(dtype)


def alibi_bias(H: int, L: int, S: int, dtype, device="cpu") -> torch.Tensor:
    """Dense [H, L, S] ALiBi bias (the reference only accepts a dense tensor: flash_attn.py:62,100-103)."""
    sl = alibi_slopes(H).to(device).view(-1, 1, 1)
    i = torch.arange(L, dtype=torch.float64, device=device).view(1, -1, 1) + (S - L)
    j = torch.arange(S, dtype=torch.float64, device=device).view(1, 1, -1)
    return (-(sl * (i - j).abs())).to(torch.float32).to(dtype)


def keypad_mask(B: int, S: int, device="cpu") -> torch.Tensor:
    """[B,1,1,S] boolean key-padding mask with per-batch valid lengths S, 7S/8, 3S/4, S/2, then repeating."""
    fr = [1.0, 0.875, 0.75, 0.5]
    valid = torch.tensor([int(S * fr[b % 4]) for b in range(B)], device=device).view(B, 1, 1, 1)
    return torch.arange(S, device=device).view(1, 1, 1, S) < valid
