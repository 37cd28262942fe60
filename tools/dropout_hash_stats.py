#!/usr/bin/env python3
"""Statistics of the dropout hash (csrc/fasn_common.h: drop_mix / drop_pair_word, stream definition 2; host mirror flash-attention-softmax-n_amd/dropout.py):
keep rate, correlation of the keep decisions of neighbouring keys / rows / heads / seeds / offsets, and the avalanche of every
input bit on every 16-bit field. Runs on the CPU (numpy)."""
import importlib.util
import os

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("dropout", os.path.join(_here, "..", "flash-attention-softmax-n_amd", "dropout.py"))
dropout = importlib.util.module_from_spec(spec)
spec.loader.exec_module(dropout)

M32, M24 = np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFF)


def mul24(a, b):
    return ((a & M24) * (np.uint64(b) & M24)) & M32


def rotl(x, r):
    r = np.uint64(r)
    return x if r == 0 else ((x << r) | (x >> (np.uint64(32) - r))) & M32


def mix(rb, sh, kq):
    x = ((rb + mul24(kq, 0x9E3779)) & M32) ^ sh
    y = (mul24(x, 0xC2B2AF) + rotl(mul24(rotl(x, 20), 0x85EBCB), 13)) & M32
    y ^= y >> np.uint64(15)
    return (y + rotl(y, 9)) & M32


def word(y, p):   # pair p = 4 q + 2 h + c of the 16-key group (stream definition 2)
    q, h, c = p >> 2, (p >> 1) & 1, p & 1
    return mul24(rotl(y, 16 * h + 4 * q + 5 * c), (0x2C1B3D, 0x297A2D, 0x1B56C5, 0x7ED55D)[2 * q + c])


def corr(a, b):
    return float(np.corrcoef(a.ravel().astype(np.float64), b.ravel().astype(np.float64))[0, 1])


L = S = 2048
sig = 1.0 / np.sqrt(L * S)
for p in (0.1, 0.5, 0.9):
    K = dropout.keep_mask(12345, 7, 1, 2, L, S, p)[0]
    print(f"p={p}: drop rate {1 - K.mean():.5f}; correlations in units of 1/sqrt(N): key+1 {corr(K[..., :-1], K[..., 1:]) / sig:+.1f} key+2 {corr(K[..., :-2], K[..., 2:]) / sig:+.1f} "
          f"key+4 {corr(K[..., :-4], K[..., 4:]) / sig:+.1f} key+8 {corr(K[..., :-8], K[..., 8:]) / sig:+.1f} key+16 {corr(K[..., :-16], K[..., 16:]) / sig:+.1f} "
          f"row+1 {corr(K[:, :-1], K[:, 1:]) / sig:+.1f} row+32 {corr(K[:, :-32], K[:, 32:]) / sig:+.1f} head+1 {corr(K[0], K[1]) / sig:+.1f} "
          f"seed+1 {corr(K, dropout.keep_mask(12346, 7, 1, 2, L, S, p)[0]) / sig:+.1f} offset+1 {corr(K, dropout.keep_mask(12345, 8, 1, 2, L, S, p)[0]) / sig:+.1f}")
    # every pair of positions inside a 16-key group (they share one 32-bit state): z-scores of the 120 pairs, pooled over rows and groups
    G = K[0].reshape(L, S // 16, 16).astype(np.float64)
    n = L * (S // 16)
    z = sorted(((abs(np.corrcoef(G[:, :, a].ravel(), G[:, :, b].ravel())[0, 1]) * np.sqrt(n), a, b) for a in range(16) for b in range(a + 1, 16)), reverse=True)
    print(f"      positions of a 16-key group, 120 pairs: largest |z| {[(round(float(v), 1), a, b) for v, a, b in z[:4]]} (the maximum of 120 standard normals is ~2.9)")
rng = np.random.default_rng(0)
N = 3000
rb, kq, sh = (rng.integers(0, 2 ** b, N, dtype=np.uint64) for b in (32, 20, 32))
pop = np.array([bin(i).count("1") for i in range(65536)])
y0 = mix(rb, sh, kq)
for pr in range(8):
    for half in (0, 1):
        fld = lambda w: (w >> np.uint64(16)) if half else (w & np.uint64(0xFFFF))
        w0 = fld(word(y0, pr))
        worst = (16.0, None)
        for nm, nb in (("row_base", 32), ("key_group", 20), ("seed_hi", 32)):
            for bit in range(nb):
                d = np.uint64(1 << bit)
                y1 = mix(rb ^ d if nm == "row_base" else rb, sh ^ d if nm == "seed_hi" else sh, kq ^ d if nm == "key_group" else kq)
                f = pop[(fld(word(y1, pr)) ^ w0).astype(np.int64)].mean()
                if f < worst[0]:
                    worst = (f, f"{nm} bit {bit}")
        print(f"pair {pr} {'odd ' if half else 'even'} key: flipping one input bit flips on average >= {worst[0]:.2f} of the 16 field bits (worst: {worst[1]})")
