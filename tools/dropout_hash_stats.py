#!/usr/bin/env python3
"""Statistics of the dropout hash (csrc/fasn_common.h: drop_mix / drop_word; host mirror flash-attention-softmax-n_amd/dropout.py):
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


def word(y, e):
    return mul24(rotl(y, (0, 24, 12, 20)[e]), (0x2C1B3D, 0x297A2D, 0x1B56C5, 0x7ED55D)[e])


def corr(a, b):
    return float(np.corrcoef(a.ravel().astype(np.float64), b.ravel().astype(np.float64))[0, 1])


for p in (0.1, 0.5, 0.9):
    K = dropout.keep_mask(12345, 7, 1, 2, 2048, 2048, p)[0]
    print(f"p={p}: drop rate {1 - K.mean():.5f}; corr key+1 {corr(K[..., :-1], K[..., 1:]):+.5f} key+4 {corr(K[..., :-4], K[..., 4:]):+.5f} "
          f"row+1 {corr(K[:, :-1], K[:, 1:]):+.5f} row+32 {corr(K[:, :-32], K[:, 32:]):+.5f} head+1 {corr(K[0], K[1]):+.5f} "
          f"seed+1 {corr(K, dropout.keep_mask(12346, 7, 1, 2, 2048, 2048, p)[0]):+.5f} offset+1 {corr(K, dropout.keep_mask(12345, 8, 1, 2, 2048, 2048, p)[0]):+.5f}")
rng = np.random.default_rng(0)
N = 3000
rb, kq, sh = (rng.integers(0, 2 ** b, N, dtype=np.uint64) for b in (32, 22, 32))
pop = np.array([bin(i).count("1") for i in range(65536)])
y0 = mix(rb, sh, kq)
for e in range(4):
    w0 = word(y0, e) >> np.uint64(16)
    worst = (16.0, None)
    for nm, nb in (("row_base", 32), ("key_quad", 22), ("seed_hi", 32)):
        for bit in range(nb):
            d = np.uint64(1 << bit)
            y1 = mix(rb ^ d if nm == "row_base" else rb, sh ^ d if nm == "seed_hi" else sh, kq ^ d if nm == "key_quad" else kq)
            f = pop[((word(y1, e) >> np.uint64(16)) ^ w0).astype(np.int64)].mean()
            if f < worst[0]:
                worst = (f, f"{nm} bit {bit}")
    print(f"field {e}: flipping one input bit flips on average >= {worst[0]:.2f} of the 16 field bits (worst: {worst[1]})")
