"""ctypes wrapper of oracle/libattn_n_ref.so (attn_n_ref.c). TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libattn_n_ref.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "attn_n_ref.c")):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.attn_n_ref_f32.restype = ctypes.c_int
        _lib.softmax_n_ref_f32.restype = ctypes.c_int
        _lib.attn_n_ref_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def attention_n(q, k, v, n=0.0, scale=None, causal=False, mask=None, bias=None, return_lse=False):
    """q [B,H,L,E], k [B,H,S,E], v [B,H,S,Ev] numpy arrays (any float dtype -> fp32). mask: bool broadcastable to
    [B,H,L,S]; bias: float broadcastable to [B,H,L,S]. Returns fp32 out (and lse)."""
    lib = load()
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    B, H, L, E = q.shape
    S, Ev = k.shape[2], v.shape[3]
    scale = 1.0 / np.sqrt(E) if scale is None else scale
    out = np.empty((B, H, L, Ev), dtype=np.float32)
    lse = np.empty((B, H, L), dtype=np.float32)

    def strides(a, item):
        a4 = np.broadcast_to(a, (B, H, L, S))
        return a4, (ctypes.c_int64 * 4)(*[s // item for s in a4.strides])

    m8 = ms = b32 = bs = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        m8, ms = strides(mask, 1)
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        b32, bs = strides(bias, 4)
    rc = lib.attn_n_ref_f32(_p(q, ctypes.c_float), _p(k, ctypes.c_float), _p(v, ctypes.c_float), _p(out, ctypes.c_float),
                            _p(lse, ctypes.c_float), B, H, L, S, E, Ev, ctypes.c_double(scale), ctypes.c_double(n), int(causal),
                            _p(mask, ctypes.c_uint8) if mask is not None else None, ms,
                            _p(bias, ctypes.c_float) if bias is not None else None, bs)
    assert rc == 0
    return (out, lse) if return_lse else out


def softmax_n(x, n=0.0):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    rows = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    rc = lib.softmax_n_ref_f32(_p(x, ctypes.c_float), _p(y, ctypes.c_float), ctypes.c_int64(rows), ctypes.c_int64(x.shape[-1]),
                               ctypes.c_double(n))
    assert rc == 0
    return y


def threads():
    return load().attn_n_ref_threads()
