"""Host side of the hot path: the reference's attention API on top of libfasn (HIP, gfx950).

Mirrors, argument for argument:
  * flash_attention_n          — flash_attention_softmax_n/core/flash_attn.py:42-124
  * flash_attention_n_triton   — flash_attention_softmax_n/core/flash_attn_triton.py:339-357
  * slow_attention_n           — flash_attention_softmax_n/core/functional.py:32-93 (signature only; same kernel)
What the reference does by materialising tensors is passed to the kernel as scalars and strides:
  n zero-padded K/V rows (flash_attn.py:66-73)        -> `softmax_n` float (real-valued n allowed)
  q * scale/default (flash_attn.py:81-83)             -> `scale` float; the vector kernels multiply Q (or K) by scale*log2e once, in
                                                         registers, rounded to the operand type like the reference's pre-scaled q;
                                                         only the element-load kernels apply it in fp32 inside the exp2 argument
  dense [B,H,L,S] mask & bias (flash_attn.py:87-113)  -> broadcast strides (0 = broadcast), `causal` flag
PyTorch is used for device memory, streams and autograd only.
"""
from math import sqrt
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import BwdArgs, FwdArgs, View4

_SUPPORTED_D = (32, 64, 128, 256)
_DTYPES = {torch.float16: _lib.FASN_DTYPE_F16, torch.bfloat16: _lib.FASN_DTYPE_BF16, torch.float32: _lib.FASN_DTYPE_F32}


def _view4(t: Optional[Tensor]) -> View4:
    v = View4()
    if t is None:
        v.ptr = None
        return v
    v.ptr = t.data_ptr()
    for i in range(4):
        v.stride[i] = t.stride(i) if t.size(i) > 1 else 0
    v.stride[3] = 1 if t.size(3) == 1 else t.stride(3)
    return v


def _rows_ok(t: Tensor) -> bool:
    """16-byte row alignment rule of the C ABI (strides % 8 elements for 16-bit types, % 4 for fp32)."""
    if t.stride(-1) != 1 and t.size(-1) != 1:
        return False
    if t.data_ptr() % 16 != 0:
        return False
    q = 16 // t.element_size()
    return all(t.stride(i) % q == 0 or t.size(i) == 1 for i in range(t.dim() - 1))


def _canon(t: Tensor) -> Tensor:
    if t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.shape[-1] * t.element_size()) % 16 == 0:
        return t   # the common case, one C++ call instead of a stride walk
    return t if _rows_ok(t) else t.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a torch.cuda.Stream object (6 us of host time per launch)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_ptr(device) -> int:
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _current_device() -> int:
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


# ---- dropout stream (reference: torch's philox generator behind core/flash_attn.py:122 and core/functional.py:92) ----
# Eager calls: a call's (seed, offset) is a PURE FUNCTION of torch's CUDA generator of the device at call time - seed =
# initial_seed(), offset = its philox offset, which the call then advances by 4 as any torch random op would - and travels by
# value. torch.manual_seed reproduces a run, and whatever saves / restores the generator state (torch.utils.checkpoint,
# HF gradient_checkpointing) makes the recomputed forward draw the mask of the original one.
# Under HIP-graph capture nothing on the host runs at replay time, so captured calls draw from a {seed, offset} pair in DEVICE
# memory instead: one tiny captured kernel (fasn_rng_advance) hands the call its copy and advances the pair, so every replay
# resamples. The pair is created by the first eager dropout call after the generator was (re)seeded, with bit 62 of the offset
# set: the graph stream and the eager stream of one seed never share a position.
_RNG_STATE = {}
_GRAPH_STREAM_BIT = 1 << 62


def _next_rng_state(device):
    """(seed, offset) Python ints for one eager forward call, or - while the current stream is capturing - an int64[2] device
    tensor {seed, offset} filled by a captured kernel. The backward of the call gets the same object."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _RNG_STATE.get(idx)
    if not torch.cuda.is_current_stream_capturing():
        gen = torch.cuda.default_generators[idx]
        seed, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
        if st is None or st["seed"] != seed:   # first use, or the generator was re-seeded: (re)start the graph stream
            signed = seed - (1 << 64) if seed >= (1 << 63) else seed
            fresh = torch.tensor([signed, _GRAPH_STREAM_BIT + off], dtype=torch.int64)
            if st is None:
                _RNG_STATE[idx] = {"seed": seed, "state": fresh.to(device)}
            else:   # IN PLACE: a HIP graph captured earlier holds this tensor's address (fasn_rng_advance adds to it at every replay);
                    # a new allocation would leave that graph writing into memory the caching allocator hands out again
                st["state"].copy_(fresh)
                st["seed"] = seed
        return seed & 0xFFFFFFFFFFFFFFFF, off & 0xFFFFFFFFFFFFFFFF
    if st is None:
        raise RuntimeError("dropout inside a HIP graph capture: run one dropout call on this device before capturing "
                           "(the device-side random state is created on first use)")
    out = torch.empty(2, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().fasn_rng_advance(st["state"].data_ptr(), out.data_ptr(), 1, _stream_ptr(device)), "fasn_rng_advance")
    return out


def _fill_fwd(a: FwdArgs, q, k, v, o, lse, mask, bias, n, scale, causal, dropout_p=0.0, seed=0, offset=0, rng=None):
    a.q, a.k, a.v, a.o = _view4(q), _view4(k), _view4(v), _view4(o)
    a.lse = lse.data_ptr()
    a.mask = _view4(mask)
    a.bias = _view4(bias)
    if bias is not None:
        a.bias_dtype = _lib.FASN_BIAS_F32 if bias.dtype == torch.float32 else _lib.FASN_BIAS_SAME
    else:
        a.bias_dtype = _lib.FASN_BIAS_NONE
    a.dtype = _DTYPES[q.dtype]
    a.B, a.H, a.Sq, a.D = q.shape
    a.Sk = k.shape[2]
    a.Dv = v.shape[3]
    a.kv_group = q.shape[1] // k.shape[1] if k.shape[1] != q.shape[1] else 0   # grouped-query attention: fewer K/V heads
    a.scale = scale
    a.softmax_n = n
    a.causal = 1 if causal else 0
    a.dropout_p = dropout_p
    if isinstance(rng, tuple):     # eager call: this call's (seed, offset) by value
        a.seed, a.offset = rng
        a.rng_state = None
    else:                          # captured call: device {seed, offset} (or no dropout)
        a.seed, a.offset = seed, offset
        a.rng_state = None if rng is None else rng.data_ptr()


# ---- host path: one planned ABI call per pass, argument blocks cached per call signature -------------------------------------
# Filling a ctypes struct costs ~0.15 us per field and a fasn_*_workspace_bytes() round trip ~2 us; both depend only on shapes,
# strides, dtype and the scalar arguments. They are computed once per distinct signature (per thread: the cached block is
# mutated in place) and a call only rewrites the pointers (reference launch site: core/flash_attn_triton.py:278-291, which
# likewise passes raw pointers + strides per call).
import threading

_TLS = threading.local()


def _sig(t: Optional[Tensor]):
    return None if t is None else (t.shape, t.stride(), t.dtype)


def _cache():
    c = getattr(_TLS, "cache", None)
    if c is None:
        c = _TLS.cache = {}
    return c


class _Plan:
    __slots__ = ("fwd", "fwd_ws", "bwd", "bwd_ws", "path", "bwd_path")


_WARNED_SLOW = set()


def _plan_for(q, k, v, mask, bias, n, scale, causal, dropout_p):
    key = (_sig(q), _sig(k), _sig(v), _sig(mask), _sig(bias), n, scale, causal, dropout_p, q.device.index)
    c = _cache()
    pl = c.get(key)
    if pl is None:
        if len(c) > 256:
            c.clear()
        lib = _lib.load()
        B, H, L, _ = q.shape
        o = torch.empty((B, H, L, v.shape[3]), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
        pl = _Plan()
        pl.fwd = FwdArgs()
        _fill_fwd(pl.fwd, q, k, v, o, lse, mask, bias, n, scale, causal, dropout_p)
        pl.bwd = BwdArgs()
        _fill_fwd(pl.bwd.fwd, q, k, v, o, lse, mask, bias, n, scale, causal, dropout_p)
        dk = torch.empty((B, k.shape[1], k.shape[2], k.shape[3]), dtype=q.dtype, device=q.device)   # contiguous outputs: only the strides matter here
        pl.bwd.dout, pl.bwd.dq, pl.bwd.dk, pl.bwd.dv = _view4(o), _view4(o), _view4(dk), _view4(dk)
        pl.bwd.flags = 0
        with torch.cuda.device(q.device):
            pl.fwd_ws = lib.fasn_fwd_workspace_bytes(pl.fwd)     # > 0: short-query / long-key shape, keys split over workgroups
            pl.bwd_ws = lib.fasn_bwd_workspace_bytes(pl.bwd)     # 0 with libfasn.so (the ABI keeps the hook for plans that need scratch)
            pl.path = lib.fasn_fwd_path(pl.fwd)
            pl.bwd_path = None   # asked at the first backward (fasn_bwd_path wants the backward's real argument block)
        if pl.path == _lib.FASN_PATH_ELEMENT:   # same results, 3-5 x slower: say so once per kind of call instead of silently
            why = (_sig(mask), _sig(bias), dropout_p > 0.0)
            if why not in _WARNED_SLOW:
                _WARNED_SLOW.add(why)
                import warnings
                warnings.warn("flash_attention_n: this call takes the element-load kernels (3-5x slower than the vector path): "
                              "a mask / bias whose rows are not aligned vectors (unaligned or strided rows; an fp32 bias with 16-bit q at head dim 256, under dropout or with rows not 16-byte aligned), "
                              "scale <= 0 with a bias, or fp16 with a very large scale. "
                              "See fasn_fwd_path in include/fasn.h.", RuntimeWarning, stacklevel=4)
        c[key] = pl
    return pl


def _set_rng(a: FwdArgs, rng):
    if isinstance(rng, tuple):     # eager call: this call's (seed, offset) by value
        a.seed, a.offset = rng
        a.rng_state = None
    else:                          # captured call: device {seed, offset} (or no dropout)
        a.seed = a.offset = 0
        a.rng_state = None if rng is None else rng.data_ptr()


def _launch_fwd(q, k, v, mask, bias, n, scale, causal, dropout_p, rng):
    lib = _lib.load()
    B, H, L, _ = q.shape
    o = torch.empty((B, H, L, v.shape[3]), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    pl = _plan_for(q, k, v, mask, bias, n, scale, causal, dropout_p)
    a = pl.fwd
    a.q.ptr, a.k.ptr, a.v.ptr, a.o.ptr, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    if mask is not None:
        a.mask.ptr = mask.data_ptr()
    if bias is not None:
        a.bias.ptr = bias.data_ptr()
    if dropout_p > 0.0:
        _set_rng(a, rng)
    dev = q.device

    def launch():
        if pl.fwd_ws:
            ws = torch.empty(pl.fwd_ws, dtype=torch.uint8, device=dev)
            _lib.check(lib.fasn_fwd_ws(a, ws.data_ptr(), pl.fwd_ws, _stream_ptr(dev)), "fasn_fwd_ws")
        else:
            rc = lib.fasn_fwd(a, _stream_ptr(dev))
            if rc:
                _lib.check(rc, "fasn_fwd")

    if _current_device() == dev.index:   # the usual case: no device-guard object on the plain path
        launch()
    else:
        with torch.cuda.device(dev):
            launch()
    return o, lse


_POISON_SCRATCH = False   # test hook: fill the backward's scratch (delta) with NaN before the launch


class _FlashAttentionSoftmaxN(torch.autograd.Function):
    """autograd glue; same role as _FlashAttentionN (flash_attn_triton.py:241-336)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, bias, n: float, scale: float, causal: bool, dropout_p: float = 0.0, rng=None, bias_small: bool = False):
        # bias_small: `bias` is the caller's [1 or B, 1 or H, L, S] tensor, not yet expanded - its gradient then comes back in that
        # shape, summed over the broadcast batch / head dimensions inside the dbias kernel (no [B,H,L,S] buffer)
        bias_k = bias.expand(q.shape[0], q.shape[1], q.shape[2], k.shape[2]) if bias_small else bias
        o, lse = _launch_fwd(q, k, v, mask, bias_k, n, scale, causal, dropout_p, rng)
        ctx.save_for_backward(q, k, v, o, lse, mask, bias)
        ctx.n, ctx.scale, ctx.causal, ctx.dropout_p, ctx.rng, ctx.bias_small = n, scale, causal, dropout_p, rng, bias_small
        return o

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, o, lse, mask, bias = ctx.saved_tensors
        dout = _canon(dout)
        B, H, L, D = q.shape
        S = k.shape[2]
        dev = q.device
        bias_small = bias if ctx.bias_small else None
        if ctx.bias_small:
            bias = bias.expand(B, H, L, S)
        dq = torch.empty((B, H, L, D), dtype=q.dtype, device=dev)
        Hkv = k.shape[1]   # grouped-query attention: each dK/dV workgroup sums the query heads of its K/V head in registers
        dk = torch.empty((B, Hkv, S, D), dtype=q.dtype, device=dev)
        dv = torch.empty((B, Hkv, S, v.shape[3]), dtype=q.dtype, device=dev)
        delta = torch.empty((B, H, L), dtype=torch.float32, device=dev)
        if _POISON_SCRATCH:   # tests: every kernel that reads delta must find it written (by fasn_bwd_delta or the dQ kernel's prologue)
            delta.fill_(float("nan"))
        pl = _plan_for(q, k, v, mask, bias, ctx.n, ctx.scale, ctx.causal, ctx.dropout_p)
        a = pl.bwd
        f = a.fwd
        f.q.ptr, f.k.ptr, f.v.ptr, f.o.ptr, f.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
        if mask is not None:
            f.mask.ptr = mask.data_ptr()
        if bias is not None:
            f.bias.ptr = bias.data_ptr()
        if ctx.dropout_p > 0.0:
            _set_rng(f, ctx.rng)
        if dout.stride() == o.stride():
            a.dout.ptr = dout.data_ptr()
        else:
            a.dout = _view4(dout)
        a.dq.ptr, a.dk.ptr, a.dv.ptr, a.delta = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
        a.workspace, a.workspace_bytes = None, 0
        # gradient of attn_bias: the kernel writes dS densely; autograd sums it over the dimensions the caller's bias broadcasts
        # (the expand / unsqueeze / dtype cast in _attention are ordinary autograd ops)
        dbias = None
        a.dbias_dtype = 0
        if bias is not None and ctx.needs_input_grad[4]:
            if bias_small is not None:   # reduced form: the kernel sums over the bias's broadcast batch / head dimensions
                dbias = torch.empty(bias_small.shape, dtype=bias_small.dtype, device=dev)
                a.dbias = _view4(dbias)   # (size-1 dimensions get stride 0 = "reduce over this dimension")
                a.dbias_dtype = _lib.FASN_BIAS_F32 if dbias.dtype == torch.float32 else _lib.FASN_BIAS_SAME
            else:
                dbias = torch.zeros((B, H, L, S), dtype=q.dtype, device=dev)   # tiles the causal walk skips are never written
                a.dbias = _view4(dbias)
        else:
            a.dbias.ptr = None
        if pl.bwd_path is None:   # once per cached call signature
            pl.bwd_path = lib.fasn_bwd_path(a)
            if pl.bwd_path == _lib.FASN_PATH_ELEMENT and pl.path != _lib.FASN_PATH_ELEMENT:
                why = ("bwd", _sig(mask), _sig(bias), D)
                if why not in _WARNED_SLOW:
                    _WARNED_SLOW.add(why)
                    import warnings
                    warnings.warn("flash_attention_n backward: dQ / dK / dV of this call take the element-load kernels (3-5x slower) although its forward "
                                  "is on the vector path. See fasn_bwd_path in include/fasn.h.", RuntimeWarning, stacklevel=2)

        def launch():
            if pl.bwd_ws:
                ws = torch.empty(pl.bwd_ws, dtype=torch.uint8, device=dev)
                a.workspace, a.workspace_bytes = ws.data_ptr(), pl.bwd_ws
            rc = lib.fasn_bwd(a, _stream_ptr(dev))
            if rc:
                _lib.check(rc, "fasn_bwd")

        try:
            if _current_device() == dev.index:   # the usual case: no device-guard object (about 10 us of host time per step)
                launch()
            else:
                with torch.cuda.device(dev):
                    launch()
        finally:   # the block is cached per call signature: whatever this call changed beyond pointers goes back, also when the launch raised
            if dout.stride() != o.stride():
                a.dout = _view4(o)
            a.dbias.ptr = None
        if dbias is not None and dbias.dtype != bias.dtype:
            dbias = dbias.to(bias.dtype)
        return dq, dk, dv, None, dbias, None, None, None, None, None, None


def _pad_feature(t: Tensor, d: int) -> Tensor:
    return t if t.shape[-1] == d else torch.nn.functional.pad(t, (0, d - t.shape[-1]))


def _prepare(query, key, value, n, scale, dropout_p, mask, bias):
    """Argument normalisation shared by flash_attention_n and kernel_path: validation, K/V head expansion, sign of the scale, feature
    padding, row alignment, mask / bias broadcasting. Returns (q, k, v, mask, bias, n, scale, dropout_p, dpad, Ev, bias_small)."""
    if not query.is_cuda:
        raise RuntimeError("flash_attention_softmax_n_amd runs on MI355X device tensors only; got a CPU tensor "
                           "(there is deliberately no CPU fallback)")
    if query.dtype not in _DTYPES:
        raise NotImplementedError(f"dtype {query.dtype}: supported are torch.float16, torch.bfloat16 and torch.float32")
    if key.dtype != query.dtype or value.dtype != query.dtype:
        raise TypeError("query, key and value must share one dtype")
    dropout_p = float(dropout_p or 0.0)
    if not 0.0 <= dropout_p < 1.0:
        raise ValueError("dropout_p must be in [0, 1)")
    if query.dim() != 4:
        raise ValueError("query must be [B, H, L, E]")
    n = 0.0 if n is None else float(n)
    if n < 0:
        raise ValueError("softmax_n_param must be >= 0")
    B, H, L, E = query.shape

    # 3-D key/value [B, S, E] = one K/V shared by all heads: a stride-0 head dimension, no copy
    if key.dim() == 3:
        key = key.unsqueeze(1).expand(B, H, key.shape[1], key.shape[2])
    if value.dim() == 3:
        value = value.unsqueeze(1).expand(B, H, value.shape[1], value.shape[2])
    S, Ev = key.shape[2], value.shape[3]
    if key.shape[3] != E or value.shape[2] != S:
        raise ValueError("key must be [B,H,S,E] and value [B,H,S,Ev]")
    Hkv = key.shape[1]
    if Hkv not in (1, H) and (H % Hkv != 0 or value.shape[1] != Hkv):
        raise ValueError(f"key/value have {Hkv} heads: must be 1, {H}, or a divisor of {H} (grouped-query attention)")
    if Hkv == H:
        pass
    elif Hkv == 1:
        key = key.expand(B, H, S, E)
        value = value.expand(B, H, S, Ev)
    else:   # grouped-query attention: query head h reads K/V head h // (H // Hkv), through the head stride (no copy)
        key = key.expand(B, Hkv, S, E)
        value = value.expand(B, Hkv, S, Ev)

    scale = (1.0 / sqrt(E)) if scale is None else float(scale)
    if scale < 0:  # exp2 folding assumes scale >= 0: move the sign into q
        query, scale = -query, -scale

    # feature dims: kernels exist for D == Dv in {32, 64, 128} and (fp16 / bf16) 256; anything else is zero-padded (exact)
    dmax = 128 if query.dtype == torch.float32 else 256
    dpad = next((d for d in _SUPPORTED_D if max(E, Ev) <= d <= dmax), None)
    if dpad is None:
        raise NotImplementedError(f"head dim {max(E, Ev)} > {dmax} is not supported for {query.dtype}")
    if E == dpad and Ev == dpad:
        q, k, v = _canon(query), _canon(key), _canon(value)
    else:
        q, k, v = (_canon(_pad_feature(t, dpad)) for t in (query, key, value))

    if mask is not None:
        if mask.dim() != 4:
            raise AssertionError("attn_mask must be 4-D and broadcastable to [B, H, L, S]")  # flash_attn.py:88
        if mask.dtype != torch.bool:
            raise TypeError("attn_mask must be boolean (True = attend); pass additive masks as attn_bias")
        # bool -> uint8 reinterpretation keeps the (possibly stride-0) broadcast strides: no dense copy
        mask = mask.expand(B, H, L, S).view(torch.uint8)
    if bias is not None:
        if bias.dim() == 3:
            bias = bias.unsqueeze(0)  # [H,L,S] -> [1,H,L,S]  (flash_attn.py:101-102)
        if bias.dim() != 4:
            raise ValueError("attn_bias must be [H, L, S] or broadcastable to [B, H, L, S]")
        if bias.dtype not in (query.dtype, torch.float32):
            bias = bias.to(query.dtype)
        # a bias that needs a gradient and broadcasts over batch and / or heads only ([H,L,S], [1,H,L,S], [B,1,L,S], [1,1,L,S]) keeps
        # its own shape: fasn_bwd then returns the gradient already summed over those dimensions (csrc/fasn_bwd_dbias.h).
        # (L == 1 stays on the dense path: a one-row bias is also a ROW broadcast, whose stride 0 the reduced form rejects)
        # Round 5: a bias that ALSO broadcasts over rows ([1,H,1,S], [B,1,1,S], [1,1,1,S] - per-key biases) is expanded over the rows as a
        # view (row stride 0: the kernels read one row) and takes the same reduced form: the kernel sums over batch / heads into a
        # [1 or B, 1 or H, L, S] buffer and the expand's own backward sums that over the rows - 1/B or 1/H of the dense [B,H,L,S] dS buffer
        # round 4 wrote for these shapes. (The sum over rows inside the kernel would be a column sum of dS in the dK/dV kernels: not built.)
        if (bias.requires_grad and torch.is_grad_enabled() and L > 1 and bias.shape[2] == 1 and bias.shape[3] == S
                and bias.shape[0] in (1, B) and bias.shape[1] in (1, H)):
            bias = bias.expand(bias.shape[0], bias.shape[1], L, S)
        bias_small = (bias.requires_grad and torch.is_grad_enabled() and dropout_p == 0.0 and query.dtype != torch.float32
                      and L > 1 and bias.shape[2] == L and bias.shape[3] == S and bias.shape[0] in (1, B) and bias.shape[1] in (1, H)
                      and ((bias.shape[0] == 1 and B > 1) or (bias.shape[1] == 1 and H > 1)) and bias.stride(3) == 1)
        if not bias_small:
            bias = bias.expand(B, H, L, S)
    if bias is None:
        bias_small = False
    return q, k, v, mask, bias, n, scale, dropout_p, dpad, Ev, bias_small


def _regroup_decode(q, k, mask, bias, dropout_p):
    """Grouped-query decode (one query position, fewer K/V heads than query heads, no dropout): the G query heads of a group become G
    query ROWS of one problem per K/V head, so each K/V head is streamed once instead of G times. With one query position a row sees
    every key, so causal needs no flag, and a [B,H,1,S] mask / bias is a per-row mask / bias of the regrouped problem. All of it is
    views. Returns (q, mask, bias) of the regrouped problem, or None when the call keeps the per-head launch."""
    B, H, L, dpad = q.shape
    Hkv, S = k.shape[1], k.shape[2]
    if not (L == 1 and Hkv != H and dropout_p == 0.0):
        return None
    G = H // Hkv
    try:
        mask_g = None if mask is None else mask.view(B, Hkv, G, S)
        bias_g = None if bias is None else bias.expand(B, H, L, S).view(B, Hkv, G, S)
    except RuntimeError:      # strides that cannot be regrouped without a copy: keep the per-head launch
        return None
    return q.view(B, Hkv, G, dpad), mask_g, bias_g


def _attention(query, key, value, n, scale, dropout_p, mask, bias, is_causal) -> Tensor:
    q, k, v, mask, bias, n, scale, dropout_p, dpad, Ev, bias_small = _prepare(query, key, value, n, scale, dropout_p, mask, bias)
    B, H, L, _ = q.shape
    S = k.shape[2]
    # dropout: this call's (seed, offset) (see _next_rng_state); the kernels derive every keep/drop bit from
    # (seed, offset, b, h, row, key) - dropout.py is the host mirror
    rng = _next_rng_state(query.device) if dropout_p > 0.0 else None
    _TLS.last_rng_state = rng   # per thread: what last_dropout_state() / last_rng_state() report
    rg = _regroup_decode(q, k, mask, bias, dropout_p)
    if rg is not None:
        q_g, mask_g, bias_g = rg
        out = _FlashAttentionSoftmaxN.apply(q_g, k, v, mask_g, bias_g, n, scale, False, 0.0, None)
        out = out.view(B, H, 1, dpad)
        return out if Ev == dpad else out[..., :Ev]
    if not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad or (bias is not None and bias.requires_grad))):
        out = _launch_fwd(q, k, v, mask, bias, n, scale, bool(is_causal), dropout_p, rng)[0]   # nothing to differentiate: no autograd node
    else:
        out = _FlashAttentionSoftmaxN.apply(q, k, v, mask, bias, n, scale, bool(is_causal), dropout_p, rng, bias_small)
    return out if Ev == dpad else out[..., :Ev]


def kernel_path(query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None, attn_bias: Optional[Tensor] = None,
                is_causal: bool = False, dropout_p: float = 0.0, scale: Optional[float] = None) -> str:
    """Name of the kernel family a flash_attention_n call with these arguments is routed to (fasn_fwd_path): "plain", "key-padding",
    "vector mask/bias", "vector bias + key-padding", "element-load (slow)" or "fp32". Launches no attention kernel and leaves the
    per-thread plan cache alone: it runs the argument normalisation of the real call (which may issue its small device ops - a feature
    pad, the copy of a misaligned operand, a bias cast) and asks fasn_fwd_path about a throw-away argument block built from the
    canonical tensors, including the regrouping of a grouped-query decode call. fasn_fwd_path looks at layouts and modes only."""
    q, k, v, mask, bias, n, sc, dp, _, _, bias_small = _prepare(query, key, value, 1.0, scale, dropout_p, attn_mask, attn_bias)
    if bias_small:
        bias = bias.expand(q.shape[0], q.shape[1], q.shape[2], k.shape[2])
    causal = bool(is_causal)
    rg = _regroup_decode(q, k, mask, bias, dp)
    if rg is not None:
        (q, mask, bias), causal = rg, False
    a = FwdArgs()
    _fill_fwd(a, q, k, v, q, q, mask, bias, n, sc, causal, dp)   # (o / lse: any device pointer, the query stands in - nothing is launched)
    with torch.cuda.device(q.device):
        return _lib.FASN_PATH_NAMES[_lib.load().fasn_fwd_path(a)]


def last_rng_state():
    """What the calling thread's most recent flash_attention_n call used as its dropout state: None (no dropout), a (seed, offset)
    tuple (eager calls: passed by value) or the device tensor {seed, offset} a HIP-graph capture reads at every replay."""
    return getattr(_TLS, "last_rng_state", None)


def last_dropout_state():
    """(seed, offset) of the most recent dropout call as Python ints (one device read), for dropout.keep_mask; None if that call
    had dropout_p == 0. Test / reproduction helper."""
    t = getattr(_TLS, "last_rng_state", None)
    if t is None:
        return None
    if isinstance(t, tuple):
        return t
    s, o = (int(x) for x in t.cpu().tolist())
    return s & 0xFFFFFFFFFFFFFFFF, o & 0xFFFFFFFFFFFFFFFF


def flash_attention_n(
        query: Tensor,
        key: Tensor,
        value: Tensor,
        softmax_n_param: Optional[float] = None,
        scale: Optional[float] = None,
        dropout_p: float = 0.,
        attn_mask: Optional[Tensor] = None,
        attn_bias: Optional[Tensor] = None,
        is_causal: bool = False
) -> Tensor:
    """Fused attention with softmax_n on MI355X; drop-in for flash_attention_softmax_n.flash_attention_n.

    :param query: [B, H, L, E] fp16 / bf16 device tensor, E <= 256 (fp32 also accepted: exact-fp32 kernels, ~1/16 of the bf16
                  rate, E <= 128).
    :param key: [B, H, S, E] (or [B, S, E], shared by all heads).
    :param value: [B, H, S, Ev].
    :param softmax_n_param: n >= 0; real values allowed (the reference's SDPA path takes integers only).
    :param scale: multiplies q.k^T; default 1/sqrt(E).
    :param dropout_p: attention-weight dropout, realised in 1/65536 steps (dropout.effective_p); the mask is a pure function of
                      the call's (seed, offset) - drawn from a per-device stream seeded by torch's CUDA generator, advanced on the
                      device so that HIP-graph replays resample - and regenerated in backward.
    :param attn_mask: bool, 4-D, broadcastable to [B, H, L, S]; True = attend.
    :param attn_bias: additive bias [H, L, S] or broadcastable to [B, H, L, S] (e.g. ALiBi). Differentiated like the reference's
                      additive mask (core/flash_attn.py:100-113): a bias that requires grad gets dS summed over the dimensions it
                      broadcasts - inside the kernel for batch / head broadcasts ([H,L,S], [B,1,L,S], ...: no [B,H,L,S] buffer).
    :param is_causal: bottom-right aligned causal mask (key j visible to row i iff j <= i + S - L).
    :return: [B, H, L, Ev] in query's dtype.
    Rows with no visible key and n == 0 return 0 (the reference returns NaN there).
    """
    return _attention(query, key, value, softmax_n_param, scale, dropout_p, attn_mask, attn_bias, is_causal)


def flash_attention_n_triton(query: Tensor, key: Tensor, value: Tensor, is_causal: bool = False,
                             scale: Optional[float] = None, softmax_n_param: Optional[float] = None) -> Tensor:
    """Signature of flash_attn_triton.py:339-345; served by the same HIP kernel (bf16 too, any L/S)."""
    return _attention(query, key, value, softmax_n_param, scale, 0.0, None, None, is_causal)


def slow_attention_n(query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None, dropout_p: float = 0.0,
                     is_causal: bool = False, scale: Optional[float] = None, softmax_n_param: Optional[float] = None,
                     softmax_dtype=None, train: bool = True) -> Tensor:
    """Signature of functional.py:32-42 on the fused kernel. Accepts (N, ..., L, E) with 3-D or 4-D inputs.
    Float masks are additive (broadcast over leading dims), boolean masks hide keys (the reference's
    eager version silently ignores boolean masks, functional.py:85-86).
    softmax_dtype: the reference casts the softmax_n weights to it (default: query's dtype, functional.py:72-73,91) and multiplies them
    with `value` next (:93) - so the only values that do not raise there are None and value's own dtype, and for those the weights reach
    the P.V contraction rounded to value's dtype. That is what the kernel does as well (P is packed to the operand dtype for the MFMA; the
    scores, the running max and the row sum stay in fp32 registers, where the reference keeps them in query's dtype). Any other
    softmax_dtype raises the RuntimeError torch's matmul raises in the reference."""
    if is_causal and attn_mask is not None:
        raise AssertionError("attn_mask and is_causal are mutually exclusive")  # functional.py:79
    if softmax_dtype is not None and softmax_dtype != value.dtype:
        raise RuntimeError(f"expected m1 and m2 to have the same dtype, but got: {softmax_dtype} != {value.dtype} "
                           "(softmax_dtype must be None or value's dtype: the weights are multiplied with value next, functional.py:91-93)")
    squeeze = query.dim() == 3
    if squeeze:
        query, key, value = query.unsqueeze(1), key.unsqueeze(1), value.unsqueeze(1)
    mask = bias = None
    if attn_mask is not None:
        am = attn_mask
        while am.dim() < 4:
            am = am.unsqueeze(0)
        if am.dtype == torch.bool:
            mask = am
        else:
            bias = am
    p = dropout_p if train else 0.0
    out = _attention(query, key, value, softmax_n_param, scale, p, mask, bias, is_causal)
    return out.squeeze(1) if squeeze else out
