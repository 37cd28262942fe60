"""CPU restatement (torch eager) of the reference's attention-with-softmax_n math. TEST INFRASTRUCTURE ONLY.

Follows, step for step:
  ref_softmax_n    <- flash_attention_softmax_n/core/functional.py:15-29  (max-shifted, n*exp(-shift) in the denominator)
  ref_attention_n  <- flash_attention_softmax_n/core/functional.py:32-93  (slow_attention_n: additive (L,S) bias built from
                      causal tril(diagonal=S-L) and a float mask, scores = q@k^T*scale + bias, softmax_n, @ v),
                      generalised the way flash_attention_softmax_n/core/flash_attn.py:87-113 combines a 4-D boolean mask,
                      an additive bias [H,L,S]/[B,H,L,S] and the causal mask into ONE additive term (hidden = -inf).
With compute_dtype=None every op runs in the input dtype exactly like the reference's eager code (a bf16 input gives a
bf16 softmax); compute_dtype=torch.float32/float64 upcasts first ("true" answer).
Deviations, deliberate: boolean masks are honoured (the reference's slow path drops them, functional.py:85-86);
a row with no visible key and n == 0 yields 0 (reference: NaN) — both documented in DESIGN.md.
Pinned by tests/test_oracle_golden.py against tests/golden/*.npz (outputs of the real reference).
"""
from math import sqrt
from typing import Optional

import torch
from torch import Tensor


def ref_softmax_n(x: Tensor, n: Optional[float] = None, dim: int = -1, dtype=None) -> Tensor:
    n = 0.0 if n is None else n
    shift = x.amax(dim=dim, keepdim=True).detach()
    shift = torch.where(torch.isinf(shift) & (shift < 0), torch.zeros_like(shift), shift)  # all-hidden row: avoid inf-inf
    num = torch.exp(x - shift)
    den = n * torch.exp(-shift) + num.sum(dim=dim, keepdim=True)
    out = num / den
    return out if dtype is None else out.to(dtype)


def additive_term(L: int, S: int, *, mask: Optional[Tensor], bias: Optional[Tensor], causal: bool, dtype, device,
                  batch_shape=()) -> Optional[Tensor]:
    """One additive tensor broadcastable to [..., L, S]: bias where visible, -inf where hidden."""
    if mask is None and bias is None and not causal:
        return None
    add = torch.zeros(*batch_shape, L, S, dtype=dtype, device=device) if bias is None else bias.to(dtype).expand(*batch_shape, L, S).clone()
    if causal:
        i = torch.arange(L, device=device).unsqueeze(-1)
        j = torch.arange(S, device=device).unsqueeze(0)
        add = add.masked_fill(j > i + (S - L), float("-inf"))  # == ~ones(L,S).tril(diagonal=S-L), functional.py:80
    if mask is not None:
        add = add.masked_fill(~mask.expand(*batch_shape, L, S), float("-inf"))
    return add


def ref_attention_n(query: Tensor, key: Tensor, value: Tensor, softmax_n_param: Optional[float] = None, scale: Optional[float] = None,
                    attn_mask: Optional[Tensor] = None, attn_bias: Optional[Tensor] = None, is_causal: bool = False,
                    compute_dtype=None) -> Tensor:
    """query [..., L, E], key [..., S, E], value [..., S, Ev]; mask bool / bias additive, broadcastable to [..., L, S]."""
    if compute_dtype is not None:
        query, key, value = query.to(compute_dtype), key.to(compute_dtype), value.to(compute_dtype)
    L, S = query.size(-2), key.size(-2)
    factor = 1.0 / sqrt(query.size(-1)) if scale is None else scale
    add = additive_term(L, S, mask=attn_mask, bias=attn_bias, causal=is_causal, dtype=query.dtype, device=query.device,
                        batch_shape=tuple(query.shape[:-2]))
    w = query @ key.transpose(-2, -1) * factor
    if add is not None:
        w = w + add
    w = ref_softmax_n(w, n=softmax_n_param, dim=-1)
    w = torch.nan_to_num(w, nan=0.0) if softmax_n_param in (None, 0, 0.0) else w  # all-hidden rows -> 0
    return w @ value  # in compute_dtype when given (the "true" answer is not rounded back)


def ref_attention_n_rows(query_rows: Tensor, row_index: Tensor, key: Tensor, value: Tensor, L_total: int, **kw) -> Tensor:
    """Attention for a subset of query rows of ONE (b,h) slice (query_rows [R,E], key [S,E], value [S,Ev]).
    The causal offset uses the full L (not the subset's length) — cf. SURVEY.md Appendix A."""
    S = key.size(-2)
    causal = kw.pop("is_causal", False)
    bias = kw.pop("attn_bias", None)
    mask = kw.pop("attn_mask", None)
    compute_dtype = kw.get("compute_dtype", None)
    dt = query_rows.dtype if compute_dtype is None else compute_dtype
    add = torch.zeros(len(row_index), S, dtype=dt) if bias is None else bias.to(dt).clone()
    if causal:
        j = torch.arange(S).unsqueeze(0)
        add = add.masked_fill(j > row_index.unsqueeze(-1) + (S - L_total), float("-inf"))
    if mask is not None:
        add = add.masked_fill(~mask, float("-inf"))
    return ref_attention_n(query_rows, key, value, attn_bias=add, **kw)


def analytic_answer(weight: float, S: int, E: int, scale: float, n: float) -> float:
    """Q=K=V=weight: out = w*S / (n*exp(-w^2 E scale) + S)   (reference tests/common.py:29-35)."""
    from math import exp
    return weight * S / (n * exp(-weight ** 2 * E * scale) + S)


def analytic_causal_answer(weight: float, L: int, S: int, E: int, scale: float, n: float):
    """Row l (1-based) sees l+S-L keys (reference tests/common.py:38-44); returns the per-row value (no N*Ev factor)."""
    from math import exp
    return [weight * (l + S - L) / (n * exp(-weight ** 2 * E * scale) + (l + S - L)) for l in range(1, L + 1)]
