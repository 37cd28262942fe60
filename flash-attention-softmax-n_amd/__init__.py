"""flash-attention-softmax-n for AMD MI355X (gfx950): the reference package's attention API on hand-written HIP.

Public surface = flash_attention_softmax_n/__init__.py:3-9 of the reference:
    flash_attention_n, softmax_n, slow_attention_n, flash_attention_n_triton (+ TRITON_INSTALLED for source compat)
    surgery.apply_attention_softmax_n / surgery.policy_registry  (flash_attention_softmax_n/surgery, composer-free)
Every function runs on device tensors through libfasn.so; importing the package without the built
library raises ImportError (no silent fallback).
"""
from . import _lib, dropout, statistics, surgery
from .flash_attn import flash_attention_n, flash_attention_n_triton, slow_attention_n
from .softmax import softmax_n

_lib.load()  # fail loudly at import if the HIP extension is missing

TRITON_INSTALLED = False  # kept for source compatibility: the Triton path is replaced by the HIP kernel
HIP_NATIVE = True

__all__ = ["flash_attention_n", "flash_attention_n_triton", "slow_attention_n", "softmax_n", "TRITON_INSTALLED", "HIP_NATIVE"]
