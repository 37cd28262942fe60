"""Model surgery: put softmax_n attention (the HIP kernel) into existing models.

Mirrors the reference's surgery interface without its MosaicML-composer dependency:
    policy_registry / PolicyRegistry.register   surgery/surgery_functions/utils.py:12-97   (same signature rules and errors)
    apply_attention_softmax_n(model, n, opts)    surgery/attention_softmax_n.py:19-63       (module replacement by policy)
A surgery function has the signature `(module: torch.nn.Module, module_index: int, softmax_n_param: float)` and returns the
module to keep in the model (the same object, modified in place, or a new one) or None to leave it alone.

Built-in policy: Hugging Face self-attention modules that dispatch through `transformers.AttentionInterface`
(transformers >= 4.48; BERT / RoBERTa here). The reference re-implements `BertSelfAttention.forward` around its eager
`softmax_n` (surgery_functions/_bert.py:24-121, pinned to transformers < 4.33); here the module keeps its own forward and its
attention function becomes `flash_attention_n` — fused, no [B,H,L,S] score tensor. XLNet (surgery_functions/_xlnet.py:25-75
re-implements `rel_attn_core` around the eager softmax_n): here `rel_attn_core` keeps computing the position and segment
scores, which become the `attn_bias` of one fused `flash_attention_n` call for content score, softmax_n, dropout and PV.
"""
from __future__ import annotations

import inspect
import logging
from typing import Callable, Dict, Optional, Sequence, Type, Union

import torch
from torch.nn import Module
from torch.optim import Optimizer

log = logging.getLogger(__name__)

AttentionSoftmaxNReplacementFunction = Callable[[Module, int, float], Optional[Module]]
HF_ATTENTION_NAME = "softmax_n_hip"   # key under which the attention function is registered with transformers

__all__ = ["PolicyRegistry", "policy_registry", "apply_attention_softmax_n", "register_hf_attention", "HF_ATTENTION_NAME"]



class PolicyRegistry(Dict[Type[Module], AttentionSoftmaxNReplacementFunction]):
    """module class -> surgery function (reference surgery_functions/utils.py:12-93)."""

    def register(self, *modules: Type[Module]):
        if len(modules) == 0:
            raise ValueError("register() needs at least one torch.nn.Module class to attach the surgery function to")

        def check_signature(func: Callable) -> None:
            params = list(inspect.signature(func).parameters.items())
            if len(params) != 3:
                raise ValueError(f"a surgery function takes (module, module_index, softmax_n_param); {func} takes {len(params)} arguments")
            (_, p_module), (_, p_index), (n_name, p_n) = params
            # annotations may be strings under `from __future__ import annotations`
            def is_(annotation, typ, name):
                return annotation is typ or annotation == name or annotation == f"torch.nn.{name}" or annotation == f"nn.{name}"
            if not is_(p_module.annotation, Module, "Module"):
                raise TypeError(f'the first argument of surgery function {func} must be annotated "torch.nn.Module"')
            if not is_(p_index.annotation, int, "int"):
                raise TypeError(f'the second argument of surgery function {func} must be annotated "int"')
            if not is_(p_n.annotation, float, "float"):
                raise TypeError(f'the third argument of surgery function {func} must be annotated "float"')
            if n_name != "softmax_n_param":
                raise NameError(f'the third argument of surgery function {func} must be named "softmax_n_param"')

        def wrapper(func: AttentionSoftmaxNReplacementFunction) -> AttentionSoftmaxNReplacementFunction:
            check_signature(func)
            for m in modules:
                if not (isinstance(m, type) and issubclass(m, Module)):
                    raise TypeError(f"{getattr(m, '__name__', m)} is not a subclass of torch.nn.Module")
                if m in self:
                    raise ValueError(f"a surgery function is already registered for {m.__name__}")
                self[m] = func
            return func

        return wrapper


policy_registry = PolicyRegistry()


def _swap_optimizer_params(optimizers, old: Module, new: Module) -> None:
    """keep optimizers that were built on `model.parameters()` pointing at the parameters of the replacement module"""
    if optimizers is None:
        return
    if isinstance(optimizers, Optimizer):
        optimizers = [optimizers]
    old_params = list(old.parameters())
    new_params = list(new.parameters())
    if {id(p) for p in old_params} == {id(p) for p in new_params}:
        return
    for opt in optimizers:
        for group in opt.param_groups:
            kept = [p for p in group["params"] if all(p is not q for q in old_params)]
            if len(kept) != len(group["params"]):     # this group held the old module's parameters
                for p in old_params:
                    opt.state.pop(p, None)
                group["params"] = kept + [p for p in new_params if all(p is not q for q in kept)]
                new_params = []


def apply_attention_softmax_n(model: Module, softmax_n_param: float,
                              optimizers: Optional[Union[Optimizer, Sequence[Optimizer]]] = None) -> int:
    """Run every registered surgery function over the matching sub-modules of `model` (reference
    surgery/attention_softmax_n.py:19-63). Returns the number of modules converted; logs a warning when it is 0."""
    replaced = 0
    index = 0
    # parents first, children collected before any replacement so a replacement's own sub-modules are not revisited
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            func = next((f for cls, f in policy_registry.items() if type(child) is cls), None)
            if func is None:
                func = next((f for cls, f in policy_registry.items() if isinstance(child, cls)), None)
            if func is None:
                continue
            new = func(child, index, softmax_n_param=float(softmax_n_param))
            index += 1
            if new is None:
                continue
            if new is not child:
                setattr(parent, name, new)
                _swap_optimizer_params(optimizers, child, new)
            replaced += 1
    if replaced == 0:
        supported = "".join(sorted("\n\t" + c.__module__ + "." + c.__name__ for c in policy_registry))
        log.warning("AttentionSoftmaxN had no effect on the model! Supported module classes: %s", supported)
    else:
        log.info("%d instances of AttentionSoftmaxN added", replaced)
    return replaced


# ------------------------------------------------------------------------------------------- Hugging Face transformers
def _hf_attention(module: Module, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                  attention_mask: Optional[torch.Tensor], scaling: Optional[float] = None, dropout: float = 0.0, **kwargs):
    """`transformers` attention-interface function: [B,H,L,E] in, ([B,L,H,Ev], None) out. The additive float mask HF builds
    ([B,1,L,S], finfo.min where hidden) is passed as `attn_bias` through its broadcast strides."""
    from .flash_attn import flash_attention_n
    n = float(getattr(module, "softmax_n_param", 0.0))
    if kwargs.get("head_mask") is not None:
        raise NotImplementedError("head_mask multiplies the attention probabilities, which the fused kernel never materialises")
    bias = mask = None
    if attention_mask is not None:
        if attention_mask.dtype == torch.bool:
            mask = attention_mask[..., : key.shape[-2]]   # row stride 0 (from _hf_mask) -> the key-padding kernels
        else:
            bias = attention_mask[..., : key.shape[-2]]
            keymask = _as_key_padding_mask(bias)
            if keymask is not None:
                # row-broadcast additive mask: its hidden keys as the kernels' key-padding mask (padded tiles skipped). HF builds
                # these as 0 / finfo.min from a binary mask, so the visible entries add nothing and the bias is dropped - unless
                # ADDITIVE_KEY_MASKS_MAY_BE_SOFT says a caller feeds non-binary masks: then the finite entries stay a [B,1,1,S] bias
                mask = keymask
                bias = torch.where(keymask, bias[..., :1, :], torch.zeros((), dtype=bias.dtype, device=bias.device)) if ADDITIVE_KEY_MASKS_MAY_BE_SOFT else None
    out = flash_attention_n(query, key, value, softmax_n_param=n, scale=scaling, dropout_p=dropout if module.training else 0.0,
                            attn_mask=mask, attn_bias=bias, is_causal=bool(kwargs.get("is_causal", False)) and query.shape[2] > 1)
    return out.transpose(1, 2).contiguous(), None


# transformers' extended attention mask is (1 - mask) * finfo.min: 0 / min for the binary masks tokenizers produce. A float mask
# with values strictly between 0 and 1 ("soft" masking) gives finite non-zero additive entries; set this to True to keep them as
# an additive key bias next to the key-padding mask (the bias + key-padding kernels instead of the plain key-padding ones).
ADDITIVE_KEY_MASKS_MAY_BE_SOFT = False


def _as_key_padding_mask(additive: torch.Tensor) -> Optional[torch.Tensor]:
    """Older `transformers` versions (and direct callers) hand every layer an ADDITIVE mask, [B,1,1,S] (or a row-broadcast
    expansion of it) with 0 for real tokens and a huge negative number for padding. A row-broadcast additive mask yields the
    boolean key mask `additive > -1e4` WITHOUT inspecting its values on the host (no device synchronisation, nothing cached -
    a cache keyed on the tensor's address would return a previous batch's mask once the allocator reuses the address):
    entries at or below the threshold (finfo.min, -inf, -1e9, -1e4) are hidden keys - the kernels skip their tiles.
    The VALUES above the threshold are not looked at here: `_hf_attention` drops them (0 by HF's construction) unless
    ADDITIVE_KEY_MASKS_MAY_BE_SOFT is set, in which case it keeps them as an additive key bias next to this mask.
    Anything with a real row dimension (e.g. causal) returns None and stays an additive bias only."""
    if additive.dim() != 4 or additive.shape[1] != 1:
        return None
    if additive.shape[-2] != 1:
        if additive.stride(-2) != 0:
            return None
        additive = additive[..., :1, :]   # expanded view of a row-broadcast mask
    return additive > -1e4


def _hf_mask(batch_size: int, cache_position: Optional[torch.Tensor] = None, kv_length: Optional[int] = None, kv_offset: int = 0,
             mask_function=None, attention_mask: Optional[torch.Tensor] = None, q_length: Optional[int] = None, q_offset: int = 0,
             **kwargs):
    """`transformers.AttentionMaskInterface` function for HF_ATTENTION_NAME. Without a registered mask function transformers
    hands a custom attention function `attention_mask=None` - every layer would silently attend to padding tokens.
    Both calling conventions are served: transformers 4.53 - 4.5x pass `cache_position` (the query positions), newer versions
    `q_length` / `q_offset`; whatever was received is forwarded to transformers' own `sdpa_mask`.
    Bidirectional models with a 2-D padding mask get it back as a boolean [B,1,L,S] view with ROW STRIDE 0 (True = attend),
    which is exactly the kernels' key-padding form (one byte per key, padded tiles skipped); everything else (causal,
    sliding window, packed sequences, or/and-mask functions) is built by transformers' own boolean `sdpa_mask`."""
    import inspect
    from transformers import masking_utils as mu
    if q_length is None and cache_position is not None:
        q_length = int(cache_position.shape[0])
    if (attention_mask is not None and attention_mask.dim() == 2 and kv_length is not None and q_length is not None
            and mask_function is getattr(mu, "bidirectional_mask_function", object())
            and attention_mask.shape[-1] >= kv_offset + kv_length):
        keys = attention_mask[:, kv_offset:kv_offset + kv_length].to(torch.bool)
        return keys[:, None, None, :].expand(batch_size, 1, q_length, kv_length)
    kwargs.pop("allow_is_bidirectional_skip", None)
    # never "skip" to None for a padded batch; an all-True mask may still come back as None, which means "no mask"
    have = dict(batch_size=batch_size, cache_position=cache_position, kv_length=kv_length, kv_offset=kv_offset,
                mask_function=mask_function, attention_mask=attention_mask, q_length=q_length, q_offset=q_offset, **kwargs)
    params = inspect.signature(mu.sdpa_mask).parameters
    if not any(prm.kind is inspect.Parameter.VAR_KEYWORD for prm in params.values()):
        have = {k_: v_ for k_, v_ in have.items() if k_ in params}   # this version's sdpa_mask takes only what it names
    elif "cache_position" not in params:
        have.pop("cache_position", None)
    if mask_function is None:
        have.pop("mask_function", None)   # sdpa_mask's own default (causal)
    return mu.sdpa_mask(**have)


def register_hf_attention() -> bool:
    """Register the attention function with transformers (once) and the built-in surgery policy for the self-attention
    classes of the installed version. Returns False when transformers has no AttentionInterface."""
    try:
        from transformers import AttentionInterface
    except Exception:   # transformers missing or too old
        return False
    if HF_ATTENTION_NAME not in AttentionInterface._global_mapping:
        AttentionInterface.register(HF_ATTENTION_NAME, _hf_attention)
    try:   # the padding mask only reaches a custom attention function whose name also has a mask function
        from transformers import AttentionMaskInterface
        if HF_ATTENTION_NAME not in AttentionMaskInterface._global_mapping:
            AttentionMaskInterface.register(HF_ATTENTION_NAME, _hf_mask)
    except ImportError:   # transformers 4.48 - 4.52: the model builds the additive mask itself and passes it on
        pass
    classes = []
    for mod_name, cls_names in (("transformers.models.bert.modeling_bert", ("BertSelfAttention", "BertCrossAttention")),
                                ("transformers.models.roberta.modeling_roberta", ("RobertaSelfAttention", "RobertaCrossAttention"))):
        try:
            mod = __import__(mod_name, fromlist=list(cls_names))
        except Exception:
            continue
        classes += [getattr(mod, c) for c in cls_names if hasattr(mod, c)]
    classes = [c for c in classes if c not in policy_registry]
    if classes:
        policy_registry.register(*classes)(hf_self_attention_surgery)
    try:
        from transformers.models.xlnet.modeling_xlnet import XLNetRelativeAttention
        if XLNetRelativeAttention not in policy_registry:
            policy_registry.register(XLNetRelativeAttention)(xlnet_relative_attention_surgery)
    except Exception:
        pass
    return True


def hf_self_attention_surgery(module: Module, module_index: int, softmax_n_param: float) -> Optional[Module]:
    """Built-in policy for HF self-attention modules: remember n on the module and route its attention call to the HIP
    kernel (the module's projections, cache handling and output reshaping stay its own)."""
    del module_index
    config = getattr(module, "config", None)
    if config is None or not hasattr(config, "_attn_implementation"):
        return None
    module.softmax_n_param = float(softmax_n_param)
    config._attn_implementation = HF_ATTENTION_NAME   # the config object is shared by all layers of the model
    return module


# ------------------------------------------------------------------------------------------------------------ XLNet
def _xlnet_rel_attn_core(self, q_head, k_head_h, v_head_h, k_head_r, seg_mat=None, attn_mask=None, head_mask=None,
                         output_attentions=False, **kwargs):
    """Replacement for `XLNetRelativeAttention.rel_attn_core` (same arguments and return value; tensors are [len, batch,
    head, dim]). Position (bd) and segment (ef) scores are computed as the module always did and enter one fused
    `flash_attention_n` call as the additive bias; the content score, softmax_n, dropout and the weighted sum happen in the
    kernel. With `output_attentions=True` the probabilities must be returned, so that case takes the module's own einsum
    route with the softmax_n row kernel."""
    from .flash_attn import flash_attention_n
    from .softmax import softmax_n
    n = float(getattr(self, "softmax_n_param", 0.0))
    bd = torch.einsum("ibnd,jbnd->bnij", q_head + self.r_r_bias, k_head_r)
    bd = self.rel_shift_bnij(bd, klen=k_head_h.shape[0])
    if seg_mat is None:
        extra = bd
    else:
        ef = torch.einsum("ibnd,snd->ibns", q_head + self.r_s_bias, self.seg_embed)
        extra = bd + torch.einsum("ijbs,ibns->bnij", seg_mat, ef)
    visible = None
    if attn_mask is not None:          # [i, j, b, n] (n may be 1), 1 = masked
        # (contiguous: the comparison keeps the permuted memory order of its input, whose key stride is not 1 - such a mask takes the
        # element-load kernels; [b, n or 1, i, j] with unit key stride takes the vector path, the head broadcast stays a stride 0)
        visible = (torch.einsum("ijbn->bnij", attn_mask) == 0).contiguous()
    if output_attentions or head_mask is not None:
        # the probabilities themselves are wanted (or multiplied by head_mask, as transformers 4.x passes it and the reference's
        # rel_attn_core applies it, surgery_functions/_xlnet.py:66-67): the module's einsum route with the softmax_n row kernel
        ac = torch.einsum("ibnd,jbnd->bnij", q_head + self.r_w_bias, k_head_h)
        score = (ac + extra) * self.scale
        if visible is not None:
            score = score.masked_fill(~visible, float("-inf"))
        prob = self.dropout(softmax_n(score, n=n, dim=3))
        if head_mask is not None:
            prob = prob * torch.einsum("ijbn->bnij", head_mask)
        if not output_attentions:
            return torch.einsum("bnij,jbnd->ibnd", prob, v_head_h)
        return torch.einsum("bnij,jbnd->ibnd", prob, v_head_h), torch.einsum("bnij->ijbn", prob)
    q = (q_head + self.r_w_bias).permute(1, 2, 0, 3)      # [b, n, i, d]
    k = k_head_h.permute(1, 2, 0, 3)
    v = v_head_h.permute(1, 2, 0, 3)
    out = flash_attention_n(q, k, v, softmax_n_param=n, scale=self.scale, attn_bias=extra * self.scale, attn_mask=visible,
                            dropout_p=self.dropout.p if self.training else 0.0)
    return out.permute(2, 0, 1, 3)                          # [i, b, n, d]


def xlnet_relative_attention_surgery(module: Module, module_index: int, softmax_n_param: float) -> Optional[Module]:
    """Built-in policy for `transformers` XLNetRelativeAttention (reference surgery_functions/_xlnet.py:11-22)."""
    from types import MethodType
    del module_index
    if not hasattr(module, "rel_attn_core") or not hasattr(module, "rel_shift_bnij"):
        return None
    module.softmax_n_param = float(softmax_n_param)
    module.rel_attn_core = MethodType(_xlnet_rel_attn_core, module)
    return module
