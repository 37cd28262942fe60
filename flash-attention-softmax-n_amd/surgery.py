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
            raise ValueError("Registry decoration without any module class inputs has no effect.")

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
    bias = mask = None
    if attention_mask is not None:
        if attention_mask.dtype == torch.bool:
            mask = attention_mask
        else:
            bias = attention_mask[..., : key.shape[-2]]
            keymask = _as_key_padding_mask(bias)
            if keymask is not None:      # 0 / "minus infinity" padding mask: the kernels' key-padding path (no bias traffic)
                mask, bias = keymask, None
    out = flash_attention_n(query, key, value, softmax_n_param=n, scale=scaling, dropout_p=dropout if module.training else 0.0,
                            attn_mask=mask, attn_bias=bias, is_causal=bool(kwargs.get("is_causal", False)) and query.shape[2] > 1)
    return out.transpose(1, 2).contiguous(), None


_KEYMASK_CACHE = {"key": None, "value": None}


def _as_key_padding_mask(additive: torch.Tensor) -> Optional[torch.Tensor]:
    """HF models hand every attention layer the same additive mask, [B,1,1,S] with 0 for real tokens and a huge negative
    number for padding. If `additive` is exactly that (row-broadcast, only 0 and values <= -1e4), return the equivalent boolean
    key mask, else None. The check costs one device synchronisation, so its result is cached per mask tensor: one sync per
    forward pass, not per layer."""
    if additive.dim() != 4 or additive.shape[1] != 1:
        return None
    if additive.shape[-2] != 1:
        if additive.stride(-2) != 0:      # a real [.., L, S] mask (e.g. causal): not a key-padding mask
            return None
        additive = additive[..., :1, :]   # expanded view of a row-broadcast mask
    key = (additive.data_ptr(), tuple(additive.shape), additive._version, additive.dtype)
    if _KEYMASK_CACHE["key"] == key:
        return _KEYMASK_CACHE["value"]
    visible = additive == 0
    binary = bool((visible | (additive <= -1e4)).all().item())
    value = visible if binary else None
    _KEYMASK_CACHE["key"], _KEYMASK_CACHE["value"] = key, value
    return value


def register_hf_attention() -> bool:
    """Register the attention function with transformers (once) and the built-in surgery policy for the self-attention
    classes of the installed version. Returns False when transformers has no AttentionInterface."""
    try:
        from transformers import AttentionInterface
    except Exception:   # transformers missing or too old
        return False
    if HF_ATTENTION_NAME not in AttentionInterface._global_mapping:
        AttentionInterface.register(HF_ATTENTION_NAME, _hf_attention)
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
def _xlnet_rel_attn_core(self, q_head, k_head_h, v_head_h, k_head_r, seg_mat=None, attn_mask=None, output_attentions=False):
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
        visible = torch.einsum("ijbn->bnij", attn_mask) == 0
    if output_attentions:
        ac = torch.einsum("ibnd,jbnd->bnij", q_head + self.r_w_bias, k_head_h)
        score = (ac + extra) * self.scale
        if visible is not None:
            score = score.masked_fill(~visible, float("-inf"))
        prob = self.dropout(softmax_n(score, n=n, dim=3))
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
