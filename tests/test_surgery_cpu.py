"""Host logic of the surgery interface (no GPU): registry rules (reference surgery_functions/utils.py:62-93), module
replacement and optimizer bookkeeping (reference attention_softmax_n.py:19-63, tests/cpu/surgery/test_register.py)."""
import logging
from types import MethodType

import pytest
import torch
from torch import Tensor
from torch.nn import Linear, Module

from oracle.ref_attention import analytic_answer, ref_attention_n

import flash_attention_softmax_n_amd.surgery as surgery
from flash_attention_softmax_n_amd.surgery import PolicyRegistry, apply_attention_softmax_n, policy_registry

LOGIT_SCALE, GAIN = 0.2, 2.0


class TwiceAttention(Module):
    """2 x softmax_0 attention (the reference test's dummy module, tests/cpu/surgery/test_register.py:26-35)"""

    def __init__(self):
        super().__init__()
        self.gain = GAIN

    def forward(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        return self.gain * ref_attention_n(q, k, v, softmax_n_param=0.0, scale=LOGIT_SCALE)


class TinyNet(Module):
    def __init__(self):
        super().__init__()
        self.attn = TwiceAttention()
        self.other = Linear(4, 4)

    def forward(self, q, k, v):
        return self.attn(q, k, v)


def _new_forward(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    return self.gain * ref_attention_n(q, k, v, softmax_n_param=self.n, scale=LOGIT_SCALE)


@pytest.fixture
def clean_registry():
    saved = dict(policy_registry)
    yield policy_registry
    policy_registry.clear()
    policy_registry.update(saved)


@pytest.mark.parametrize("weight", [10, 1, 0.1, -0.1, -1, -10])
def test_register_and_apply(clean_registry, weight):
    seen = []

    @policy_registry.register(TwiceAttention)
    def converter(module: Module, module_index: int, softmax_n_param: float) -> Module:
        seen.append(module_index)
        module.n = softmax_n_param
        setattr(module, "forward", MethodType(_new_forward, module))
        return module

    N, L, S, E = 2, 3, 5, 8
    model = TinyNet()
    q, k, v = (weight * torch.ones(N, sz, E) for sz in (L, S, S))
    before = model(q, k, v)
    assert torch.allclose(before, torch.full_like(before, GAIN * analytic_answer(weight, S, E, LOGIT_SCALE, 0.0)), atol=1e-5)
    assert apply_attention_softmax_n(model, softmax_n_param=2.0) == 1
    assert seen == [0] and model.attn.n == 2.0
    after = model(q, k, v)
    assert torch.allclose(after, torch.full_like(after, GAIN * analytic_answer(weight, S, E, LOGIT_SCALE, 2.0)), atol=1e-5)


def test_registry_rejects_bad_surgery_functions():
    reg = PolicyRegistry()
    with pytest.raises(ValueError):
        reg.register()
    with pytest.raises(ValueError):
        reg.register(TwiceAttention)(lambda module, module_index: module)

    def bad_first(module: int, module_index: int, softmax_n_param: float):
        return None

    def bad_second(module: Module, module_index: float, softmax_n_param: float):
        return None

    def bad_third(module: Module, module_index: int, softmax_n_param: int):
        return None

    def bad_name(module: Module, module_index: int, n: float):
        return None

    def good(module: Module, module_index: int, softmax_n_param: float):
        return None

    for f, err in ((bad_first, TypeError), (bad_second, TypeError), (bad_third, TypeError), (bad_name, NameError)):
        with pytest.raises(err):
            reg.register(TwiceAttention)(f)
    with pytest.raises(TypeError):
        reg.register(int)(good)
    reg.register(TwiceAttention)(good)
    with pytest.raises(ValueError):
        reg.register(TwiceAttention)(good)
    assert reg[TwiceAttention] is good


def test_no_match_warns_and_none_result_keeps_module(clean_registry, caplog):
    policy_registry.clear()
    model = TinyNet()
    with caplog.at_level(logging.WARNING, logger=surgery.__name__):
        assert apply_attention_softmax_n(model, 1.0) == 0
    assert "had no effect" in caplog.text

    @policy_registry.register(TwiceAttention)
    def declines(module: Module, module_index: int, softmax_n_param: float):
        return None

    old = model.attn
    assert apply_attention_softmax_n(model, 1.0) == 0 and model.attn is old


def test_replacement_module_and_optimizer_params(clean_registry):
    @policy_registry.register(Linear)
    def widen(module: Module, module_index: int, softmax_n_param: float) -> Module:
        return Linear(module.in_features, module.out_features, bias=False)

    model = TinyNet()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    old_params = list(model.other.parameters())
    assert apply_attention_softmax_n(model, 1.0, optimizers=opt) == 1
    new_params = list(model.other.parameters())
    assert len(new_params) == 1 and model.other.bias is None
    in_opt = [p for g in opt.param_groups for p in g["params"]]
    assert all(any(p is q for q in in_opt) for p in new_params)
    assert not any(any(p is q for q in in_opt) for p in old_params)


def test_hf_policy_registration_is_idempotent(clean_registry):
    pytest.importorskip("transformers")
    if not surgery.register_hf_attention():
        pytest.skip("transformers without AttentionInterface")
    n1 = len(policy_registry)
    assert surgery.register_hf_attention() and len(policy_registry) == n1
    from transformers import AttentionInterface
    from transformers.models.bert.modeling_bert import BertSelfAttention
    assert surgery.HF_ATTENTION_NAME in AttentionInterface._global_mapping
    assert policy_registry[BertSelfAttention] is surgery.hf_self_attention_surgery
    from transformers.models.xlnet.modeling_xlnet import XLNetRelativeAttention
    assert policy_registry[XLNetRelativeAttention] is surgery.xlnet_relative_attention_surgery


def test_padding_mask_reaches_the_attention_function_after_surgery(clean_registry, monkeypatch):
    """A padded batch through a surgically converted BERT (CPU plumbing test: the kernel call is replaced by the oracle):
    the attention function must RECEIVE the padding mask (transformers hands `None` to a custom attention name that has no
    mask function registered - every layer would then attend to padding), as a boolean row-broadcast [B,1,L,S] view, and with
    n = 0 the converted model must reproduce the model's own eager attention on the real tokens."""
    pytest.importorskip("transformers")
    if not surgery.register_hf_attention():
        pytest.skip("transformers without AttentionInterface")
    from transformers import BertConfig, BertModel
    import flash_attention_softmax_n_amd.flash_attn as fa
    seen = []

    def oracle_flash(query, key, value, softmax_n_param=None, scale=None, dropout_p=0.0, attn_mask=None, attn_bias=None, is_causal=False):
        seen.append(attn_mask)
        return ref_attention_n(query, key, value, softmax_n_param=softmax_n_param, scale=scale, attn_mask=attn_mask, attn_bias=attn_bias,
                               is_causal=is_causal)

    monkeypatch.setattr(fa, "flash_attention_n", oracle_flash)
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    model.config._attn_implementation = "eager"
    ids = torch.randint(0, 100, (3, 20))
    att = torch.ones(3, 20, dtype=torch.long)
    att[1, 10:] = 0
    att[2, 3:] = 0
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=att).last_hidden_state
    assert apply_attention_softmax_n(model, softmax_n_param=0.0) == cfg.num_hidden_layers
    with torch.no_grad():
        got = model(input_ids=ids, attention_mask=att).last_hidden_state
    assert len(seen) == cfg.num_hidden_layers
    for m in seen:
        assert m is not None and m.dtype == torch.bool and m.shape == (3, 1, 20, 20)
        assert m.stride(2) == 0                      # row-broadcast: the kernels' key-padding form
        assert torch.equal(m[:, 0, 0], att.bool())
    valid = att.bool().unsqueeze(-1)
    assert ((got - want) * valid).abs().max().item() < 1e-5
    # and the padded positions of sequence 1 DO differ from an unmasked run (the mask is really applied)
    with torch.no_grad():
        unmasked = model(input_ids=ids).last_hidden_state
    assert ((unmasked - got) * valid).abs().max().item() > 1e-3


def test_additive_padding_masks_are_converted_without_a_cache():
    """the float-mask route (transformers 4.48+, direct callers): successive batches whose masks happen to live at the same
    address must each get their own key mask (a pointer-keyed cache returned the previous batch's mask)"""
    counts = []
    for valid in (60, 70, 50, 64, 33):
        add = torch.zeros(2, 1, 1, 80)
        add[:, :, :, valid:] = torch.finfo(torch.float32).min
        km = surgery._as_key_padding_mask(add.expand(2, 1, 16, 80))
        counts.append(int(km[0].sum()))
        del add
    assert counts == [60, 70, 50, 64, 33]
    assert surgery._as_key_padding_mask(torch.zeros(2, 1, 16, 80)) is None      # a real row dimension: stays a bias
    assert surgery._as_key_padding_mask(torch.zeros(2, 4, 1, 80)) is None       # per-head: not HF's padding mask


def test_hf_mask_serves_both_transformers_calling_conventions():
    """transformers 4.53 - 4.5x call a registered mask function with `cache_position` (no q_length / q_offset), newer versions with
    q_length / q_offset: both must give the boolean [B,1,L,S] key-padding view (row stride 0) for a bidirectional padded batch and
    otherwise reach transformers' own sdpa_mask without a TypeError."""
    mu = pytest.importorskip("transformers.masking_utils")
    bidir = getattr(mu, "bidirectional_mask_function", None)
    am = torch.ones(2, 12, dtype=torch.long)
    am[1, 9:] = 0
    if bidir is not None:
        new = surgery._hf_mask(batch_size=2, q_length=12, kv_length=12, q_offset=0, kv_offset=0, mask_function=bidir, attention_mask=am)
        old = surgery._hf_mask(batch_size=2, cache_position=torch.arange(12), kv_length=12, kv_offset=0, mask_function=bidir, attention_mask=am)
        for m in (new, old):
            assert m.dtype == torch.bool and tuple(m.shape) == (2, 1, 12, 12) and m.stride(2) == 0
            assert m[1, 0, 0].tolist() == [True] * 9 + [False] * 3
    # a causal request goes to sdpa_mask under either convention
    for kw in (dict(q_length=6, q_offset=0), dict(cache_position=torch.arange(6))):
        try:
            m = surgery._hf_mask(batch_size=2, kv_length=6, kv_offset=0, mask_function=mu.causal_mask_function, attention_mask=am[:, :6], **kw)
        except TypeError as e:   # only acceptable when THIS transformers version's sdpa_mask itself lacks the argument we could not derive
            pytest.fail(f"_hf_mask raised {e}")
        assert m is None or (m.dtype == torch.bool and m.shape[-2:] == (6, 6))
