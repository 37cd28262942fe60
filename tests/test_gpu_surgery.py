"""Surgery on a real model (-m gpu): a small random-weight BERT whose self-attention is switched to the HIP softmax_n
kernel, against the same weights with an eager softmax_n attention built from the oracle (the reference's
tests/cpu/surgery/test_bert.py checks its own patched forward the same way)."""
import pytest
import torch

from oracle.ref_attention import ref_softmax_n

pytestmark = pytest.mark.gpu

transformers = pytest.importorskip("transformers")


def _oracle_attention(module, query, key, value, attention_mask, scaling=None, dropout=0.0, **kwargs):
    """eager attention with the oracle's softmax_n, fp32 arithmetic"""
    n = float(getattr(module, "softmax_n_param", 0.0))
    scaling = query.size(-1) ** -0.5 if scaling is None else scaling
    w = torch.matmul(query.float(), key.float().transpose(2, 3)) * scaling
    if attention_mask is not None:
        w = w + attention_mask.float()
    w = ref_softmax_n(w, n=n)
    out = torch.matmul(w, value.float()).to(query.dtype)
    return out.transpose(1, 2).contiguous(), None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n", [0.0, 1.0, 2.5])
def test_bert_with_softmax_n_attention(pkg, dev, n, dtype):
    from transformers import AttentionInterface, BertConfig, BertModel
    from flash_attention_softmax_n_amd import surgery
    if not surgery.register_hf_attention():
        pytest.skip("transformers without AttentionInterface")
    if "oracle_softmax_n" not in AttentionInterface._global_mapping:
        AttentionInterface.register("oracle_softmax_n", _oracle_attention)
        try:   # without a mask function transformers passes attention_mask=None to a custom attention name
            from transformers import AttentionMaskInterface
            from transformers.masking_utils import eager_mask
            AttentionMaskInterface.register("oracle_softmax_n", eager_mask)
        except ImportError:
            pass
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=100, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=160, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).to(dev).to(dtype).eval()
    ids = torch.randint(0, 100, (3, 150), device=dev)
    att = torch.ones(3, 150, dtype=torch.long, device=dev)
    att[1, 100:] = 0          # padded sequences -> additive mask -> attn_bias path
    att[2, 37:] = 0

    # expected: same weights, eager softmax_n attention
    for m in model.modules():
        if type(m).__name__ == "BertSelfAttention":
            m.softmax_n_param = n
    model.config._attn_implementation = "oracle_softmax_n"
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=att).last_hidden_state.float()

    if n == 0.0:   # softmax_0 = the model's own attention: the oracle route itself must honour the padding mask
        model.config._attn_implementation = "eager"
        with torch.no_grad():
            own = model(input_ids=ids, attention_mask=att).last_hidden_state.float()
        err_own = ((own - want) * att.bool().unsqueeze(-1)).abs().max().item()
        assert err_own <= 4 * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * want.abs().max().item()
        with torch.no_grad():
            nomask = model(input_ids=ids).last_hidden_state.float()
        if dtype == torch.float16:   # (bf16 rounding through two layers is as large as the effect of the padding here)
            assert ((nomask - want) * att.bool().unsqueeze(-1)).abs().max().item() > 3 * err_own   # padding matters in this batch

    count = surgery.apply_attention_softmax_n(model, softmax_n_param=n)
    assert count == cfg.num_hidden_layers
    assert model.config._attn_implementation == surgery.HF_ATTENTION_NAME
    with torch.no_grad():
        got = model(input_ids=ids, attention_mask=att).last_hidden_state.float()
    assert torch.isfinite(got).all()
    valid = att.bool().unsqueeze(-1)
    err = ((got - want) * valid).abs().max().item()
    # two encoder layers in a 16-bit model: allow two units in the last place of the largest activation
    ulp = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * want.abs().max().item()
    assert err <= 2 * ulp, f"max-abs {err:.3e} over the un-padded positions (2 ulp = {2 * ulp:.3e})"

    # gradients flow through the kernel's backward into the projection weights
    model.train()
    out = model(input_ids=ids, attention_mask=att).last_hidden_state
    (out.float() * valid).pow(2).sum().backward()   # sum, not mean: fp16 gradients of a mean underflow
    g = model.encoder.layer[0].attention.self.query.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max().item() > 0


def _xlnet_core_with_oracle_softmax_n(self, q_head, k_head_h, v_head_h, k_head_r, seg_mat=None, attn_mask=None, output_attentions=False):
    """the module's own einsum formulation in fp32 with the oracle's softmax_n (what the reference's patched XLNet computes)"""
    n = float(getattr(self, "softmax_n_param", 0.0))
    f = lambda t: t.float()
    ac = torch.einsum("ibnd,jbnd->bnij", f(q_head + self.r_w_bias), f(k_head_h))
    bd = torch.einsum("ibnd,jbnd->bnij", f(q_head + self.r_r_bias), f(k_head_r))
    bd = self.rel_shift_bnij(bd, klen=ac.shape[3])
    ef = 0
    if seg_mat is not None:
        ef = torch.einsum("ibns,ijbs->bnij", torch.einsum("ibnd,snd->ibns", f(q_head + self.r_s_bias), f(self.seg_embed)), f(seg_mat))
    score = (ac + bd + ef) * self.scale
    if attn_mask is not None:
        score = score.masked_fill(torch.einsum("ijbn->bnij", attn_mask) != 0, float("-inf"))
    prob = ref_softmax_n(score, n=n)
    return torch.einsum("bnij,jbnd->ibnd", prob, f(v_head_h)).to(q_head.dtype)


@pytest.mark.parametrize("n", [0.0, 1.5])
def test_xlnet_with_softmax_n_attention(pkg, dev, n):
    from types import MethodType
    from transformers import XLNetConfig, XLNetModel
    from flash_attention_softmax_n_amd import surgery
    if not surgery.register_hf_attention():
        pytest.skip("transformers without AttentionInterface")
    dtype = torch.float16
    torch.manual_seed(0)
    cfg = XLNetConfig(vocab_size=100, d_model=128, n_layer=2, n_head=2, d_inner=256, dropout=0.0)
    model = XLNetModel(cfg).to(dev).to(dtype).eval()
    ids = torch.randint(0, 100, (3, 96), device=dev)
    att = torch.ones(3, 96, device=dev)
    att[1, 60:] = 0
    seg = torch.zeros(3, 96, dtype=torch.long, device=dev)
    seg[:, 40:] = 1
    layers = [m for m in model.modules() if type(m).__name__ == "XLNetRelativeAttention"]
    for m in layers:       # expected: same weights, the module's einsum route with the oracle's softmax_n
        m.softmax_n_param = n
        m.rel_attn_core = MethodType(_xlnet_core_with_oracle_softmax_n, m)
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=att, token_type_ids=seg).last_hidden_state.float()
    for m in layers:
        del m.rel_attn_core
    assert surgery.apply_attention_softmax_n(model, softmax_n_param=n) == cfg.n_layer
    import warnings
    with torch.no_grad(), warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = model(input_ids=ids, attention_mask=att, token_type_ids=seg).last_hidden_state.float()
    # the fused call runs on the vector path (position / segment scores as an aligned bias, the visibility mask with unit key stride):
    # round 4 handed the kernel a permuted mask view and took the element-load kernels here
    assert not [w for w in caught if "element-load" in str(w.message)], [str(w.message)[:120] for w in caught]
    with torch.no_grad():
        got_p = model(input_ids=ids, attention_mask=att, token_type_ids=seg, output_attentions=True)
    assert torch.isfinite(got).all()
    valid = att.bool().unsqueeze(-1)
    ulp = 2.0 ** -10 * want.abs().max().item()
    err = ((got - want) * valid).abs().max().item()
    assert err <= 3 * ulp, f"max-abs {err:.3e} (3 ulp = {3 * ulp:.3e})"
    err_p = ((got_p.last_hidden_state.float() - want) * valid).abs().max().item()
    assert err_p <= 3 * ulp and got_p.attentions is not None
