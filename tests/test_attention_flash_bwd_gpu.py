"""csrc/attention_flash_bwd.hip (key-tiled backward of the rel-pos attention core: any T, head dim <= 64, no [T,T] matrices
in HBM) against fp64 autograd through a restatement of RelPositionMultiHeadAttention.forward
(multi_head_attention.py:197-250) on the kernel's bf16 inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_attention_flash_gpu import _inputs  # noqa: E402


def _forward64(q, k, v, p, bu, bv, lens, T, dk, keep=None, keep_scale=1.0):
    """q,k,v [B,H,T,dk], p [H,2T-1,dk] (float64, may require grad) -> o [B,H,T,dk]."""
    dev = q.device
    qu = q + bu[None, :, None, :]
    qv = q + bv[None, :, None, :]
    ac = qu @ k.transpose(-1, -2)
    full = torch.einsum("bhid,hrd->bhir", qv, p)
    i = torch.arange(T, device=dev)[:, None]; j = torch.arange(T, device=dev)[None, :]
    bd = full.gather(-1, (T - 1 - i + j).expand(*ac.shape))
    s = (ac + bd) / math.sqrt(dk)
    valid = j[None] < lens[:, None, None]
    s = s.masked_fill(~valid[:, None], float("-inf"))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep * keep_scale
    o = pr @ v
    qvalid = (torch.arange(T, device=dev)[None, :] < lens[:, None])[:, None, :, None]
    return o * qvalid


def _grads64(qkv, pl, bu, bv, lens, dctx, B, T, H, dk, keep=None, keep_scale=1.0):
    d = H * dk
    x = qkv.double().view(B, T, 3, H, dk)
    q, k, v = (x[:, :, n].transpose(1, 2).clone().requires_grad_(True) for n in range(3))
    p = pl.double()[:2 * T - 1].view(2 * T - 1, H, dk).permute(1, 0, 2).clone().requires_grad_(True)
    u, w = bu.double().clone().requires_grad_(True), bv.double().clone().requires_grad_(True)
    o = _forward64(q, k, v, p, u, w, lens, T, dk, keep, keep_scale)
    g = dctx.double().view(B, T, H, dk).transpose(1, 2)
    (o * g).sum().backward()
    dqkv = torch.stack([t.grad.transpose(1, 2) for t in (q, k, v)], dim=2).reshape(B * T, 3 * d)
    dpl = p.grad.permute(1, 0, 2).reshape(2 * T - 1, d)
    return dqkv, dpl, u.grad, w.grad


def _check(name, got, ref, rel):
    err = (got.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rel * scale + 1e-4, (name, err, scale)


@pytest.mark.parametrize("B,T,H,dk,lens", [(3, 100, 2, 64, None), (2, 376, 4, 64, None), (2, 751, 8, 64, [751, 500]),
                                           (3, 126, 4, 36, [126, 64, 5]), (2, 64, 1, 64, [64, 63]), (1, 65, 2, 48, [65]),
                                           (4, 200, 2, 64, [128, 200, 1, 129]), (2, 17, 2, 64, [17, 9])])
def test_flash_backward_matches_fp64_autograd(B, T, H, dk, lens):
    from indic_cl_asr_amd.ops import fast
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=T + dk + 1, lens=lens)
    g0 = torch.Generator().manual_seed(T)
    dctx = (torch.randn(B * T, H * dk, generator=g0) * 0.5).bfloat16().cuda()
    ctx, lse = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, want_lse=True)
    dqkv, dpl, du, dv = fast.relpos_attention_flash_bwd(qkv, pl, bu, bv, ln, ctx, dctx, lse, B, T, H, dk)
    r_qkv, r_pl, r_u, r_v = _grads64(qkv, pl, bu, bv, ln, dctx, B, T, H, dk)
    d = H * dk
    # bf16 probabilities / score gradients as MFMA operands, bf16 outputs: a few 1e-2 of each tensor's largest element
    _check("dq", dqkv[:, :d], r_qkv[:, :d], 3e-2)
    _check("dk", dqkv[:, d:2 * d], r_qkv[:, d:2 * d], 3e-2)
    _check("dv", dqkv[:, 2 * d:], r_qkv[:, 2 * d:], 3e-2)
    _check("dpl", dpl[:2 * T - 1], r_pl, 3e-2)
    _check("du", du, r_u, 3e-2)
    _check("dv_bias", dv, r_v, 3e-2)
    # rows of padded frames carry no gradient at all
    for b in range(B):
        n = int(ln[b])
        if n < T:
            assert dqkv.view(B, T, -1)[b, n:].abs().max().item() == 0.0


def test_flash_backward_regenerates_the_forward_dropout_mask():
    """T = 64, dk = 64: with one-hot values the forward's output IS dropout(P), which exposes the keep mask of (seed, head,
    query, key); fp64 autograd with that mask must reproduce the kernel's gradients."""
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 2, 64, 2, 64
    p_drop, seed = 0.25, 77
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=5, lens=[64, 50])
    probe = qkv.clone().view(B, T, 3, H, dk)
    probe[:, :, 2] = torch.eye(64, device="cuda", dtype=torch.bfloat16)[None, :, None, :]
    pd = fast.relpos_attention_flash(probe.view(B * T, -1), pl, bu, bv, ln, B, T, H, dk, dropout_p=p_drop, seed=seed)
    keep = (pd.view(B, T, H, dk).transpose(1, 2) > 0).double()                       # [B,H,T(query),64(key)]
    frac = keep[0, :, :, :].mean().item()
    assert abs(frac - 0.75) < 0.03
    g0 = torch.Generator().manual_seed(3)
    dctx = (torch.randn(B * T, H * dk, generator=g0) * 0.5).bfloat16().cuda()
    ctx, lse = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, dropout_p=p_drop, seed=seed, want_lse=True)
    dqkv, dpl, du, dv = fast.relpos_attention_flash_bwd(qkv, pl, bu, bv, ln, ctx, dctx, lse, B, T, H, dk, dropout_p=p_drop, seed=seed)
    ks = 256.0 / (256.0 - round(p_drop * 256))
    # keys beyond the length never show in the probe (P = 0 there): treat them as kept -- they carry no gradient either way
    r_qkv, r_pl, r_u, r_v = _grads64(qkv, pl, bu, bv, ln, dctx, B, T, H, dk, keep=keep, keep_scale=ks)
    d = H * dk
    _check("dq", dqkv[:, :d], r_qkv[:, :d], 3e-2)
    _check("dk", dqkv[:, d:2 * d], r_qkv[:, d:2 * d], 3e-2)
    _check("dv", dqkv[:, 2 * d:], r_qkv[:, 2 * d:], 3e-2)
    _check("dpl", dpl[:2 * T - 1], r_pl, 3e-2)
    _check("du", du, r_u, 3e-2)


def test_flash_backward_agrees_with_the_row_pass_backward():
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 4, 376, 4, 64
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=3)
    dctx = (torch.randn(B * T, H * dk, device="cuda") * 0.5).bfloat16()
    ctx, lse = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, want_lse=True)
    a = fast.relpos_attention_flash_bwd(qkv, pl, bu, bv, ln, ctx, dctx, lse, B, T, H, dk)
    ctx2 = fast.relpos_attention(qkv, pl, bu, bv, ln, B, T, H, dk)
    b = fast.relpos_attention_bwd(qkv, pl, bu, bv, ln, ctx2, dctx, B, T, H, dk)
    for x, y, nm in zip(a, b, ("dqkv", "dpl", "du", "dv")):
        err = (x.float() - y.float()).abs().max().item()
        assert err <= 3e-2 * y.float().abs().max().item() + 1e-4, (nm, err)


def test_flash_backward_with_an_empty_utterance_gives_zero_gradients_for_it():
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 3, 70, 2, 64
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=2, lens=[70, 0, 33])
    dctx = (torch.randn(B * T, H * dk, device="cuda") * 0.5).bfloat16()
    ctx, lse = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, want_lse=True)
    dqkv, dpl, du, dv = fast.relpos_attention_flash_bwd(qkv, pl, bu, bv, ln, ctx, dctx, lse, B, T, H, dk)
    assert ctx.view(B, T, -1)[1].abs().max().item() == 0.0
    assert dqkv.view(B, T, -1)[1].abs().max().item() == 0.0
    r_qkv, r_pl, r_u, r_v = _grads64(qkv, pl, bu, bv, ln.clamp(min=0), dctx, B, T, H, dk)
    keep = torch.tensor([0, 2], device="cuda")
    got = dqkv.view(B, T, -1)[keep].double(); ref = torch.nan_to_num(r_qkv.view(B, T, -1)[keep])
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item() + 1e-4
    assert torch.isfinite(dpl.float()).all() and torch.isfinite(du).all()
