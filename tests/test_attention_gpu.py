"""HIP rel-pos attention kernel against an fp64 restatement of the reference's arithmetic (rel_shift pad/view trick
included) on the same bf16-quantised operands."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(q, k, v, p, u, vb, lens):
    """q,k,v [B,h,T,dk] (bf16-quantised, fp64 math); p [h,2T-1,dk]; reference formula multi_head_attention.py:197-250."""
    B, h, T, dk = q.shape
    qu = (q + u.view(1, h, 1, dk)).bfloat16().double()
    qv = (q + vb.view(1, h, 1, dk)).bfloat16().double()
    ac = qu @ k.double().transpose(-2, -1)
    bd = qv @ p.double().unsqueeze(0).transpose(-2, -1)           # [B,h,T,2T-1]
    bd = torch.nn.functional.pad(bd, (1, 0)).view(B, h, -1, T)[:, :, 1:].view(B, h, T, 2 * T - 1)[:, :, :, :T]
    scores = (ac + bd) / math.sqrt(dk)
    valid = torch.arange(T)[None, :] < lens[:, None]
    mask = ~(valid[:, :, None] & valid[:, None, :])
    scores = scores.masked_fill(mask.unsqueeze(1), -10000.0)
    attn = torch.softmax(scores, -1).masked_fill(mask.unsqueeze(1), 0.0)
    return attn @ v.double()


@pytest.mark.parametrize("B,T,H", [(2, 37, 2), (3, 64, 1), (2, 376, 4), (1, 17, 3)])
def test_relpos_attention_matches_reference_formula(B, T, H):
    from indic_cl_asr_amd.ops import fast
    dk = 64
    g = torch.Generator().manual_seed(T)
    d = H * dk
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.8).bfloat16()
    pl = (torch.randn(2 * T - 1, d, generator=g) * 0.8).bfloat16()
    u = torch.randn(H, dk, generator=g) * 0.3
    vb = torch.randn(H, dk, generator=g) * 0.3
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g); lens[0] = T
    ctx = fast.relpos_attention(qkv.cuda(), pl.cuda(), u.cuda(), vb.cuda(), lens.cuda(), B, T, H, dk).float().cpu()
    x = qkv.float().view(B, T, 3, H, dk)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    ref = _reference(q, k, v, pl.float().view(-1, H, dk).transpose(0, 1), u, vb, lens)   # [B,h,T,dk]
    ref = ref.transpose(1, 2).reshape(B * T, d)
    valid = (torch.arange(T)[None, :] < lens[:, None]).reshape(B * T, 1)
    err = ((ctx - ref) * valid).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item(), err          # bf16 P and bf16 output rounding
    assert (ctx * (~valid)).abs().max().item() == 0.0        # padded queries -> zero context


def _keepmask(B, H, T, p, seed):
    from indic_cl_asr_amd import _lib
    keep = torch.empty(B, H, T, T, dtype=torch.bfloat16, device="cuda")
    _lib.check(_lib.lib().ia_attn_keepmask(B, H, T, float(p), seed, _lib.ptr(keep), _lib.stream_ptr()), "keepmask")
    return keep.float().cpu()


@pytest.mark.parametrize("B,T,H,p", [(2, 37, 2, 0.0), (2, 40, 1, 0.25), (2, 376, 4, 0.1), (1, 17, 3, 0.0), (3, 100, 2, 0.5)])
def test_relpos_attention_backward_matches_autograd_of_reference_formula(B, T, H, p):
    """dqkv, d(pos projection), d(pos_bias_u/v) of the HIP backward (row pass + batched GEMMs) against fp64 autograd of
    the reference formula on the same bf16 operands and -- with dropout -- the same counter-based keep mask."""
    from indic_cl_asr_amd.ops import fast
    dk, seed = 64, 977
    g = torch.Generator().manual_seed(T + 1)
    d = H * dk
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.8).bfloat16()
    pl = (torch.randn(2 * T - 1, d, generator=g) * 0.8).bfloat16()
    u = torch.randn(H, dk, generator=g) * 0.3
    vb = torch.randn(H, dk, generator=g) * 0.3
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g); lens[0] = T
    valid = (torch.arange(T)[None, :] < lens[:, None]).reshape(B * T, 1)
    dctx = ((torch.randn(B * T, d, generator=g) * valid).bfloat16())     # padded frames never receive gradient
    dev = lambda t: t.cuda()
    ctx = fast.relpos_attention(dev(qkv), dev(pl), dev(u), dev(vb), dev(lens), B, T, H, dk, p, seed)
    dqkv, dpl, du, dvb = fast.relpos_attention_bwd(dev(qkv), dev(pl), dev(u), dev(vb), dev(lens), ctx, dev(dctx), B, T, H, dk, p, seed)
    # fp64 autograd reference
    x = qkv.double().view(B, T, 3, H, dk).requires_grad_(True)
    P = pl.double().view(-1, H, dk).requires_grad_(True)
    U, V = u.double().requires_grad_(True), vb.double().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    qu, qv = q + U.view(1, H, 1, dk), q + V.view(1, H, 1, dk)
    ac = qu @ k.transpose(-2, -1)
    bd = qv @ P.transpose(0, 1).unsqueeze(0).transpose(-2, -1)
    bd = torch.nn.functional.pad(bd, (1, 0)).view(B, H, -1, T)[:, :, 1:].view(B, H, T, 2 * T - 1)[:, :, :, :T]
    scores = (ac + bd) / math.sqrt(dk)
    vm = torch.arange(T)[None, :] < lens[:, None]
    mask = ~(vm[:, :, None] & vm[:, None, :])
    attn = torch.softmax(scores.masked_fill(mask.unsqueeze(1), -10000.0), -1).masked_fill(mask.unsqueeze(1), 0.0)
    if p > 0:
        attn = attn * _keepmask(B, H, T, p, seed).double()
    out = (attn @ v).transpose(1, 2).reshape(B * T, d)
    out.backward(dctx.double())
    def rel(a, b):
        return (a.double().cpu() - b).norm().item() / (b.norm().item() + 1e-12)
    gq = x.grad.reshape(B * T, 3 * d)
    assert rel(dqkv[:, :d], gq[:, :d]) < 2.5e-2, ("dq", rel(dqkv[:, :d], gq[:, :d]))
    assert rel(dqkv[:, d:2 * d], gq[:, d:2 * d]) < 2.5e-2, ("dk", rel(dqkv[:, d:2 * d], gq[:, d:2 * d]))
    assert rel(dqkv[:, 2 * d:], gq[:, 2 * d:]) < 2.5e-2, ("dv", rel(dqkv[:, 2 * d:], gq[:, 2 * d:]))
    assert rel(dpl, P.grad.reshape(-1, d)) < 2.5e-2, ("dpos", rel(dpl, P.grad.reshape(-1, d)))
    assert rel(du, U.grad) < 2.5e-2 and rel(dvb, V.grad) < 2.5e-2
    # rows of padded queries get no gradient
    assert (dqkv[:, :d].float().cpu() * (~valid)).abs().max().item() == 0.0
