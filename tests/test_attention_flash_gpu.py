"""csrc/attention_flash.hip (rel-pos attention forward, key-tile loop + online softmax: any T, head dim <= 64) against an
fp64 restatement of RelPositionMultiHeadAttention.forward (multi_head_attention.py:197-250) on the kernel's bf16 inputs,
and against the all-keys-in-registers kernel (csrc/attention.hip) where that one applies."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, T, H, dk, seed=0, lens=None):
    g = torch.Generator().manual_seed(seed)
    d = H * dk
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.8).bfloat16().cuda()
    pl = (torch.randn(2 * T - 1, d, generator=g) * 0.8).bfloat16().cuda()
    bu = (torch.randn(H, dk, generator=g) * 0.3).cuda()
    bv = (torch.randn(H, dk, generator=g) * 0.3).cuda()
    if lens is None:
        lens = [T] + [int(torch.randint(max(1, T // 3), T + 1, (1,), generator=g)) for _ in range(B - 1)]
    return qkv, pl, bu, bv, torch.tensor(lens, dtype=torch.int64).cuda()


def _reference(qkv, pl, bu, bv, lens, B, T, H, dk):
    d = H * dk
    x = qkv.double().view(B, T, 3, H, dk)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)   # [B,H,T,dk]
    p = pl.double()[:2 * T - 1].view(2 * T - 1, H, dk).permute(1, 0, 2)                            # [H,2T-1,dk]
    qu = (q + bu.double()[None, :, None, :]).bfloat16().double()                                   # the kernel's operand rounding
    qv = (q + bv.double()[None, :, None, :]).bfloat16().double()
    ac = qu @ k.transpose(-1, -2)
    full = torch.einsum("bhid,hrd->bhir", qv, p)                                                   # [B,H,T,2T-1]
    i = torch.arange(T, device=qkv.device)[:, None]; j = torch.arange(T, device=qkv.device)[None, :]
    bd = full.gather(-1, (T - 1 - i + j).expand(B, H, T, T))                                       # rel_shift as index arithmetic
    s = (ac + bd) / math.sqrt(dk)
    valid = j[None] < lens[:, None, None]                                                          # [B,1,T] keys
    s = s.masked_fill(~valid[:, None], float("-inf"))
    pr = torch.softmax(s, -1)
    o = pr @ v
    qvalid = (torch.arange(T, device=qkv.device)[None, :] < lens[:, None])[:, None, :, None]
    o = o * qvalid
    return o.transpose(1, 2).reshape(B * T, d)


@pytest.mark.parametrize("B,T,H,dk,lens", [(3, 100, 2, 64, None), (2, 376, 4, 64, None), (2, 751, 8, 64, [751, 500]),
                                           (3, 126, 4, 36, [126, 64, 5]), (2, 64, 1, 64, [64, 63]), (1, 65, 2, 48, [65]),
                                           (4, 200, 2, 64, [128, 200, 1, 129])])
def test_flash_attention_matches_fp64_restatement(B, T, H, dk, lens):
    from indic_cl_asr_amd.ops import fast
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=T + dk, lens=lens)
    ref = _reference(qkv, pl, bu, bv, ln, B, T, H, dk)
    out = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk).double()
    err = (out - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, (err, ref.abs().max().item())   # bf16 band, bf16 probabilities
    # padded queries are exactly zero
    for b in range(B):
        n = int(ln[b])
        if n < T:
            assert out.view(B, T, -1)[b, n:].abs().max().item() == 0.0


def test_flash_attention_agrees_with_the_register_resident_kernel():
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 4, 376, 4, 64
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=3)
    a = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk).float()
    b = fast.relpos_attention(qkv, pl, bu, bv, ln, B, T, H, dk).float()
    assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()


def test_flash_attention_dropout_is_unbiased_and_deterministic():
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 2, 256, 2, 64
    qkv, pl, bu, bv, ln = _inputs(B, T, H, dk, seed=9, lens=[256, 256])
    base = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk).float()
    d1 = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, dropout_p=0.25, seed=5).float()
    d2 = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, dropout_p=0.25, seed=5).float()
    d3 = fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, dropout_p=0.25, seed=6).float()
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    assert (d1 - base).abs().max().item() > 1e-3                       # the mask does something
    # inverted dropout keeps the expectation: averaged over many seeds the output returns to the undropped one
    acc = torch.zeros_like(base)
    n = 24
    for s in range(n):
        acc += fast.relpos_attention_flash(qkv, pl, bu, bv, ln, B, T, H, dk, dropout_p=0.25, seed=100 + s).float()
    rel = ((acc / n - base).norm() / base.norm()).item()
    single = ((d1 - base).norm() / base.norm()).item()
    assert rel < 0.35 * single, (rel, single)                           # ~ 1/sqrt(24) of one draw's deviation
