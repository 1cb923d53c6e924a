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
