"""HIP CTC loss vs the oracle (oracle/rnnt_ref.c::oracle_ctc_loss, itself pinned to ATen CPU ctc_loss)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,V,S", [(3, 12, 9, 5), (4, 50, 257, 20), (32, 376, 257, 105), (2, 30, 17, 40), (2, 9, 6, 0)])
def test_ctc_matches_oracle(B, T, V, S):
    from indic_cl_asr_amd.losses.ctc import CTCLoss
    from oracle import rnnt_oracle as orc
    g = torch.Generator().manual_seed(B * 7 + T)
    lp = torch.randn(B, T, V, generator=g).log_softmax(-1)
    tg = torch.randint(0, V - 1, (B, max(S, 1)), generator=g)[:, :S] if S > 0 else torch.zeros(B, 0, dtype=torch.long)
    if S >= 2:
        tg[0, 1] = tg[0, 0]  # repeated label needs a blank in between
    il = torch.randint(max(1, T // 2), T + 1, (B,), generator=g); il[0] = T
    tl = torch.randint(0, S + 1, (B,), generator=g)
    if S > 0:
        tl[0] = S
    if B > 1 and S > 3:
        il[1] = 3; tl[1] = min(S, 4)  # infeasible: more labels than frames -> inf -> zeroed
    nll_ref, grad_ref = orc.ctc_loss(lp.transpose(0, 1).contiguous().numpy(), tg.numpy(), il.numpy(), tl.numpy(), V - 1)
    grad_ref = np.transpose(grad_ref, (1, 0, 2))
    lpc = lp.cuda().requires_grad_(True)
    loss = CTCLoss(num_classes=V - 1, zero_infinity=True, reduction='none')
    loss._apply_reduction = False; loss._ctc_reduction = 'none'
    nll = loss(lpc, tg.cuda(), il.cuda(), tl.cuda())
    w = torch.linspace(0.5, 1.5, B)
    (nll * w.cuda()).sum().backward()
    assert np.allclose(nll.detach().cpu().numpy(), nll_ref, rtol=1e-5, atol=1e-4)
    # fp32 alpha+beta at |log-lik| ~ 2e3 (T = 376) carries ~1e-4 absolute error per posterior, as ATen's own fp32 kernels do
    atol = 2e-5 if T <= 100 else 4e-4
    assert np.allclose(lpc.grad.cpu().numpy(), grad_ref * w.numpy()[:, None, None], rtol=2e-3, atol=atol)


def test_ctc_head_hip_matches_aten_linear():
    """Single-language CTC head (decoder._CtcHeadHip: padded bf16 GEMM forward, library dgrad + ia_gemm_tn_bf16 for the
    weight and bias gradients) against F.linear on the same bf16-rounded operands."""
    import torch.nn.functional as F
    from indic_cl_asr_amd.decoder import _CtcHeadHip
    torch.manual_seed(3)
    B, T, d, V = 3, 70, 128, 257
    x = torch.randn(B, T, d, device="cuda")
    w = (torch.randn(V, d, device="cuda") * 0.1)
    b = torch.randn(V, device="cuda") * 0.1
    dy = torch.randn(B, T, V, device="cuda")
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
    ya = _CtcHeadHip.apply(xa, wa, ba)
    ya.backward(dy)
    xr, wr, br = (t.bfloat16().float().requires_grad_(True) for t in (x, w, b))
    br = b.clone().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    yr.backward(dy.bfloat16().float())
    def rel(a, r):
        return (a.float() - r).norm().item() / (r.norm().item() + 1e-12)
    assert rel(ya, yr) < 5e-3
    assert rel(xa.grad, xr.grad) < 1e-2 and rel(wa.grad, wr.grad) < 1e-2 and rel(ba.grad, br.grad) < 1e-2
