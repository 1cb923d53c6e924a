"""csrc/tail_ops.hip + the logits form of csrc/ctc.hip against the ATen compositions they replace (ops/tail.py):
language-row selection and its scatter-add backward, CTC head + loss on raw logits (conv_asr.py:459-490 + A/losses/ctc.py:68-82)
with all three gradients, the loss combination (hybrid_rnnt_ctc_models.py:899-913), the prediction network's SOS + embedding
input with its deterministic backward, multi-tensor axpy, 16-bit multi-transpose, the time-major <-> batch-major cast and
the f16 output of the projection GEMM."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _L():
    from indic_cl_asr_amd import _lib
    return _lib, _lib.lib()


def test_select_rows_cast_and_scatter_add():
    _lib, L = _L()
    g = torch.Generator().manual_seed(1)
    n, K, row0, nrows, extra, rows_out, ldt = 700, 200, 256, 130, 699, 136, 144
    W = torch.randn(n, K, generator=g).cuda(); b = torch.randn(n, generator=g).cuda()
    for f16 in (0, 1):
        dt = torch.float16 if f16 else torch.bfloat16
        out = torch.full((rows_out, K), 7.0, dtype=dt, device="cuda")
        outT = torch.full((K, ldt), 7.0, dtype=dt, device="cuda")
        bo = torch.full((rows_out,), 7.0, device="cuda")
        _lib.check(L.ia_select_rows_cast(_lib.ptr(W), K, _lib.ptr(b), row0, nrows, extra, K, rows_out, 0.5, f16, _lib.ptr(out), _lib.ptr(outT),
                                         ldt, _lib.ptr(bo), _lib.stream_ptr()), "sel")
        idx = torch.cat([torch.arange(row0, row0 + nrows), torch.tensor([extra])]).cuda()
        ref = torch.zeros(rows_out, K, device="cuda"); ref[:nrows + 1] = W[idx] * 0.5
        assert torch.equal(out, ref.to(dt))
        refT = torch.zeros(K, ldt, device="cuda"); refT[:, :rows_out] = ref.t()
        assert torch.equal(outT, refT.to(dt))
        rb = torch.zeros(rows_out, device="cuda"); rb[:nrows + 1] = b[idx]
        assert torch.equal(bo, rb)
    dst = torch.randn(n, K, generator=g).cuda(); dstb = torch.randn(n, generator=g).cuda()
    src = torch.randn(rows_out, K, generator=g).cuda(); srcb = torch.randn(rows_out, generator=g).cuda()
    want, wantb = dst.clone(), dstb.clone()
    want[idx] += 0.25 * src[:nrows + 1]; wantb[idx] += 0.25 * srcb[:nrows + 1]
    _lib.check(L.ia_rows_scatter_add(_lib.ptr(dst), K, _lib.ptr(src), K, row0, nrows, extra, K, 0.25, _lib.ptr(dstb), _lib.ptr(srcb),
                                     _lib.stream_ptr()), "scat")
    assert torch.allclose(dst, want, rtol=0, atol=1e-6) and torch.allclose(dstb, wantb, rtol=0, atol=1e-6)


@pytest.mark.parametrize("B,T,d,S", [(3, 50, 64, 7), (32, 376, 256, 105), (2, 9, 144, 1)])
def test_ctc_head_loss_on_logits_matches_linear_logsoftmax_ctc(B, T, d, S):
    """nll and d/dx, d/dW, d/db of the fused node against F.linear(bf16-rounded operands) -> log_softmax -> F.ctc_loss in fp64."""
    from indic_cl_asr_amd.ops import tail
    g = torch.Generator().manual_seed(B * 1000 + T)
    n_lang, v = 3, 16
    n = n_lang * v + 1
    x = (torch.randn(B, T, d, generator=g) * 0.5).cuda().requires_grad_(True)
    W = torch.nn.Parameter((torch.randn(n, d, 1, generator=g) * 0.2).cuda()); b = torch.nn.Parameter((torch.randn(n, generator=g) * 0.1).cuda())
    il = torch.randint(max(1, T // 2), T + 1, (B,), generator=g); il[0] = T
    tl = torch.tensor([min(S, max(0, int(il[i]) // 3)) for i in range(B)]); tl[0] = min(S, int(il[0]) // 3)
    tg = torch.randint(0, v, (B, S), generator=g)
    lang = 1
    tail.DIRECT_ACCUMULATE = False
    try:
        keep = {}
        nll = tail.ctc_head_loss(x, W, b, tg.cuda(), il.cuda(), tl.cuda(), lang * v, v, n - 1, blank=v, zero_infinity=True, keep=keep)
        wts = torch.randn(B, generator=g).cuda()
        (nll * wts).sum().backward()
    finally:
        tail.DIRECT_ACCUMULATE = True
    rows = torch.cat([torch.arange(lang * v, (lang + 1) * v), torch.tensor([n - 1])])
    xr = x.detach().bfloat16().double().cpu().requires_grad_(True)
    Wr = W.detach().view(n, d).bfloat16().double().cpu().requires_grad_(True); br = b.detach().double().cpu().requires_grad_(True)
    lp = F.log_softmax(F.linear(xr, Wr[rows], br[rows]), -1)
    ref = F.ctc_loss(lp.transpose(0, 1), tg, il, tl, blank=v, reduction='none', zero_infinity=True)
    (ref * wts.double().cpu()).sum().backward()
    assert torch.allclose(nll.double().cpu(), ref, rtol=2e-4, atol=2e-3), (nll, ref)
    assert keep["logits"].shape[:2] == (B, T) and keep["V"] == v + 1

    def rel(a, r):
        return float((a.double().cpu() - r).norm() / (r.norm() + 1e-30))
    assert rel(x.grad, xr.grad) <= 2e-2               # bf16 gradient operand (softmax - occupancy) and bf16 W in dX = dY W
    gW = torch.zeros(n, d, dtype=torch.double); gW += Wr.grad
    assert rel(W.grad.view(n, d), gW) <= 2e-2
    assert rel(b.grad, br.grad) <= 2e-2
    others = torch.ones(n, dtype=torch.bool); others[rows] = False
    assert float(W.grad.view(n, d)[others.cuda()].abs().max()) == 0.0 and float(b.grad[others.cuda()].abs().max()) == 0.0


def test_ctc_head_loss_direct_accumulation_adds_into_existing_grads():
    from indic_cl_asr_amd.ops import tail
    g = torch.Generator().manual_seed(5)
    B, T, d, v = 2, 20, 32, 8
    n = 2 * v + 1
    x = torch.randn(B, T, d, generator=g).cuda().requires_grad_(True)
    W = torch.nn.Parameter(torch.randn(n, d, 1, generator=g).cuda() * 0.3); b = torch.nn.Parameter(torch.randn(n, generator=g).cuda() * 0.1)
    il, tl, tg = torch.tensor([20, 13]).cuda(), torch.tensor([4, 2]).cuda(), torch.randint(0, v, (B, 4), generator=g).cuda()
    tail.DIRECT_ACCUMULATE = False
    tail.ctc_head_loss(x, W, b, tg, il, tl, 0, v, n - 1, blank=v).sum().backward()
    gW, gb = W.grad.clone(), b.grad.clone()
    tail.DIRECT_ACCUMULATE = True
    W.grad = torch.full_like(W, 0.5); b.grad = torch.full_like(b, -0.25)          # a pre-loaded penalty (EWC: set_grads)
    tail.ctc_head_loss(x, W, b, tg, il, tl, 0, v, n - 1, blank=v).sum().backward()
    assert torch.allclose(W.grad, gW + 0.5, atol=1e-6) and torch.allclose(b.grad, gb - 0.25, atol=1e-6)


def test_loss_combine_values_gradients_and_flags():
    from indic_cl_asr_amd.ops import tail
    g = torch.Generator().manual_seed(2)
    B, w = 37, 0.3
    c = (torch.rand(B, generator=g) * 100).cuda().requires_grad_(True); k = (torch.rand(B, generator=g) * 50).cuda().requires_grad_(True)
    flags = [torch.tensor(0, dtype=torch.int32, device="cuda"), torch.tensor(5, dtype=torch.int32, device="cuda")]
    total, vals = tail.loss_combine(c, k, w, flags)
    (total * 2.0).backward()
    r, q = c.detach().double().mean().item(), k.detach().double().mean().item()
    assert math.isclose(vals[0].item(), r, rel_tol=1e-6) and math.isclose(vals[1].item(), q, rel_tol=1e-6)
    assert math.isclose(vals[2].item(), (1 - w) * r + w * q, rel_tol=1e-6) and math.isclose(total.item(), vals[2].item(), rel_tol=0)
    assert vals[3].item() == 1.0
    assert torch.allclose(c.grad, torch.full_like(c, 2.0 * (1 - w) / B)) and torch.allclose(k.grad, torch.full_like(k, 2.0 * w / B))


@pytest.mark.parametrize("B,U,H,n_rows", [(4, 9, 64, 50), (32, 105, 640, 5633)])
def test_embed_sos_forward_and_deterministic_backward(B, U, H, n_rows):
    from indic_cl_asr_amd.ops import tail
    g = torch.Generator().manual_seed(B + U)
    E = torch.nn.Parameter(torch.randn(n_rows, H, generator=g).cuda())
    tok = torch.randint(0, min(n_rows - 1, 256), (B, U), generator=g).cuda()
    tail.DIRECT_ACCUMULATE = False
    try:
        x = tail.embed_sos(E, tok, pad_row=n_rows - 1)
        ref = torch.cat([torch.zeros(B, 1, H, device="cuda"), F.embedding(tok, E.detach())], 1).transpose(0, 1)     # [U+1, B, H]
        assert x.shape == (U + 1, B, H) and x.dtype == torch.bfloat16 and torch.equal(x, ref.bfloat16())
        gy = torch.randn(U + 1, B, H, generator=g).cuda().bfloat16()
        x.backward(gy)
        g1 = E.grad.clone()
        E.grad = None
        x2 = tail.embed_sos(E, tok, pad_row=n_rows - 1)
        x2.backward(gy)
        assert torch.equal(E.grad, g1)                                        # bit-reproducible (ordered sums, no atomics)
    finally:
        tail.DIRECT_ACCUMULATE = True
    Er = E.detach().double().requires_grad_(True)
    F.embedding(tok, Er, padding_idx=n_rows - 1).backward(gy[1:].transpose(0, 1).double())
    assert torch.allclose(g1.double(), Er.grad, rtol=1e-5, atol=1e-4)


def test_multi_axpy_transpose16_swap01_and_f16_gemm_output():
    from indic_cl_asr_amd.ops import fast, tail
    g = torch.Generator().manual_seed(9)
    pairs, want = [], []
    for i, n in enumerate((7, 64, 1000, 65536 + 3, 300000)):
        d, s = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
        sc = 0.5 + i
        want.append(d + sc * s); pairs.append((d, s, sc))
    tail.multi_axpy(pairs)
    for (d, _, _), w in zip(pairs, want):
        assert torch.allclose(d, w, rtol=1e-6, atol=1e-6)
    mats = [torch.randn(r, c, generator=g).cuda().to(dt) for (r, c), dt in zip([(640, 256), (2560, 640), (8, 8), (72, 200)],
                                                                              [torch.bfloat16, torch.bfloat16, torch.float16, torch.float16])]
    for m, t in zip(mats, fast.transpose16_multi(mats)):
        assert torch.equal(t, m.t().contiguous())
    x = torch.randn(11, 5, 48, generator=g).cuda()
    assert torch.equal(fast.swap01_cast(x, torch.bfloat16), x.transpose(0, 1).contiguous().bfloat16())
    xb = x.bfloat16()
    assert torch.equal(fast.swap01_cast(xb, torch.float32), xb.transpose(0, 1).contiguous().float())
    a = (torch.randn(300, 64, generator=g) * 0.5).cuda().bfloat16(); w = (torch.randn(72, 64, generator=g) * 0.3).cuda().bfloat16()
    bias = torch.randn(72, generator=g).cuda()
    _, h = fast.gemm(a, w, bias, out_f16=True)
    ref = a.double() @ w.double().t() + bias.double()
    assert h.dtype == torch.float16 and torch.allclose(h.double(), ref, rtol=2e-3, atol=2e-3)
