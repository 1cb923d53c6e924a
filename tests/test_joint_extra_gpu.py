"""The MAS / LwF terms on the fused joint's f16 lattice (csrc/joint_extra.hip, ops.joint.LatticeStash) against an fp64
restatement of the reference's arithmetic on the per-sub-batch stash tensors:
  MAS   mean_sub( mean_cells( sum_v z^2 ) )                          R/cl_baseline_mas.py:258-265
  LwF   mean_sub( F.kl_div(z_sub, exp(t_sub), 'batchmean') )         R/cl_baseline_lwf.py:242-257
with each sub-batch narrowed to its own max T / max U+1 (A/modules/rnnt.py:1436-1447) -- padded cells inside the box count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, T, U1, H, V, seed):
    g0 = torch.Generator().manual_seed(seed)
    f = torch.randn(B, T, H, generator=g0) * 0.7
    g = torch.randn(B, U1, H, generator=g0) * 0.7
    W = torch.randn(V, H, generator=g0) * 0.15
    b = torch.randn(V, generator=g0) * 0.1
    labels = torch.randint(0, V - 1, (B, U1 - 1), generator=g0)
    fl = torch.randint(max(1, T // 2), T + 1, (B,), generator=g0); fl[0] = T
    gl = torch.randint(0, U1, (B,), generator=g0); gl[-1] = U1 - 1
    return f, g, W, b, labels, fl, gl


def _logits64(f, g, W, b):
    """(f16-rounded logits as float64 leaf-connected graph, pre-activation) -- same quantisation as tests/test_joint_gpu.py."""
    pre = (f.half()[:, :, None, :] + g.half()[:, None, :, :])
    hid = torch.relu(pre).double()
    z = hid @ W.half().double().t() + b.double()
    return z, hid, pre


def _sub_terms(z, t, fl, gl, sub, V):
    """The reference's loops over the stash lists, fp64."""
    B = z.shape[0]
    n_sub = (B + sub - 1) // sub
    sq, kd = 0.0, 0.0
    for b0 in range(0, B, sub):
        b1 = min(b0 + sub, B)
        mt, mu1 = int(fl[b0:b1].max()), int(gl[b0:b1].max()) + 1
        zs = z[b0:b1, :mt, :mu1]
        sq = sq + (zs.flatten(end_dim=-2) ** 2).sum(-1).mean()
        if t is not None:
            ts = t[b0:b1, :mt, :mu1]
            kd = kd + torch.nn.functional.kl_div(zs, ts.exp(), reduction='batchmean')
    return sq / n_sub, kd / n_sub


def _close(a, ref, what, rel=6e-3):
    a, ref = a.detach().cpu().double(), ref.double()
    tol = rel * ref.abs().max().item() + 1e-9
    assert (a - ref).abs().max().item() <= tol, (what, (a - ref).abs().max().item(), ref.abs().max().item())


@pytest.mark.parametrize("B,T,U1,H,V,sub", [(5, 37, 19, 320, 257, 2), (4, 50, 33, 320, 130, 4), (3, 21, 9, 640, 257, 1)])
def test_importance_term_and_its_gradient_match_the_stash_loop(B, T, U1, H, V, sub):
    from indic_cl_asr_amd.ops import joint as J
    f, g, W, b, labels, fl, gl = _inputs(B, T, U1, H, V, seed=B * 31 + T)
    # fp64 reference with autograd through the f16-quantised operands
    leaves = [x.clone().double().requires_grad_(True) for x in (f.half(), g.half(), W.half(), b)]
    pre = leaves[0][:, :, None, :] + leaves[1][:, None, :, :]
    pre = pre + ((f.half()[:, :, None, :] + g.half()[:, None, :, :]).double() - pre.detach())   # the kernel adds in f16
    z = torch.relu(pre) @ leaves[2].t() + leaves[3]
    z = z + (z.detach().half().double() - z.detach())    # value rounded to f16 as the kernel stores it, gradient straight through
    sq_ref, _ = _sub_terms(z, None, fl, gl, sub, V)
    (0.7 * sq_ref).backward()
    fc, gc, Wc, bc = (t.cuda().requires_grad_(True) for t in (f, g, W, b))
    req = {"sub": sub, "h_enc": fl.tolist(), "h_tgt": gl.tolist(), "detach": False}
    costs = J.fused_joint_rnnt(fc, gc, Wc, bc, labels.cuda(), fl.cuda(), gl.cuda(), V - 1, scale_hint=0.25, stash_req=req)
    stash = req["out"]
    assert len(stash) == (B + sub - 1) // sub
    subs = list(stash)     # the reference's list view of the stash
    assert subs[0].shape == (min(sub, B), int(fl[:sub].max()), int(gl[:sub].max()) + 1, V)
    assert torch.allclose(subs[0].float().cpu(), z.detach()[:sub, :subs[0].shape[1], :subs[0].shape[2]].float(), atol=2e-3, rtol=2e-3)
    sq = J.lattice_sumsq_term(stash)
    assert abs(sq.item() - sq_ref.item()) <= 1e-3 * abs(sq_ref.item())
    (0.7 * sq).backward()     # the transducer costs are not part of this loss: its gradient kernel must not run
    assert stash.logits is None
    with pytest.raises(RuntimeError, match="overwritten"):
        J.lattice_sumsq_term(stash)
    for got, ref, what in zip((fc, gc, Wc, bc), leaves, ("df", "dg", "dW", "dbias")):
        _close(got.grad, ref.grad, what)


@pytest.mark.parametrize("B,T,U1,H,V,sub", [(4, 33, 17, 320, 257, 2), (3, 45, 21, 640, 100, 4)])
def test_distillation_term_adds_to_the_transducer_gradient(B, T, U1, H, V, sub):
    from indic_cl_asr_amd.ops import joint as J
    from oracle import rnnt_oracle as orc
    f, g, W, b, labels, fl, gl = _inputs(B, T, U1, H, V, seed=B * 17 + U1)
    ft, gt, Wt, bt, *_ = _inputs(B, T, U1, H, V, seed=B * 17 + U1 + 1)
    ft, gt, Wt, bt = f + 0.1 * ft, g + 0.1 * gt, W + 0.02 * Wt, b + 0.02 * bt       # a teacher near the student
    with torch.no_grad():
        treq = {"sub": sub, "h_enc": fl.tolist(), "h_tgt": gl.tolist(), "detach": True}
        J.fused_joint_rnnt(ft.cuda(), gt.cuda(), Wt.cuda(), bt.cuda(), labels.cuda(), fl.cuda(), gl.cuda(), V - 1, stash_req=treq)
    teacher = treq["out"]
    t64, _, _ = _logits64(ft, gt, Wt, bt)
    t64 = t64.half().double()
    leaves = [x.clone().double().requires_grad_(True) for x in (f.half(), g.half(), W.half(), b)]
    pre = leaves[0][:, :, None, :] + leaves[1][:, None, :, :]
    pre = pre + ((f.half()[:, :, None, :] + g.half()[:, None, :, :]).double() - pre.detach())   # the kernel adds in f16
    hid = torch.relu(pre)
    z = hid @ leaves[2].t() + leaves[3]
    z = z + (z.detach().half().double() - z.detach())
    _, kd_ref = _sub_terms(z, t64, fl, gl, sub, V)
    wts = torch.tensor([0.2, 0.15, 0.25, 0.1][:B])
    r = orc.rnnt_loss(z.detach().float().numpy(), labels.numpy(), fl.numpy(), gl.numpy(), V - 1)
    Grn = torch.from_numpy(r["grads"]).double() * wts.double().view(-1, 1, 1, 1)
    (0.3 * kd_ref + (z * Grn).sum()).backward()         # d/dz [costs . wts] = Grn: inject it through a linear term
    fc, gc, Wc, bc = (t.cuda().requires_grad_(True) for t in (f, g, W, b))
    req = {"sub": sub, "h_enc": fl.tolist(), "h_tgt": gl.tolist(), "detach": False}
    costs = J.fused_joint_rnnt(fc, gc, Wc, bc, labels.cuda(), fl.cuda(), gl.cuda(), V - 1, scale_hint=0.25, stash_req=req)
    kd = J.lattice_kd_term(req["out"], teacher)
    assert np.allclose(costs.detach().cpu().numpy(), r["costs"], rtol=2e-4, atol=2e-3)
    assert abs(kd.item() - kd_ref.item()) <= 2e-3 * abs(kd_ref.item()) + 1e-4, (kd.item(), kd_ref.item())
    ((costs * wts.cuda()).sum() + 0.3 * kd).backward()
    for got, ref, what in zip((fc, gc, Wc, bc), leaves, ("df", "dg", "dW", "dbias")):
        _close(got.grad, ref.grad, what)


def test_stash_on_the_model_path_is_a_lattice_and_unused_stash_costs_nothing():
    """joint.store_list after a fused forward with store_sub_logits set; a backward that never touches it equals the plain step."""
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel
    from indic_cl_asr_amd.ops.joint import LatticeStash
    torch.manual_seed(0)
    cfg = model_config('tiny', d_model=64, n_layers=2, n_heads=1, pred_hidden=64, joint_hidden=320, vocab_per_lang=64,
                       compute_dtype="bf16")
    m = EncDecHybridRNNTCTCModel(cfg).cuda()
    m.train()
    sig = torch.randn(3, 16000).cuda() * 0.1
    sl = torch.tensor([16000, 12000, 8000]).cuda()
    tok = torch.randint(0, cfg.vocab_per_lang, (3, 6)).cuda()
    tl = torch.tensor([6, 4, 5]).cuda()
    m.joint.store_sub_logits = True
    m.ctc_decoder.return_logits_ = True
    loss, _ = m.training_step((sig, sl, tok, tl), [cfg.languages[0]] * 3)
    assert isinstance(m.joint.store_list, LatticeStash)
    imp = cl.mas_importance_loss(m, 0.3)
    imp.backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert m.joint.joint_net[-1][cfg.languages[0]].weight.grad.abs().sum().item() > 0
