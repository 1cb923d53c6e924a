"""csrc/block_train.hip (one trainable Conformer block as one C call forward + two C calls backward) against the per-op
Python autograd node it replaces (ops/block.py::_ConformerBlockFn): same kernels and dropout masks in the forward (equal
outputs), bf16 MFMA data-gradient GEMMs on transposed weight images instead of library GEMMs in the backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,heads,B,T,p", [(128, 2, 4, 96, 0.0), (256, 4, 3, 200, 0.1), (128, 2, 2, 33, 0.5)])
def test_native_block_matches_python_node(d, heads, B, T, p):
    from indic_cl_asr_amd.encoder import ConformerLayer
    from indic_cl_asr_amd.ops import block
    torch.manual_seed(d + T)
    layer = ConformerLayer(d, 4 * d, heads, 31, p, p).cuda().train()
    with torch.no_grad():
        layer.self_attn.pos_bias_u.normal_(0, 0.2); layer.self_attn.pos_bias_v.normal_(0, 0.2)
        layer.conv.batch_norm.weight.uniform_(0.5, 1.5); layer.conv.batch_norm.bias.normal_(0, 0.2)
    lens = torch.tensor([T] + [max(1, T - 7 * (i + 1)) for i in range(B - 1)], device="cuda")
    x = torch.randn(B * T, d, device="cuda")
    pe = block.pad_pos_emb(torch.randn(1, 2 * T - 1, d, device="cuda") * 0.5, d)
    R = torch.randn(B * T, d, device="cuda")
    res = []
    for native in (True, False):
        block.USE_NATIVE_BLOCKS = native
        try:
            for q in layer.parameters():
                q.grad = torch.full_like(q, 0.25)          # the backward ADDS into existing .grad buffers
            bn = layer.conv.batch_norm
            rm, rv, nb = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
            xg = x.clone().requires_grad_(True)
            out = block.conformer_block(xg, layer, lens, pe, B, T, 4321)
            (out * R).sum().backward()
            res.append((out.detach().clone(), xg.grad.clone(), {n: q.grad.clone() for n, q in layer.named_parameters()},
                        bn.running_var.clone()))
            bn.running_mean.copy_(rm); bn.running_var.copy_(rv); bn.num_batches_tracked.copy_(nb)
        finally:
            block.USE_NATIVE_BLOCKS = True
    (o1, dx1, g1, rv1), (o2, dx2, g2, rv2) = res
    assert (o1 - o2).abs().max().item() <= 1e-5 * max(1.0, o2.abs().max().item())        # same kernels, same masks
    assert torch.allclose(rv1, rv2, rtol=1e-4, atol=1e-6)
    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(dx1, dx2) < 2e-2, rel(dx1, dx2)
    for n in g1:
        a, b = g1[n] - 0.25, g2[n] - 0.25
        if n.endswith("depthwise_conv.bias") or n.endswith("linear_k.bias"):          # structurally zero gradients: noise
            assert a.abs().max().item() < 5e-2 * (1 + R.abs().max().item()), n
            continue
        assert rel(a, b) < 3e-2, (n, rel(a, b))


def test_weight_gradients_on_the_side_stream_change_nothing(monkeypatch):
    """csrc/block_train.hip can run the block's weight-gradient launches on an internal side stream beside the data-gradient chain
    (IA_WGRAD_SIDE=1; off by default: measured slower in the step)
    (forked behind their last operand, joined before an operand is overwritten and at the end of the call pair): same kernels on
    the same operands, so the gradients equal the single-stream order's (IA_WGRAD_SIDE=0; bit for bit wherever the launches are the same), at the bench's block size
    (where the launches really overlap) and with back-to-back blocks sharing the workspace."""
    from indic_cl_asr_amd.encoder import ConformerLayer
    from indic_cl_asr_amd.ops import block
    torch.manual_seed(5)
    d, heads, B, T = 256, 4, 32, 376
    layers = [ConformerLayer(d, 4 * d, heads, 31, 0.1, 0.1).cuda().train() for _ in range(2)]
    lens = torch.tensor([T] + [max(1, T - 9 * (i + 1)) for i in range(B - 1)], device="cuda")
    x = torch.randn(B * T, d, device="cuda")
    pe = block.pad_pos_emb(torch.randn(1, 2 * T - 1, d, device="cuda") * 0.5, d)
    R = torch.randn(B * T, d, device="cuda")
    res = []
    for side, merge in (("0", "0"), ("1", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("IA_WGRAD_SIDE", side)
        monkeypatch.setenv("IA_TN_MERGE", merge)     # (the default: all nine weight gradients of a block in one grouped launch)
        for layer in layers:
            for q in layer.parameters():
                q.grad = torch.full_like(q, 0.125)
            bn = layer.conv.batch_norm
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        xg = x.clone().requires_grad_(True)
        h = xg
        for i, layer in enumerate(layers):
            h = block.conformer_block(h, layer, lens, pe, B, T, 77 + i)
        (h * R).sum().backward()
        torch.cuda.synchronize()
        res.append([("dx", xg.grad.clone())] + [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate(layers)
                                                for n, q in layer.named_parameters()])
    assert all(torch.isfinite(t).all() for _, t in res[0])
    # two runs with the side stream: identical (a race would not repeat itself bit for bit)
    assert all(torch.equal(a, b) for (_, a), (_, b) in zip(res[1], res[2]))
    # against the single-stream order: the second call's four weight gradients go out as two grouped launches instead of one, so
    # their split-K partial sums are cut differently (fp32 rounding); everything else is the same launch on the same operands
    regrouped = ("linear_q", "linear_k", "linear_v", "linear_pos", "feed_forward1.linear")
    worst = 0.0
    for (n, a), (_, b) in zip(res[0], res[1]):
        if any(r in n for r in regrouped):
            e = ((a - b).norm() / (b.norm() + 1e-20)).item()
            worst = max(worst, e)
            assert e < 2e-6, (n, e)
        else:
            assert torch.equal(a, b), n
    print("regrouped weight gradients: worst relative L2 difference", worst)
    # one grouped launch for all nine projections of a block (fewer, longer splits): the data gradient is the same launch sequence,
    # the weight / bias gradients of the projections differ by fp32 rounding of differently cut partial sums
    proj = ("linear", "pointwise_conv")
    worst = 0.0
    for (n, a), (_, b) in zip(res[0], res[3]):
        if n != "dx" and any(r in n for r in proj):
            e = ((a - b).norm() / (b.norm() + 1e-20)).item()
            worst = max(worst, e)
            assert e < 2e-6, (n, e)
        else:
            assert torch.equal(a, b), n
    print("one launch per block: worst relative L2 difference", worst)


def test_native_block_refuses_a_second_backward():
    from indic_cl_asr_amd.encoder import ConformerLayer
    from indic_cl_asr_amd.ops import block
    torch.manual_seed(0)
    layer = ConformerLayer(128, 512, 2, 31, 0.0, 0.0).cuda().train()
    B, T = 2, 40
    lens = torch.tensor([40, 30], device="cuda")
    x = torch.randn(B * T, 128, device="cuda", requires_grad=True)
    pe = block.pad_pos_emb(torch.randn(1, 2 * T - 1, 128, device="cuda"), 128)
    out = block.conformer_block(x, layer, lens, pe, B, T, 1).sum()
    out.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        out.backward()


def test_layernorm_bwd_emits_the_dropout_scaled_bf16_gradient():
    """ia_layernorm_bwd_drop == ia_layernorm_bwd followed by ia_scale_dropout_bf16 on its result (same mask bits)."""
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    for N, d, p, alpha in ((777, 256, 0.1, 0.5), (130, 144, 0.0, 1.0), (64, 512, 0.25, 0.5)):
        g = torch.Generator().manual_seed(N + d)
        x = torch.randn(N, d, generator=g).cuda()
        dy = torch.randn(N, d, generator=g).cuda()
        dx_in = torch.randn(N, d, generator=g).cuda()
        gamma = (1 + 0.1 * torch.randn(d, generator=g)).cuda()
        scr = torch.empty(L.ia_layernorm_bwd_scratch_elems(N, d), device="cuda")
        dxa, dga, dba = torch.empty(N, d, device="cuda"), torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
        _lib.check(L.ia_layernorm_bwd(_lib.ptr(x), d, _lib.ptr(dy), None, d, N, d, _lib.ptr(gamma), 1e-5, _lib.ptr(dx_in), _lib.ptr(dxa), d,
                                      _lib.ptr(dga), _lib.ptr(dba), _lib.ptr(scr), _lib.stream_ptr()), "ia_layernorm_bwd")
        ha = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
        _lib.check(L.ia_scale_dropout_bf16(_lib.ptr(dxa), N, d, alpha, p, 77, _lib.ptr(ha), _lib.stream_ptr()), "ia_scale_dropout_bf16")
        dxb, dgb, dbb = torch.empty(N, d, device="cuda"), torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
        hb = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
        _lib.check(L.ia_layernorm_bwd_drop(_lib.ptr(x), d, _lib.ptr(dy), None, d, N, d, _lib.ptr(gamma), 1e-5, _lib.ptr(dx_in), _lib.ptr(dxb), d,
                                           _lib.ptr(dgb), _lib.ptr(dbb), alpha, p, 77, _lib.ptr(hb), d, _lib.ptr(scr), _lib.stream_ptr()),
                   "ia_layernorm_bwd_drop")
        torch.cuda.synchronize()
        assert torch.equal(dxa, dxb) and torch.equal(dga, dgb) and torch.equal(dba, dbb)
        assert torch.equal(ha, hb)


def test_partials_finish_multi_sums_several_partial_sets_in_one_launch():
    """ia_partials_finish_multi: the column sums of several partial-row sets (a block's five LayerNorm gradients)."""
    import ctypes
    from indic_cl_asr_amd import _lib
    L = _lib.lib()

    class Job(ctypes.Structure):
        _fields_ = [("part", ctypes.c_void_p), ("G", ctypes.c_int), ("C", ctypes.c_int), ("C0", ctypes.c_int),
                    ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p)]

    g = torch.Generator().manual_seed(0)
    shapes = [(300, 512, 256), (7, 288, 144), (1024, 512, 256), (33, 64, 64)]
    parts = [torch.randn(G, C, generator=g).cuda() for G, C, _ in shapes]
    out0 = [torch.empty(C0, device="cuda") for _, _, C0 in shapes]
    out1 = [torch.empty(max(C - C0, 1), device="cuda") for _, C, C0 in shapes]
    jobs = (Job * len(shapes))()
    for i, (G, C, C0) in enumerate(shapes):
        jobs[i] = Job(parts[i].data_ptr(), G, C, C0, out0[i].data_ptr(), out1[i].data_ptr())
    _lib.check(L.ia_partials_finish_multi(ctypes.addressof(jobs), len(shapes), _lib.stream_ptr()), "ia_partials_finish_multi")
    torch.cuda.synchronize()
    for i, (G, C, C0) in enumerate(shapes):
        ref = parts[i].double().sum(0)
        assert torch.allclose(out0[i].double(), ref[:C0], rtol=1e-5, atol=1e-4)
        if C > C0:
            assert torch.allclose(out1[i][:C - C0].double(), ref[C0:], rtol=1e-5, atol=1e-4)
