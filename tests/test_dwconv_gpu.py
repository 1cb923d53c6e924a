"""Depthwise-conv kernels (csrc/dwconv.hip) through the C ABI against ATen conv1d in fp32
(CausalConv1D as configured by the Conformer convolution module, causal_convs.py:72-150)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _scratch(L, B, T, d, k):
    return torch.empty(max(1, L.ia_dwconv_scratch_elems(B, T, d, k)), dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("B,T,d,k", [(3, 50, 32, 9), (2, 77, 256, 31), (4, 33, 80, 5), (2, 40, 192, 17), (1, 7, 64, 31),
                                     (32, 376, 256, 31)])
def test_dwconv_time_forward_dgrad_wgrad(B, T, d, k):
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    torch.manual_seed(B * 1000 + T + d + k)
    x = torch.randn(B, T, d, device="cuda")
    w = torch.randn(d, k, device="cuda") * 0.3
    b = torch.randn(d, device="cuda")
    dy = torch.randn(B, T, d, device="cuda")
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yr = F.conv1d(xr.transpose(1, 2), wr.unsqueeze(1), br, padding=(k - 1) // 2, groups=d).transpose(1, 2)
    yr.backward(dy)
    y = torch.empty_like(x)
    _lib.check(L.ia_dwconv_time(_lib.ptr(x), B, T, d, k, _lib.ptr(w), _lib.ptr(b), 0, _lib.ptr(y), _lib.stream_ptr()), "fwd")
    assert torch.allclose(y, yr, rtol=1e-4, atol=1e-4)
    dx = torch.empty_like(x)
    _lib.check(L.ia_dwconv_time(_lib.ptr(dy), B, T, d, k, _lib.ptr(w), None, 1, _lib.ptr(dx), _lib.stream_ptr()), "dgrad")
    assert torch.allclose(dx, xr.grad, rtol=1e-4, atol=1e-4)
    dw = torch.full((d, k), float("nan"), device="cuda"); db = torch.full((d,), float("nan"), device="cuda")
    _lib.check(L.ia_dwconv_time_wgrad(_lib.ptr(x), _lib.ptr(dy), B, T, d, k, _lib.ptr(dw), _lib.ptr(db),
                                      _lib.ptr(_scratch(L, B, T, d, k)), _lib.stream_ptr()), "wgrad")
    scale = wr.grad.abs().max().item()
    assert torch.allclose(dw, wr.grad.view(d, k), rtol=1e-4, atol=1e-5 * scale + 1e-4)
    assert torch.allclose(db, br.grad, rtol=1e-4, atol=1e-5 * br.grad.abs().max().item() + 1e-4)


@pytest.mark.parametrize("B,T,d,k", [(3, 50, 64, 9), (4, 101, 256, 31), (2, 20, 96, 15)])
def test_glu_dwconv_masks_padding_and_returns_batchnorm_sums(B, T, d, k):
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    torch.manual_seed(T)
    x2 = torch.randn(B, T, 2 * d, device="cuda").bfloat16()
    lens = torch.randint(1, T + 1, (B,), device="cuda"); lens[0] = T
    w = torch.randn(d, k, device="cuda") * 0.3
    b = torch.randn(d, device="cuda")
    g = F.glu(x2.float(), dim=-1) * (torch.arange(T, device="cuda")[None, :] < lens[:, None]).unsqueeze(-1)
    zr = F.conv1d(g.transpose(1, 2), w.unsqueeze(1), b, padding=(k - 1) // 2, groups=d).transpose(1, 2)
    z = torch.empty(B, T, d, device="cuda")
    sums = torch.full((2, d), float("nan"), device="cuda")
    _lib.check(L.ia_glu_dwconv(_lib.ptr(x2), _lib.ptr(lens), B, T, d, k, _lib.ptr(w), _lib.ptr(b), _lib.ptr(z),
                               _lib.ptr(sums[0]), _lib.ptr(sums[1]), _lib.ptr(_scratch(L, B, T, d, k)), _lib.stream_ptr()), "glu")
    assert torch.allclose(z, zr, rtol=1e-3, atol=1e-3)
    assert torch.allclose(sums[0], zr.sum((0, 1)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(sums[1], (zr * zr).sum((0, 1)), rtol=1e-3, atol=1e-2)


def test_glu_dwconv_fixed_point_sums_match_the_partial_row_sums():
    """ia_glu_dwconv_fixed: same z, BatchNorm sums in 2^-24 fixed point (integer atomics) == the fp32 partial-row sums."""
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    for B, T, d, ksz in ((4, 200, 256, 31), (2, 77, 144, 9)):
        g = torch.Generator().manual_seed(B * T)
        x2 = torch.randn(B * T, 2 * d, generator=g).bfloat16().cuda()
        lens = torch.tensor([T] + [max(1, T - 13 * (i + 1)) for i in range(B - 1)], dtype=torch.long).cuda()
        w = (torch.randn(d, ksz, generator=g) * 0.2).cuda()
        bias = (torch.randn(d, generator=g) * 0.1).cuda()
        za, zb = torch.empty(B * T, d, device="cuda"), torch.empty(B * T, d, device="cuda")
        sums = torch.empty(2 * d, device="cuda")
        scr = torch.empty(L.ia_dwconv_scratch_elems(B, T, d, ksz), device="cuda")
        _lib.check(L.ia_glu_dwconv(_lib.ptr(x2), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(za), _lib.ptr(sums[:d]),
                                   _lib.ptr(sums[d:]), _lib.ptr(scr), _lib.stream_ptr()), "ia_glu_dwconv")
        for _ in range(2):      # twice: the accumulators must give the same bits run to run
            acc = torch.zeros(8, 2 * d, dtype=torch.int64, device="cuda")
            _lib.check(L.ia_glu_dwconv_fixed(_lib.ptr(x2), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(zb),
                                             _lib.ptr(acc), _lib.stream_ptr()), "ia_glu_dwconv_fixed")
            torch.cuda.synchronize()
            if _ == 0:
                first = acc.clone()
            else:
                assert torch.equal(acc, first)
        assert torch.equal(za, zb)
        got = acc.sum(0).double() / 2 ** 24
        assert torch.allclose(got, sums.double(), rtol=1e-5, atol=1e-4)


def test_glu_dwconv_backward_in_two_launches_matches_the_four_launch_sequence():
    """ia_dwconv_glu_bwd / ia_dwconv_glu_wgrad == ia_dwconv_time(flip) + ia_glu_bwd / ia_glu_mask + ia_dwconv_time_wgrad."""
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    for B, T, d, ksz in ((3, 150, 128, 31), (2, 70, 144, 9)):
        g = torch.Generator().manual_seed(B + T)
        c2 = torch.randn(B * T, 2 * d, generator=g).bfloat16().cuda()
        dz = torch.randn(B * T, d, generator=g).cuda()
        lens = torch.tensor([T, max(1, T - 40), 7][:B], dtype=torch.long).cuda()
        w = (torch.randn(d, ksz, generator=g) * 0.2).cuda()
        scr = torch.empty(L.ia_dwconv_scratch_elems(B, T, d, ksz), device="cuda")
        dG, Gm = torch.empty(B * T, d, device="cuda"), torch.empty(B * T, d, device="cuda")
        dc2a, dc2b = (torch.empty(B * T, 2 * d, dtype=torch.bfloat16, device="cuda") for _ in range(2))
        dwa, dba, dwb, dbb = (torch.empty(*s, device="cuda") for s in ((d, ksz), (d,), (d, ksz), (d,)))
        _lib.check(L.ia_dwconv_time(_lib.ptr(dz), B, T, d, ksz, _lib.ptr(w), None, 1, _lib.ptr(dG), _lib.stream_ptr()), "dwconv_time")
        _lib.check(L.ia_glu_mask(_lib.ptr(c2), _lib.ptr(lens), B, T, d, _lib.ptr(Gm), _lib.stream_ptr()), "glu_mask")
        _lib.check(L.ia_dwconv_time_wgrad(_lib.ptr(Gm), _lib.ptr(dz), B, T, d, ksz, _lib.ptr(dwa), _lib.ptr(dba), _lib.ptr(scr),
                                          _lib.stream_ptr()), "wgrad")
        _lib.check(L.ia_glu_bwd(_lib.ptr(c2), _lib.ptr(dG), _lib.ptr(lens), B, T, d, _lib.ptr(dc2a), _lib.stream_ptr()), "glu_bwd")
        _lib.check(L.ia_dwconv_glu_bwd(_lib.ptr(dz), _lib.ptr(c2), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(dc2b),
                                       _lib.stream_ptr()), "ia_dwconv_glu_bwd")
        _lib.check(L.ia_dwconv_glu_wgrad(_lib.ptr(c2), _lib.ptr(lens), _lib.ptr(dz), B, T, d, ksz, _lib.ptr(dwb), _lib.ptr(dbb),
                                         _lib.ptr(scr), _lib.stream_ptr()), "ia_dwconv_glu_wgrad")
        torch.cuda.synchronize()
        # (same arithmetic, but the compiler's fma contraction differs between the kernels: at most one bf16 ulp)
        assert torch.allclose(dc2a.float(), dc2b.float(), rtol=8e-3, atol=1e-6)
        # (the regenerated window element of the last tap is contracted differently by the compiler: 1 ulp in that tap's sums)
        assert torch.allclose(dwa, dwb, rtol=1e-6, atol=1e-5) and torch.equal(dba, dbb)
