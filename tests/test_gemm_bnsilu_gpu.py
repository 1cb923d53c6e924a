"""BatchNorm + SiLU folded into the pointwise-convolution GEMM (csrc/gemm_bnsilu.hip) against the two launches it replaces
(ia_bn_silu then ia_gemm_bf16): same rounding points, same k order, same dropout mask -> identical bits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(M, d, N, training, p, seed=11):
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + d + N)
    dev = "cuda"
    z = (torch.randn(M, d, generator=g) * 1.5 + 0.3).to(dev)
    W = (torch.randn(N, d, generator=g) * 0.1).bfloat16().to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dev)
    beta = (0.1 * torch.randn(d, generator=g)).to(dev)
    x0 = torch.randn(M, N, generator=g).to(dev)
    s1, s2 = z.sum(0).contiguous(), (z * z).sum(0).contiguous()

    def stats():
        return (torch.full((d,), 0.05, device=dev), torch.full((d,), 0.9, device=dev), torch.zeros(1, dtype=torch.long, device=dev))

    # reference: two launches
    rm_a, rv_a, nbt_a = stats()
    c3 = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    _lib.check(L.ia_bn_silu(_lib.ptr(z), M, d, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(rm_a), _lib.ptr(rv_a),
                            _lib.ptr(nbt_a), 0.1, 1e-5, int(training), _lib.ptr(c3), _lib.stream_ptr()), "ia_bn_silu")
    xa = x0.clone()
    _lib.check(L.ia_gemm_bf16(_lib.ptr(c3), d, _lib.ptr(W), d, M, N, d, _lib.ptr(bias), 0, float(p), seed, 1.0, _lib.ptr(xa), N, _lib.ptr(xa), N,
                              None, 0, _lib.stream_ptr()), "ia_gemm_bf16")
    # fused
    rm_b, rv_b, nbt_b = stats()
    xb = x0.clone()
    yb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    _lib.check(L.ia_gemm_bnsilu_bf16(_lib.ptr(z), d, M, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(rm_b),
                                     _lib.ptr(rv_b), _lib.ptr(nbt_b), 0.1, 1e-5, int(training), _lib.ptr(W), d, M, N, d, _lib.ptr(bias),
                                     float(p), seed, 1.0, _lib.ptr(xb), N, _lib.ptr(xb), N, _lib.ptr(yb), N, None, _lib.stream_ptr()),
               "ia_gemm_bnsilu_bf16")
    # ... and the variant that keeps SiLU(BN(z)) for a backward
    rm_c, rv_c, nbt_c = stats()
    xc = x0.clone()
    c3k = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    _lib.check(L.ia_gemm_bnsilu_bf16_keep(_lib.ptr(z), d, M, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(rm_c),
                                          _lib.ptr(rv_c), _lib.ptr(nbt_c), 0.1, 1e-5, int(training), _lib.ptr(W), d, M, N, d,
                                          _lib.ptr(bias), float(p), seed, 1.0, _lib.ptr(xc), N, _lib.ptr(xc), N, None, 0, None,
                                          _lib.ptr(c3k), d, _lib.stream_ptr()), "ia_gemm_bnsilu_bf16_keep")
    torch.cuda.synchronize()
    assert torch.equal(xa, xb)
    assert torch.equal(xa, xc) and torch.equal(c3, c3k)
    assert torch.equal(yb, xb.bfloat16())
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b) and torch.equal(nbt_a, nbt_b)
    # and against plain torch arithmetic (bf16 operands, fp32 accumulation)
    mean = s1 / M if training else torch.full((d,), 0.05, device=dev)
    var = (s2 / M - mean * mean).clamp_min(0) if training else torch.full((d,), 0.9, device=dev)
    y = (z - mean) * torch.rsqrt(var + 1e-5) * gamma + beta
    a = torch.nn.functional.silu(y).bfloat16().float()
    ref = a @ W.float().t() + bias
    if p == 0.0:
        assert torch.allclose(xb - x0, ref, rtol=2e-2, atol=2e-2)
    else:
        kept = (xb - x0) != 0
        assert 0.8 < kept.float().mean().item() < 0.97          # p = 0.1
        assert torch.allclose((xb - x0)[kept], (ref / (1 - round(p * 256) / 256))[kept], rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("M,d,N", [(12032, 256, 256), (1000, 144, 144), (70, 512, 512), (333, 256, 640)])
@pytest.mark.parametrize("training", [True, False])
def test_bnsilu_gemm_matches_two_launches(M, d, N, training):
    _run(M, d, N, training, 0.0)
    _run(M, d, N, training, 0.1)


def test_glu_epilogue_of_the_regrouped_pointwise_conv1():
    """ia_gemm_bf16_ex act 4 on fast.glu_regrouped weights == GLU(x @ W^T + b) of the original layout."""
    from indic_cl_asr_amd import _lib
    from indic_cl_asr_amd.ops import fast
    L = _lib.lib()
    for M, d in ((1000, 256), (130, 512), (64, 64)):
        g = torch.Generator().manual_seed(M + d)
        x = torch.randn(M, d, generator=g).bfloat16().cuda()
        w = torch.nn.Parameter((torch.randn(2 * d, d, 1, generator=g) * 0.1).cuda())
        b = torch.nn.Parameter((torch.randn(2 * d, generator=g) * 0.1).cuda())
        wg, bg = fast.glu_regrouped(w, b)
        out = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
        _lib.check(L.ia_gemm_bf16_ex(_lib.ptr(x), d, _lib.ptr(wg), d, M, 2 * d, d, _lib.ptr(bg), 4, 0.0, 0, 1.0, None, 0, None, 0,
                                     _lib.ptr(out), d, None, 0, None, 0, _lib.stream_ptr()), "ia_gemm_bf16_ex")
        ref = torch.nn.functional.glu(x.float() @ w.detach().reshape(2 * d, d).bfloat16().float().t() + b.detach(), dim=1)
        assert torch.allclose(out.float(), ref, rtol=1e-2, atol=1e-2)
    assert fast.glu_regrouped(torch.nn.Parameter(torch.zeros(288, 144, 1)), torch.nn.Parameter(torch.zeros(288))) is None


def test_depthwise_conv_on_gated_rows_matches_the_glu_variant():
    """ia_dwconv_gated_fixed(GLU(c2) rounded to bf16) against ia_glu_dwconv_fixed(c2): same masking, conv and sums up to the
    extra bf16 rounding of the gated value."""
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    B, T, d, ksz = 3, 150, 128, 31
    g = torch.Generator().manual_seed(5)
    c2 = torch.randn(B * T, 2 * d, generator=g).bfloat16().cuda()
    lens = torch.tensor([150, 97, 20], dtype=torch.long).cuda()
    w = (torch.randn(d, ksz, generator=g) * 0.2).cuda()
    bias = (torch.randn(d, generator=g) * 0.1).cuda()
    za, zb = torch.empty(B * T, d, device="cuda"), torch.empty(B * T, d, device="cuda")
    acc_a = torch.zeros(8, 2 * d, dtype=torch.int64, device="cuda")
    acc_b = torch.zeros(8, 2 * d, dtype=torch.int64, device="cuda")
    _lib.check(L.ia_glu_dwconv_fixed(_lib.ptr(c2), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(za), _lib.ptr(acc_a),
                                     _lib.stream_ptr()), "ia_glu_dwconv_fixed")
    gated = torch.nn.functional.glu(c2.float(), dim=1).bfloat16().contiguous()
    _lib.check(L.ia_dwconv_gated_fixed(_lib.ptr(gated), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(zb),
                                       _lib.ptr(acc_b), _lib.stream_ptr()), "ia_dwconv_gated_fixed")
    torch.cuda.synchronize()
    assert torch.allclose(za, zb, rtol=2e-2, atol=2e-2)
    sa, sb = acc_a.sum(0).double() / 2 ** 24, acc_b.sum(0).double() / 2 ** 24
    assert torch.allclose(sa, sb, rtol=2e-2, atol=0.5)
