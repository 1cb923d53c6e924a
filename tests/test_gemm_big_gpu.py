"""csrc/gemm_big.hip (256 x 256 tiles, 32x32x16 MFMAs, operands global -> LDS by DMA) against the 128-row projection GEMM
it replaces for very large problems: same arguments, same epilogue; the two kernels add the k products in the same order
inside a 64-deep stage, so the results agree to fp32 rounding of the accumulated sums."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(777, 256, 128), (2048 + 13, 512, 320), (256, 768, 1024)])
def test_large_tile_gemm_matches_the_128_row_kernel(M, N, K):
    from indic_cl_asr_amd.ops import fast
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    outs = []
    try:
        for mode in ("0", "1"):
            os.environ["IA_GEMM_BIG"] = mode
            of, oh = fast.gemm(a, w, bias, act=1, dropout_p=0.1, seed=3, alpha=0.5, residual=res, out_f32=torch.empty_like(res))
            outs.append((of, oh))
    finally:
        os.environ.pop("IA_GEMM_BIG", None)
    torch.cuda.synchronize()
    ref = (a.double() @ w.double().t() + bias.double())
    ref = ref * torch.sigmoid(ref)
    scale = ref.abs().max().item()
    keep = outs[0][0] != res                                        # the dropped elements equal the residual in both
    assert torch.equal(keep, outs[1][0] != res)
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2e-6 * scale
    got = (outs[1][0].double() - res.double()) / (0.5 / 0.8984375)     # undo alpha and the keep scale 256 / (256 - 26)
    assert ((got - ref).abs() * keep).max().item() <= 4e-3 * scale       # bf16 operands, fp32 accumulation


@pytest.mark.parametrize("M,K,p", [(12032, 256, 0.1), (333, 256, 0.0), (1000, 512, 0.25)])
def test_projection_with_layernorm_epilogue_matches_the_two_launches(M, K, p):
    """ia_gemm_bf16_ln (64 x 256 tiles, LayerNorm of the finished rows in the epilogue) against ia_gemm_bf16 + ia_layernorm:
    the residual update is the same arithmetic in the same order (bit-identical), the LayerNorm output agrees to bf16 rounding
    of values whose row statistics were summed in another order."""
    from indic_cl_asr_amd import _lib
    from indic_cl_asr_amd.ops import fast
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(M)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(256, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(256, device="cuda", generator=g)
    x0 = torch.randn(M, 256, device="cuda", generator=g)
    ln = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
    xa = x0.clone()
    fast.gemm(a, w, bias, dropout_p=p, seed=9, alpha=1.0, residual=xa, out_f32=xa, want_bf16=False)
    ya = fast.layernorm(xa, ln.weight, ln.bias, ln.eps)
    xb = x0.clone()
    yb = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda")
    _lib.check(L.ia_gemm_bf16_ln(_lib.ptr(a), K, _lib.ptr(w), K, M, 256, K, _lib.ptr(bias), p, 9, 1.0, _lib.ptr(xb), 256, _lib.ptr(xb), 256,
                                 _lib.ptr(ln.weight), _lib.ptr(ln.bias), ln.eps, _lib.ptr(yb), 256, _lib.stream_ptr()), "ia_gemm_bf16_ln")
    torch.cuda.synchronize()
    assert torch.equal(xa, xb)
    assert (ya.float() - yb.float()).abs().max().item() <= 2.0 ** -7 * ya.float().abs().max().item()
    assert (ya != yb).float().mean().item() < 0.02


@pytest.mark.parametrize("M,N,K", [(12032, 256, 1024), (777, 264, 512), (3000, 768, 768), (130, 128, 1536)])
def test_k_pipelined_projection_kernel_is_bit_identical_to_the_register_staged_one(M, N, K):
    """gemm_bf16_nt_dma_kernel (operands by LDS-DMA into a three-stage ring, chosen for K >= 512 with 64-row tiles) against
    gemm_bf16_nt_kernel<64, 128>: same tile shape, same MFMA order, same epilogue -- identical bits, ragged M / N included."""
    from indic_cl_asr_amd.ops import fast
    g = torch.Generator(device="cuda").manual_seed(K + M)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    outs = []
    try:
        for mode in ("0", "1"):
            os.environ["IA_GEMM_DMA"] = mode
            os.environ["IA_GEMM_BM"] = "64"
            of, oh = fast.gemm(a, w, bias, act=1, dropout_p=0.1, seed=3, alpha=0.5, residual=res, out_f32=torch.empty_like(res))
            outs.append((of, oh))
    finally:
        os.environ.pop("IA_GEMM_DMA", None)
        os.environ.pop("IA_GEMM_BM", None)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = a.double() @ w.double().t() + bias.double()
    ref = ref * torch.sigmoid(ref)
    keep = outs[1][0] != res
    got = (outs[1][0].double() - res.double()) / (0.5 / 0.8984375)
    assert ((got - ref).abs() * keep).max().item() <= 4e-3 * ref.abs().max().item()
