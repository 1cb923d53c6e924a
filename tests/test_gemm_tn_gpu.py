"""Weight-gradient GEMM (csrc/gemm_tn.hip, transposing LDS reads + split-K partial tiles) through the C ABI against an
fp32 ATen product of the same bf16 operands; bias gradient as the column sums of dY."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,n,k,pad", [(12032, 256, 1024, 0), (12032, 1024, 256, 0), (1000, 768, 256, 0), (77, 72, 40, 8),
                                       (64, 128, 128, 0), (4097, 264, 136, 16), (333, 8, 8, 0)])
def test_gemm_tn_matches_fp32_product(M, n, k, pad):
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + n + k)
    dYs = (torch.randn(M, n + pad, generator=g) * 0.5).bfloat16().cuda()
    Xs = (torch.randn(M, k + pad, generator=g) * 0.5).bfloat16().cuda()
    dY, X = dYs[:, :n], Xs[:, :k]                       # row strides larger than the logical widths when pad > 0
    dW = torch.full((n, k), float("nan"), device="cuda")
    db = torch.full((n,), float("nan"), device="cuda")
    scr = torch.empty(max(1, L.ia_gemm_tn_scratch_elems(M, n, k)), dtype=torch.float32, device="cuda")
    _lib.check(L.ia_gemm_tn_bf16(_lib.ptr(dYs), dYs.stride(0), _lib.ptr(Xs), Xs.stride(0), M, n, k, _lib.ptr(dW), _lib.ptr(db),
                                 _lib.ptr(scr), _lib.stream_ptr()), "ia_gemm_tn_bf16")
    ref = dY.float().t() @ X.float()
    refb = dY.float().sum(0)
    tol = 2e-3 * ref.abs().max().item() + 1e-4          # fp32 accumulation in a different order
    assert (dW - ref).abs().max().item() <= tol, ((dW - ref).abs().max().item(), ref.abs().max().item())
    assert (db - refb).abs().max().item() <= 2e-3 * refb.abs().max().item() + 1e-3
    # without the bias output
    dW2 = torch.empty(n, k, device="cuda")
    _lib.check(L.ia_gemm_tn_bf16(_lib.ptr(dYs), dYs.stride(0), _lib.ptr(Xs), Xs.stride(0), M, n, k, _lib.ptr(dW2), None,
                                 _lib.ptr(scr), _lib.stream_ptr()), "ia_gemm_tn_bf16")
    assert torch.equal(dW2, dW)                          # deterministic (no atomics)


def test_grouped_weight_gradients_equal_the_single_launches():
    """The grouped form (one GEMM launch + one finishing launch for a block's projections) against one launch per problem:
    same kernel body, different split counts -> equal up to the order of the f32 split sums."""
    from indic_cl_asr_amd.ops import fast
    g = torch.Generator().manual_seed(3)
    shapes = [(3008, 256, 1024), (3008, 1024, 256), (3008, 256, 256), (3008, 512, 256), (751, 256, 256), (3008, 768, 256)]
    pairs = [((torch.randn(M, N, generator=g) * 0.3).bfloat16().cuda(), (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda())
             for M, N, K in shapes]
    outs = fast.gemm_tn_grouped(pairs)
    for (dy, x), (dW, db) in zip(pairs, outs):
        rW, rb = fast.gemm_tn(dy, x)
        assert (dW - rW).abs().max().item() <= 2e-5 * rW.abs().max().item() + 1e-6
        assert (db - rb).abs().max().item() <= 2e-5 * rb.abs().max().item() + 1e-6
        ref = dy.double().t() @ x.double()
        assert (dW.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
