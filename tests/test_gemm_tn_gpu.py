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
