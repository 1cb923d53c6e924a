"""csrc/gemm_bf16.hip ConvSubsampling path (direct 1->C conv, implicit-GEMM 3x3/s2 conv on MFMA, permuted Linear) against the
module's fp32 ATen forward (A/parts/submodules/subsampling.py:217-253,385-437), for channel counts with and without
64-alignment (Conformer-small has 144 channels: k-tiles of the implicit GEMM straddle taps)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,d,B,Tm", [(256, 256, 3, 301), (144, 144, 2, 200), (64, 64, 2, 77), (176, 176, 1, 50)])
def test_conv_subsampling_matches_the_module(C, d, B, Tm):
    from indic_cl_asr_amd.encoder import ConvSubsampling
    from indic_cl_asr_amd.ops import fast
    torch.manual_seed(C + Tm)
    feat_in = 80
    m = ConvSubsampling(feat_in, d, C).cuda()
    assert fast.subsample_supported(C, d, feat_in)
    x = torch.randn(B, feat_in, Tm, device="cuda")
    lens = torch.full((B,), Tm, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        ref, _ = m(x.transpose(1, 2), lens)                                   # [B, T2, d] fp32
        out = fast.conv_subsampling(x, m.conv[0], m.conv[2], m.out).view(ref.shape)
    err = (out - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), (err, ref.abs().max().item())   # bf16 operands through two convolutions + Linear
