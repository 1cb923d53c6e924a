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


@pytest.mark.parametrize("tag", ["a", "b"])
def test_conv_subsampling_hip_matches_reference_file_outputs(tag):
    """The HIP path against what the REFERENCE's own subsampling.py returned on seeded inputs (tests/golden/
    subsampling_cases.npz, made by tests/golden/make_golden.py: the file loaded unmodified behind empty package stubs)."""
    import os

    import numpy as np
    from conftest import GOLDEN
    from indic_cl_asr_amd.encoder import ConvSubsampling, subsampled_length
    from indic_cl_asr_amd.ops import fast
    Z = np.load(os.path.join(GOLDEN, "subsampling_cases.npz"))
    pre = f"sub/{tag}/param/"
    sd = {k[len(pre):]: torch.tensor(Z[k]) for k in Z.files if k.startswith(pre)}
    C, d = sd["conv.0.weight"].shape[0], sd["out.weight"].shape[0]
    x, want = torch.tensor(Z[f"sub/{tag}/x"]), torch.tensor(Z[f"sub/{tag}/y"])
    m = ConvSubsampling(x.shape[-1], d, C)
    m.load_state_dict(sd)
    m = m.cuda()
    assert fast.subsample_supported(C, d, x.shape[-1])
    with torch.no_grad():
        out = fast.conv_subsampling(x.transpose(1, 2).contiguous().cuda(), m.conv[0], m.conv[2], m.out).view(want.shape)
    err = (out.cpu() - want).abs().max().item()
    assert err <= 2e-2 * want.abs().max().item(), (err, want.abs().max().item())   # bf16 operands (see the test above)
    lens = torch.tensor(Z[f"sub/{tag}/lens"])
    assert torch.equal(subsampled_length(lens), torch.tensor(Z[f"sub/{tag}/ylen"]).long())


def test_k_pipelined_implicit_gemm_convolution_is_bit_identical_to_the_register_staged_one():
    """ia_subsample_conv2 on the LDS-DMA kernel (C % 64 == 0: padding taps read a page of zeros through an address select)
    against the register-staged implicit GEMM: same products in the same order."""
    import os
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    B, T1, F1, C, N = 3, 151, 40, 256, 256
    g = torch.Generator(device="cuda").manual_seed(4)
    x = (torch.randn(B, T1, F1, C, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, 9 * C, device="cuda", generator=g) * 0.02).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    outs = []
    try:
        for mode in ("0", "1"):
            os.environ["IA_CONV_DMA"] = mode
            o = torch.empty(B * T2 * F2, N, dtype=torch.bfloat16, device="cuda")
            _lib.check(L.ia_subsample_conv2(_lib.ptr(x), B, T1, F1, C, _lib.ptr(w), _lib.ptr(b), N, _lib.ptr(o), _lib.stream_ptr()), "conv2")
            outs.append(o)
    finally:
        os.environ.pop("IA_CONV_DMA", None)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    # and against ATen on the same bf16 operands
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(N, 3, 3, C).permute(0, 3, 1, 2), b, stride=2, padding=1)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(B * T2 * F2, N)
    assert (outs[1].float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
