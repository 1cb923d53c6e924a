"""One-pass log-mel front end (csrc/frontend_fft.hip: pre-emphasis, then frames -> window -> 512-point FFT -> power -> sparse
mel -> log in one kernel) against an fp64 restatement of FilterbankFeatures.forward (A/parts/preprocessing/features.py:408-444)
and against the GEMM front end it replaces."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_logmel(x, window, fb, n_fft=512, hop=160, preemph=0.97, guard=2 ** -24):
    x = x.double()
    y = torch.cat([x[:, :1], x[:, 1:] - preemph * x[:, :-1]], dim=1)
    spec = torch.stft(y, n_fft=n_fft, hop_length=hop, win_length=window.numel(), center=True, window=window.double(),
                      return_complex=True, pad_mode="reflect")
    power = spec.real ** 2 + spec.imag ** 2                     # [B, 257, Tm]
    mel = torch.matmul(fb.double(), power)
    return torch.log(mel + guard)


def _setup():
    from indic_cl_asr_amd.features import mel_filterbank_slaney
    fb = torch.as_tensor(mel_filterbank_slaney()).float()
    window = torch.hann_window(400, periodic=False)
    return fb, window


@pytest.mark.parametrize("B,L", [(3, 24000), (2, 4001), (1, 330), (4, 16000 * 5 + 77)])
def test_logmel_fft_matches_fp64(B, L):
    from indic_cl_asr_amd.ops import frontend
    fb, window = _setup()
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, L, generator=g) * 0.1
    x[0, : L // 3] *= 1e-3                                      # a quiet stretch: small powers next to large ones
    ref = _ref_logmel(x, window, fb)
    got = frontend.log_mel(x.cuda(), window.cuda(), fb.cuda(), dither=0.0).cpu().double()
    assert got.shape == ref.shape
    # fp32 transform of a unit-scale frame: absolute spectrum error ~1e-6 of the frame's largest bin -> compare the mel
    # energies relative to each frame's largest one
    mel_ref, mel_got = ref.exp(), got.exp()
    scale = mel_ref.amax(dim=1, keepdim=True)
    assert ((mel_got - mel_ref).abs() / scale).max().item() < 2e-5
    assert (got - ref).abs().max().item() < 2e-3


def test_logmel_fft_matches_gemm_front_end_with_dither():
    from indic_cl_asr_amd.ops import frontend
    fb, window = _setup()
    x = (torch.randn(2, 12345, generator=torch.Generator().manual_seed(1)) * 0.05).cuda()
    a = frontend.log_mel(x, window.cuda(), fb.cuda(), dither=1e-5, seed=77)
    os.environ["IA_FRONTEND"] = "gemm"
    try:
        b = frontend.log_mel(x, window.cuda(), fb.cuda(), dither=1e-5, seed=77)
    finally:
        del os.environ["IA_FRONTEND"]
    assert a.shape == b.shape
    assert (a - b).abs().max().item() < 2e-3                    # same counter-based noise in both paths
    c = frontend.log_mel(x, window.cuda(), fb.cuda(), dither=1e-5, seed=78)
    assert (a - c).abs().max().item() > 0.0                     # another seed, another noise


def test_logmel_fft_dense_filterbank_falls_back():
    """A filterbank without the triangles' sparsity does not fit the chunk table: the GEMM front end answers."""
    from indic_cl_asr_amd.ops import frontend
    _, window = _setup()
    fb = torch.rand(80, 257, generator=torch.Generator().manual_seed(2)) + 0.1
    x = torch.randn(1, 3000, generator=torch.Generator().manual_seed(3)) * 0.1
    ref = _ref_logmel(x, window, fb)
    got = frontend.log_mel(x.cuda(), window.cuda(), fb.cuda()).cpu().double()
    assert (got - ref).abs().max().item() < 2e-3
