"""Front-end ops: log-mel features, per-feature normalisation + length masking + SpecAugment fill."""
import math

import torch


_BASIS = {}


def _dft_basis(window, n_fft, ldf, device):
    """[2*half, ldf] f32: Hann-windowed cos / sin rows of the centred n_fft-point real DFT restricted to the window's
    support (torch.stft pads the window to n_fft centred: sample n of the window sits at n + (n_fft - win)//2)."""
    key = (window.data_ptr(), window._version, n_fft, ldf, str(device))
    hit = _BASIS.get(key)
    if hit is not None:
        return hit
    win = window.numel()
    nb = n_fft // 2 + 1
    half = (nb + 15) // 16 * 16
    off = (n_fft - win) // 2
    n = torch.arange(win, dtype=torch.float64)
    k = torch.arange(nb, dtype=torch.float64).unsqueeze(1)
    ang = 2.0 * math.pi * k * (n + off) / n_fft
    w = window.detach().double().cpu()
    basis = torch.zeros(2 * half, ldf, dtype=torch.float64)
    basis[:nb, :win] = torch.cos(ang) * w
    basis[half:half + nb, :win] = torch.sin(ang) * w
    out = (basis.float().to(device).contiguous(), half, nb)
    _BASIS[key] = out
    return out


def log_mel(signal, window, fb, n_fft=512, hop=160, preemph=0.97, dither=0.0, seed=0, log_guard=2 ** -24):
    """[B,L] f32 audio -> [B,n_mels,Tm] f32 log-mel power (features.py:408-444) on the HIP front end
    (csrc/frontend.hip + csrc/gemm_f32.hip): framing kernel, exact-fp32 MFMA DFT, power, exact-fp32 MFMA mel
    projection, log + transpose."""
    from .. import _lib
    L_ = _lib.lib()
    if not signal.is_cuda:
        raise RuntimeError("log_mel: device tensor required (no CPU path in the product)")
    x = signal.float().contiguous()
    B, L = x.shape
    win = window.numel()
    Tm = (L + (n_fft // 2) * 2 - n_fft) // hop + 1
    ldf = (win + 15) // 16 * 16
    basis, half, nb = _dft_basis(window, n_fft, ldf, x.device)
    M = B * Tm
    dev = x.device
    frames = torch.empty(M, ldf, dtype=torch.float32, device=dev)
    _lib.check(L_.ia_feat_frames(_lib.ptr(x), B, L, Tm, win, hop, float(preemph), float(dither), int(seed) & 0xFFFFFFFF,
                                 _lib.ptr(frames), ldf, _lib.stream_ptr()), "ia_feat_frames")
    spec = torch.empty(M, 2 * half, dtype=torch.float32, device=dev)
    _lib.check(L_.ia_gemm_f32(_lib.ptr(frames), ldf, _lib.ptr(basis), ldf, M, 2 * half, ldf, _lib.ptr(spec), 2 * half,
                              _lib.stream_ptr()), "ia_gemm_f32")
    power = torch.empty(M, half, dtype=torch.float32, device=dev)
    _lib.check(L_.ia_feat_power(_lib.ptr(spec), M, 2 * half, half, nb, _lib.ptr(power), half, _lib.stream_ptr()),
               "ia_feat_power")
    nm = fb.shape[0]
    fbp = _BASIS.get(("fb", fb.data_ptr(), fb._version, half))
    if fbp is None:
        fbp = torch.zeros(nm, half, dtype=torch.float32, device=dev)
        fbp[:, :nb] = fb.float()
        _BASIS[("fb", fb.data_ptr(), fb._version, half)] = fbp
    mel = torch.empty(M, nm, dtype=torch.float32, device=dev)
    _lib.check(L_.ia_gemm_f32(_lib.ptr(power), half, _lib.ptr(fbp), half, M, nm, half, _lib.ptr(mel), nm,
                              _lib.stream_ptr()), "ia_gemm_f32")
    out = torch.empty(B, nm, Tm, dtype=torch.float32, device=dev)
    _lib.check(L_.ia_feat_logmel_t(_lib.ptr(mel), B, Tm, nm, nm, float(log_guard), _lib.ptr(out), _lib.stream_ptr()),
               "ia_feat_logmel_t")
    return out


def normalize_mask(x, seq_len, spec_aug=None, eps=1e-5, mask_value=0.0):
    """Per-utterance per-feature mean / unbiased std over the valid frames, +1e-5, zero beyond seq_len
    (features.py:59-76,458-462) with the SpecAugment fill in the same pass: one HIP launch (csrc/frontend.hip)."""
    from .. import _lib
    B, F, T = x.shape
    if not x.is_cuda or T > 4096:
        raise RuntimeError("normalize_mask: device tensor with Tm <= 4096 required")
    x = x.float().contiguous()
    y = torch.empty_like(x)
    if spec_aug is not None:
        fs, fw, ts, tw = (t.int().contiguous() for t in spec_aug)
        nf, nt = fs.shape[1], ts.shape[1]
    else:
        fs = fw = ts = tw = None
        nf = nt = 0
    st = _lib.lib().ia_feat_normalize(_lib.ptr(x), _lib.ptr(seq_len.long().contiguous()), B, F, T, float(eps), _lib.ptr(fs),
                                      _lib.ptr(fw), nf, _lib.ptr(ts), _lib.ptr(tw), nt, float(mask_value), _lib.ptr(y),
                                      _lib.stream_ptr())
    _lib.check(st, "ia_feat_normalize")
    return y


def spec_augment_(x, length, spans, mask_value=0.0):
    """In-place SpecAugment fill (spec_aug_numba.py:26-95): frequency spans over all frames, time spans only
    below length[b]."""
    fs, fw, ts, tw = spans
    B, F, T = x.shape
    f = torch.arange(F, device=x.device).view(1, 1, F)
    fmask = ((f >= fs.unsqueeze(-1)) & (f < (fs + fw).unsqueeze(-1))).any(1)  # [B,F]
    t = torch.arange(T, device=x.device).view(1, 1, T)
    tmask = ((t >= ts.unsqueeze(-1)) & (t < (ts + tw).unsqueeze(-1))).any(1)  # [B,T]
    tmask = tmask & (t.view(1, T) < length.view(B, 1))
    return x.masked_fill_(fmask.unsqueeze(2) | tmask.unsqueeze(1), mask_value)
