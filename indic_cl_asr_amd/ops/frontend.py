"""Front-end ops: log-mel features, per-feature normalisation + length masking + SpecAugment fill."""
import math
import os

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


_FB_CHUNK = 8          # bins per filterbank chunk (csrc/frontend_fft.hip FF_CH)
_FB_MAXCHUNKS = 128


def _fft_tables(window, fb, n_fft, device):
    """Tables of the one-pass front end (csrc/frontend_fft.hip): twiddles exp(-2 pi i j / n_fft) rounded from fp64, and the
    filterbank cut into chunks of 8 consecutive bins over each filter's non-zero span.  None when the filterbank does not fit
    128 chunks (then the GEMM front end runs)."""
    key = ("fft", fb.data_ptr(), fb._version, n_fft, str(device))
    hit = _BASIS.get(key)
    if hit is not None:
        return hit[0]
    nb = n_fft // 2 + 1
    f = fb.detach().float().cpu()
    nm = f.shape[0]
    starts, vals, filt = [], [], []
    for m in range(nm):
        nz = torch.nonzero(f[m, :nb]).flatten()
        first = len(starts)
        if nz.numel():
            lo, hi = int(nz[0]), int(nz[-1]) + 1
            for s in range(lo, hi, _FB_CHUNK):
                row = torch.zeros(_FB_CHUNK)
                e = min(s + _FB_CHUNK, hi)
                row[:e - s] = f[m, s:e]
                starts.append(s)
                vals.append(row)
        filt.append((first, len(starts) - first))
    n_chunks = len(starts)
    tables = None
    if n_chunks <= _FB_MAXCHUNKS and nm <= 128 and f.shape[1] >= nb:
        cs = torch.zeros(_FB_MAXCHUNKS, dtype=torch.int32)
        cv = torch.zeros(_FB_MAXCHUNKS, _FB_CHUNK)
        if n_chunks:
            cs[:n_chunks] = torch.tensor(starts, dtype=torch.int32)
            cv[:n_chunks] = torch.stack(vals)
        j = torch.arange(n_fft, dtype=torch.float64)
        ang = 2.0 * math.pi * j / n_fft
        tw = torch.stack([torch.cos(ang), -torch.sin(ang)], dim=1).float()
        tables = (tw.to(device).contiguous(), cs.to(device), cv.to(device).contiguous(),
                  torch.tensor(filt, dtype=torch.int32).to(device).contiguous(), n_chunks)
    _BASIS[key] = (tables, fb)     # fb kept alive: the key is its data pointer
    return tables


def log_mel(signal, window, fb, n_fft=512, hop=160, preemph=0.97, dither=0.0, seed=0, log_guard=2 ** -24):
    """[B,L] f32 audio -> [B,n_mels,Tm] f32 log-mel power (features.py:408-444) on the HIP front end.  n_fft = 512 (every
    recipe of the reference): csrc/frontend_fft.hip -- pre-emphasis pass, then ONE kernel from frames to log-mel (window,
    FFT, power, sparse mel projection, log; nothing in HBM in between).  Other sizes (or IA_FRONTEND=gemm):
    csrc/frontend.hip + csrc/gemm_f32.hip -- framing kernel, exact-fp32 MFMA DFT, power, exact-fp32 MFMA mel projection,
    log + transpose."""
    from .. import _lib
    L_ = _lib.lib()
    if not signal.is_cuda:
        raise RuntimeError("log_mel: device tensor required (no CPU path in the product)")
    x = signal.float().contiguous()
    B, L = x.shape
    win = window.numel()
    Tm = (L + (n_fft // 2) * 2 - n_fft) // hop + 1
    nm = fb.shape[0]
    if (os.environ.get("IA_FRONTEND", "fft") != "gemm" and L_.ia_feat_logmel_fft_supported(n_fft, win, nm, 0)):
        tables = _fft_tables(window, fb, n_fft, x.device)
        if tables is not None:
            tw, cs, cv, filt, n_chunks = tables
            y = torch.empty_like(x)
            _lib.check(L_.ia_feat_preemph(_lib.ptr(x), B, L, float(preemph), float(dither), int(seed) & 0xFFFFFFFF,
                                          _lib.ptr(y), _lib.stream_ptr()), "ia_feat_preemph")
            w = window.detach().to(device=x.device, dtype=torch.float32).contiguous()
            out = torch.empty(B, nm, Tm, dtype=torch.float32, device=x.device)
            _lib.check(L_.ia_feat_logmel_fft(_lib.ptr(y), B, L, Tm, _lib.ptr(w), win, n_fft, hop, _lib.ptr(tw), _lib.ptr(cs),
                                             _lib.ptr(cv), _lib.ptr(filt), nm, n_chunks, float(log_guard), _lib.ptr(out),
                                             _lib.stream_ptr()), "ia_feat_logmel_fft")
            return out
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
