"""Front-end ops: log-mel features, per-feature normalisation + length masking + SpecAugment fill."""
import torch


def log_mel(signal, window, fb, n_fft=512, hop=160, preemph=0.97, dither=0.0, seed=0, log_guard=2 ** -24):
    """[B,L] f32 audio -> [B,n_mels,Tm] f32 log-mel power (features.py:408-444)."""
    x = signal
    if dither > 0.0:
        g = torch.Generator(device=x.device)
        g.manual_seed(int(seed))
        x = x + dither * torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=g)
    x = torch.cat((x[:, :1], x[:, 1:] - preemph * x[:, :-1]), dim=1)
    with torch.autocast(device_type=x.device.type, enabled=False):
        spec = torch.stft(x.float(), n_fft=n_fft, hop_length=hop, win_length=window.numel(), center=True,
                          window=window.float(), return_complex=True)
        power = spec.real.square() + spec.imag.square()
        mel = torch.matmul(fb.float(), power)
        return torch.log(mel + log_guard)


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
