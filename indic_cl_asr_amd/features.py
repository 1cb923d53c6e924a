"""Log-mel front end + SpecAugment for MI355X (reference surface: AudioToMelSpectrogramPreprocessor /
SpectrogramAugmentation, A/modules/audio_preprocessing.py:88-94,530-540; arithmetic of FilterbankFeatures
A/parts/preprocessing/features.py:400-471 and spec_aug_numba.py:26-95,250-305).

Device-resident throughout: frame counts are integer arithmetic on the host lengths the batch already carries,
normalisation has no per-utterance Python loop, SpecAugment spans are drawn on the device (torch.randint
semantics of the reference's CUDA branch) and applied in the same pass as the normalisation epilogue.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


def mel_filterbank_slaney(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=None):
    """Slaney-style mel triangles with area normalisation == librosa.filters.mel(norm='slaney', htk=False),
    which the reference stores as the `fb` buffer (features.py:327-333).  float64 construction, float32 result."""
    fmax = float(fmax or sr / 2)
    lin_step = 200.0 / 3.0           # Hz per mel below 1 kHz
    brk_hz, brk_mel = 1000.0, 1000.0 / lin_step
    log_step = math.log(6.4) / 27.0  # mel per natural-log octave fraction above 1 kHz

    def to_mel(hz):
        return hz / lin_step if hz < brk_hz else brk_mel + math.log(hz / brk_hz) / log_step

    def to_hz(mel):
        return lin_step * mel if mel < brk_mel else brk_hz * math.exp(log_step * (mel - brk_mel))

    lo, hi = to_mel(fmin), to_mel(fmax)
    edges = np.array([to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)], np.float64)
    bins = np.arange(n_fft // 2 + 1, dtype=np.float64) * (sr / n_fft)
    fb = np.zeros((n_mels, bins.size), np.float64)
    for m in range(n_mels):
        left, centre, right = edges[m], edges[m + 1], edges[m + 2]
        up = (bins - left) / (centre - left)
        down = (right - bins) / (right - centre)
        fb[m] = np.clip(np.minimum(up, down), 0.0, None) * (2.0 / (right - left))
    return fb.astype(np.float32)


def mel_frame_count(n_samples: int, n_fft=512, hop=160) -> int:
    """features.py:390-394 with center=True: floor((L + 2*(n_fft//2) - n_fft) / hop) + 1."""
    return (n_samples + (n_fft // 2) * 2 - n_fft) // hop + 1


class FilterbankFeaturizer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.win_length, self.hop_length, self.n_fft = cfg.n_window_size, cfg.n_window_stride, cfg.n_fft
        self.register_buffer("window", torch.hann_window(cfg.n_window_size, periodic=False))
        self.register_buffer("fb", torch.from_numpy(
            mel_filterbank_slaney(cfg.sample_rate, cfg.n_fft, cfg.feat_in)).unsqueeze(0))
        self.dither, self.preemph, self.pad_to = cfg.dither, cfg.preemph, cfg.pad_to
        self.log_zero_guard_value = 2 ** -24

    def get_seq_len(self, seq_len):
        return torch.div(seq_len + (self.n_fft // 2) * 2 - self.n_fft, self.hop_length, rounding_mode="floor").long() + 1


class AudioToMelSpectrogramPreprocessor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.featurizer = FilterbankFeaturizer(cfg)

    @torch.no_grad()
    def forward(self, input_signal, length, spec_aug=None, dither=True, seed=0, seq_len=None):
        """-> (processed_signal [B, feat_in, Tm] f32, processed_length [B] i64).
        `seq_len`: optional precomputed frame counts ([B] i64 device tensor, same rule, from host lengths).
        `spec_aug`: optional (freq_starts, freq_widths, time_starts, time_widths) i32 [B,M] device tensors applied
        in the normalisation epilogue."""
        f = self.featurizer
        use_dither = bool(dither and self.training and f.dither > 0)
        if seq_len is None:
            seq_len = f.get_seq_len(length)
        x = ops.log_mel(input_signal, f.window, f.fb[0], n_fft=f.n_fft, hop=f.hop_length, preemph=f.preemph,
                        dither=f.dither if use_dither else 0.0, seed=seed, log_guard=f.log_zero_guard_value)
        x = ops.normalize_mask(x, seq_len, spec_aug)
        if f.pad_to > 0 and x.size(-1) % f.pad_to != 0:
            x = torch.nn.functional.pad(x, (0, f.pad_to - x.size(-1) % f.pad_to))
        return x, seq_len


class SpectrogramAugmentation(nn.Module):
    """Draws the SpecAugment spans (spec_aug_numba.py:250-305 semantics); the fill itself runs inside
    ops.normalize_mask / ops.spec_augment_ so the feature tensor makes one HBM round trip."""

    def __init__(self, cfg):
        super().__init__()
        self.freq_masks, self.time_masks = cfg.freq_masks, cfg.time_masks
        self.freq_width, self.time_width = cfg.freq_width, cfg.time_width
        self.mask_value = 0.0

    @torch.no_grad()
    def draw(self, length, n_freq, generator=None):
        B, dev = length.shape[0], length.device
        if self.freq_masks > 0:
            fs = torch.randint(0, n_freq - self.freq_width + 1, (B, self.freq_masks), device=dev, generator=generator)
            fw = torch.randint(0, self.freq_width + 1, (B, self.freq_masks), device=dev, generator=generator)
        else:
            fs = fw = torch.zeros(B, 1, dtype=torch.int64, device=dev)
        if self.time_masks > 0:
            if isinstance(self.time_width, float):
                tw = (length * self.time_width).int().clamp(min=1)
            else:
                tw = torch.full((B,), int(self.time_width), dtype=torch.int32, device=dev)
            hi_start = (length - tw).clamp(min=1).unsqueeze(1).float()
            hi_len = (tw + 1).unsqueeze(1).float()
            # randint(0, hi) per row without a per-sample loop: floor(U[0,1) * hi)
            u1 = torch.rand(B, self.time_masks, device=dev, generator=generator)
            u2 = torch.rand(B, self.time_masks, device=dev, generator=generator)
            ts = torch.minimum((u1 * hi_start).floor(), hi_start - 1).long()
            tl = torch.minimum((u2 * hi_len).floor(), hi_len - 1).long()
        else:
            ts = tl = torch.zeros(B, 1, dtype=torch.int64, device=dev)
        return fs.int().contiguous(), fw.int().contiguous(), ts.int().contiguous(), tl.int().contiguous()

    @torch.no_grad()
    def draw_host(self, lengths, n_freq, device, generator=None):
        """Same spans as draw() drawn with a CPU generator from HOST lengths (training_step has them): one pinned buffer and one
        asynchronous H2D copy instead of ~30 tiny device kernels at the head of every step."""
        B = len(lengths)
        length = torch.as_tensor(lengths, dtype=torch.int64)
        if self.freq_masks > 0:
            fs = torch.randint(0, n_freq - self.freq_width + 1, (B, self.freq_masks), generator=generator)
            fw = torch.randint(0, self.freq_width + 1, (B, self.freq_masks), generator=generator)
        else:
            fs = fw = torch.zeros(B, 1, dtype=torch.int64)
        if self.time_masks > 0:
            if isinstance(self.time_width, float):
                tw = (length * self.time_width).int().clamp(min=1)
            else:
                tw = torch.full((B,), int(self.time_width), dtype=torch.int32)
            hi_start = (length - tw).clamp(min=1).unsqueeze(1).float()
            hi_len = (tw + 1).unsqueeze(1).float()
            u1 = torch.rand(B, self.time_masks, generator=generator)
            u2 = torch.rand(B, self.time_masks, generator=generator)
            ts = torch.minimum((u1 * hi_start).floor(), hi_start - 1).long()
            tl = torch.minimum((u2 * hi_len).floor(), hi_len - 1).long()
        else:
            ts = tl = torch.zeros(B, 1, dtype=torch.int64)
        parts = [fs.int().reshape(-1), fw.int().reshape(-1), ts.int().reshape(-1), tl.int().reshape(-1)]
        sizes = [p.numel() for p in parts]
        host = torch.empty(sum(sizes), dtype=torch.int32, pin_memory=torch.device(device).type == "cuda")
        torch.cat(parts, out=host)
        dev = host.to(device, non_blocking=True)
        outs, o = [], 0
        for p, n in zip((fs, fw, ts, tl), sizes):
            outs.append(dev[o:o + n].view(p.shape))
            o += n
        return tuple(outs)

    @torch.no_grad()
    def forward(self, input_spec, length, generator=None):
        spans = self.draw(length, input_spec.shape[1], generator)
        return ops.spec_augment_(input_spec.clone(), length, spans, self.mask_value)
