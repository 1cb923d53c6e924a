"""CPU fp32 restatement of the reference's Conformer hybrid RNNT-CTC training step and CL arithmetic.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this file.  It follows the reference's text line by line (citations per function;
A/ = NeMo/nemo/collections/asr/, C/ = NeMo/nemo/collections/common/, R/ = reference repo root) and keeps the
reference's parameter names so a state_dict moves between this oracle, the product model and (if ever
supplied) a real IndicConformer checkpoint.

Pinning: the transducer loss is pinned to the reference's known answers through
oracle/rnnt_ref.c; rel-pos attention, the depthwise conv and the prediction LSTM are pinned against the
reference files that load standalone (tests/golden/module_cases.npz, tests/test_oracle_step.py).  The mel
filterbank constants restate librosa.filters.mel(norm='slaney') (features.py:327-333); librosa is absent from
the container, so that single boundary is "parity unpinned" (closed-form properties are tested instead).
Everything else composes this container's torch CPU ops exactly as the reference's source does.
"""
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import rnnt_oracle

LANGS22 = ['as', 'bn', 'brx', 'doi', 'gu', 'hi', 'kn', 'kok', 'ks', 'mai', 'ml', 'mni', 'mr', 'ne', 'or', 'pa', 'sa',
           'sat', 'sd', 'ta', 'te', 'ur']


# ------------------------------------------------------------------------------------------------ features
def slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=None):
    """librosa.filters.mel(htk=False, norm='slaney') restated (call site A/parts/preprocessing/features.py:327-333)."""
    fmax = fmax or sr / 2.0
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0

    def hz2mel(f):
        f = np.asarray(f, np.float64)
        m = f / f_sp
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, m)

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        f = f_sp * m
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


class FilterbankFeatures(nn.Module):
    """A/parts/preprocessing/features.py:229-471 with the recipe constants (yaml :63-74)."""

    def __init__(self, sample_rate=16000, n_window_size=400, n_window_stride=160, n_fft=512, nfilt=80, preemph=0.97,
                 dither=1e-5, pad_to=0):
        super().__init__()
        self.win_length, self.hop_length, self.n_fft = n_window_size, n_window_stride, n_fft
        self.preemph, self.dither, self.pad_to, self.nfilt = preemph, dither, pad_to, nfilt
        self.register_buffer("window", torch.hann_window(n_window_size, periodic=False))  # :306
        self.register_buffer("fb", torch.tensor(slaney_mel_filterbank(sample_rate, n_fft, nfilt)).unsqueeze(0))
        self.log_zero_guard_value = 2 ** -24

    def get_seq_len(self, seq_len):  # :390-394
        pad_amount = self.n_fft // 2 * 2
        return (torch.floor_divide((seq_len + pad_amount - self.n_fft), self.hop_length) + 1).to(dtype=torch.long)

    @torch.no_grad()
    def forward(self, x, seq_len, dither_noise=None):
        seq_len = self.get_seq_len(seq_len)
        if dither_noise is not None:  # :410-411 (x += dither * randn_like(x)); noise passed in for determinism
            x = x + self.dither * dither_noise
        x = torch.cat((x[:, 0].unsqueeze(1), x[:, 1:] - self.preemph * x[:, :-1]), dim=1)  # :414
        x = torch.stft(x, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length, center=True,
                       window=self.window, return_complex=True)  # :308-316, :418
        x = torch.view_as_real(x)
        x = torch.sqrt(x.pow(2).sum(-1))  # :424
        x = x.pow(2.0)  # :433
        x = torch.matmul(self.fb.to(x.dtype), x)  # :440
        x = torch.log(x + self.log_zero_guard_value)  # :444
        # normalize_batch 'per_feature' :59-76
        CONSTANT = 1e-5
        x_mean = torch.zeros((seq_len.shape[0], x.shape[1]), dtype=x.dtype)
        x_std = torch.zeros((seq_len.shape[0], x.shape[1]), dtype=x.dtype)
        for i in range(x.shape[0]):
            x_mean[i, :] = x[i, :, : seq_len[i]].mean(dim=1)
            x_std[i, :] = x[i, :, : seq_len[i]].std(dim=1)
        x_std += CONSTANT
        x = (x - x_mean.unsqueeze(2)) / x_std.unsqueeze(2)
        max_len = x.size(-1)  # :458-462
        mask = torch.arange(max_len).repeat(x.size(0), 1) >= seq_len.unsqueeze(1)
        x = x.masked_fill(mask.unsqueeze(1), 0.0)
        if self.pad_to > 0 and x.size(-1) % self.pad_to != 0:
            x = F.pad(x, (0, self.pad_to - x.size(-1) % self.pad_to), value=0.0)
        return x, seq_len


def spec_augment_apply(x, freq_spans, time_spans, mask_value=0.0):
    """Mask fill of A/parts/submodules/spectr_augment.py:83-113 / spec_aug_numba.py:26-95 given explicit spans.
    freq_spans[b] / time_spans[b]: lists of (start, width)."""
    x = x.clone()
    for b in range(x.shape[0]):
        for (s, w) in freq_spans[b]:
            x[b, s:s + w, :] = mask_value
        for (s, w) in time_spans[b]:
            x[b, :, s:s + w] = mask_value
    return x


# ------------------------------------------------------------------------------------------------ encoder
def calc_length(lengths, all_paddings=2, kernel_size=3, stride=2, repeat_num=2):
    """A/parts/submodules/subsampling.py:566-576 (float32 arithmetic, floor)."""
    add_pad = float(all_paddings - kernel_size)
    for _ in range(repeat_num):
        lengths = torch.div(lengths.to(dtype=torch.float) + add_pad, stride) + 1.0
        lengths = torch.floor(lengths)
    return lengths.to(dtype=torch.int)


class ConvSubsampling(nn.Module):
    """'striding' x4: A/parts/submodules/subsampling.py:217-253,369,385-437."""

    def __init__(self, feat_in, feat_out, conv_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, conv_channels, 3, 2, 1), nn.ReLU(True),
                                  nn.Conv2d(conv_channels, conv_channels, 3, 2, 1), nn.ReLU(True))
        out_len = int(calc_length(torch.tensor(feat_in, dtype=torch.float)))
        self.out = nn.Linear(conv_channels * out_len, feat_out)

    def forward(self, x, lengths):
        lengths = calc_length(lengths)
        x = self.conv(x.unsqueeze(1))
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).reshape(b, t, -1))
        return x, lengths


class RelPositionalEncoding(nn.Module):
    """A/parts/submodules/multi_head_attention.py:872-979 (xscale, sin/cos table, centre slice); dropout omitted
    (p given to F.dropout by the caller when training with dropout on)."""

    def __init__(self, d_model, max_len=5000, xscale=None):
        super().__init__()
        self.d_model, self.xscale = d_model, xscale
        positions = torch.arange(max_len - 1, -max_len, -1, dtype=torch.float32).unsqueeze(1)  # :944
        pe = torch.zeros(positions.size(0), d_model)
        div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(positions * div_term)
        pe[:, 1::2] = torch.cos(positions * div_term)
        self.register_buffer("pe", pe.unsqueeze(0), persistent=False)

    def forward(self, x):
        if self.xscale:
            x = x * self.xscale
        input_len = x.size(1)
        center_pos = self.pe.size(1) // 2 + 1
        return x, self.pe[:, center_pos - input_len: center_pos + input_len - 1]


class RelPositionMultiHeadAttention(nn.Module):
    """A/parts/submodules/multi_head_attention.py:52-250 (forward_qkv, rel_shift pad/view, -10000 fill,
    post-softmax zeroing)."""

    def __init__(self, n_head, n_feat):
        super().__init__()
        self.d_k, self.h = n_feat // n_head, n_head
        self.s_d_k = math.sqrt(self.d_k)
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(self.h, self.d_k))  # :178-179
        self.pos_bias_v = nn.Parameter(torch.zeros(self.h, self.d_k))

    @staticmethod
    def rel_shift(x):  # :184-195
        b, h, qlen, pos_len = x.size()
        x = F.pad(x, pad=(1, 0))
        x = x.view(b, h, -1, qlen)
        return x[:, :, 1:].view(b, h, qlen, pos_len)

    def forward(self, x, mask, pos_emb):
        B = x.size(0)
        q = self.linear_q(x).view(B, -1, self.h, self.d_k)
        k = self.linear_k(x).view(B, -1, self.h, self.d_k).transpose(1, 2)
        v = self.linear_v(x).view(B, -1, self.h, self.d_k).transpose(1, 2)
        p = self.linear_pos(pos_emb).view(pos_emb.size(0), -1, self.h, self.d_k).transpose(1, 2)
        q_u = (q + self.pos_bias_u).transpose(1, 2)
        q_v = (q + self.pos_bias_v).transpose(1, 2)
        ac = torch.matmul(q_u, k.transpose(-2, -1))
        bd = self.rel_shift(torch.matmul(q_v, p.transpose(-2, -1)))[:, :, :, : ac.size(-1)]
        scores = (ac + bd) / self.s_d_k
        m = mask.unsqueeze(1)
        scores = scores.masked_fill(m, -10000.0)
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)  # :108-111
        o = torch.matmul(attn, v).transpose(1, 2).reshape(B, -1, self.h * self.d_k)
        return self.linear_out(o)


class ConformerFeedForward(nn.Module):  # A/parts/submodules/conformer_modules.py:385-404
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)

    def forward(self, x):
        return self.linear2(F.silu(self.linear1(x)))


class _DW(nn.Conv1d):
    """CausalConv1D as configured by the Conformer conv module (symmetric pad 15/15): causal_convs.py:72-150."""

    def __init__(self, ch, k):
        super().__init__(ch, ch, k, stride=1, padding=0, groups=ch, bias=True)
        self._pad = (k - 1) // 2

    def forward(self, x):
        return super().forward(F.pad(x, (self._pad, self._pad)))


class ConformerConvolution(nn.Module):  # conformer_modules.py:280-370
    def __init__(self, d_model, kernel_size):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, 1)
        self.depthwise_conv = _DW(d_model, kernel_size)
        self.batch_norm = nn.BatchNorm1d(d_model)
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, 1)

    def forward(self, x, pad_mask):
        x = x.transpose(1, 2)
        x = F.glu(self.pointwise_conv1(x), dim=1)
        x = x.float().masked_fill(pad_mask.unsqueeze(1), 0.0)  # :351
        x = self.depthwise_conv(x)
        x = self.batch_norm(x)
        x = F.silu(x)
        return self.pointwise_conv2(x).transpose(1, 2)


class ConformerLayer(nn.Module):  # conformer_modules.py:60-214
    def __init__(self, d_model, d_ff, n_heads, conv_kernel_size):
        super().__init__()
        self.norm_feed_forward1 = nn.LayerNorm(d_model)
        self.feed_forward1 = ConformerFeedForward(d_model, d_ff)
        self.norm_conv = nn.LayerNorm(d_model)
        self.conv = ConformerConvolution(d_model, conv_kernel_size)
        self.norm_self_att = nn.LayerNorm(d_model)
        self.self_attn = RelPositionMultiHeadAttention(n_heads, d_model)
        self.norm_feed_forward2 = nn.LayerNorm(d_model)
        self.feed_forward2 = ConformerFeedForward(d_model, d_ff)
        self.norm_out = nn.LayerNorm(d_model)

    def forward(self, x, att_mask, pos_emb, pad_mask):
        residual = x
        residual = residual + self.feed_forward1(self.norm_feed_forward1(x)) * 0.5
        residual = residual + self.self_attn(self.norm_self_att(residual), att_mask, pos_emb)
        residual = residual + self.conv(self.norm_conv(residual), pad_mask)
        residual = residual + self.feed_forward2(self.norm_feed_forward2(residual)) * 0.5
        return self.norm_out(residual)


class ConformerEncoder(nn.Module):
    """A/modules/conformer_encoder.py:259-662 + _create_masks :686-740; dropout/stochastic depth off (p=0)."""

    def __init__(self, feat_in, n_layers, d_model, n_heads, ff_expansion_factor=4, conv_kernel_size=31,
                 pos_emb_max_len=5000):
        super().__init__()
        self.pre_encode = ConvSubsampling(feat_in, d_model, d_model)
        self.pos_enc = RelPositionalEncoding(d_model, pos_emb_max_len, xscale=math.sqrt(d_model))
        self.layers = nn.ModuleList(
            [ConformerLayer(d_model, d_model * ff_expansion_factor, n_heads, conv_kernel_size) for _ in range(n_layers)])
        self.encoder_frozen_till = -1

    def forward(self, audio_signal, length):
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.encoder_frozen_till > 0):  # :510-512
            x = torch.transpose(audio_signal, 1, 2)
            x, length = self.pre_encode(x, length)
            length = length.to(torch.int64)
            T = x.size(1)
            x, pos_emb = self.pos_enc(x)
            valid = torch.arange(0, T).expand(length.size(0), -1) < length.unsqueeze(-1)
            att_mask = ~(valid.unsqueeze(1).repeat([1, T, 1]) & valid.unsqueeze(1).repeat([1, T, 1]).transpose(1, 2))
            pad_mask = ~valid
        for lth, layer in enumerate(self.layers):
            with torch.set_grad_enabled(torch.is_grad_enabled() and not self.encoder_frozen_till > lth):  # :577
                x = layer(x, att_mask, pos_emb, pad_mask)
        return torch.transpose(x, 1, 2), length


# ------------------------------------------------------------------------------------------------ decoder / joint / ctc
class LSTMDropout(nn.Module):  # C/parts/rnn.py:151-235 (forget gate bias 1.0, hidden-hidden forget bias * 0)
    def __init__(self, input_size, hidden_size, forget_gate_bias=1.0):
        super().__init__()
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=1)
        with torch.no_grad():
            self.lstm.bias_ih_l0[hidden_size:2 * hidden_size].fill_(forget_gate_bias)
            self.lstm.bias_hh_l0[hidden_size:2 * hidden_size] *= 0.0

    def forward(self, x, h=None):
        return self.lstm(x, h)


class RNNTDecoder(nn.Module):
    """A/modules/rnnt.py:524-792: Embedding(V_tot+1, H, padding_idx=blank) -> prepend zero SOS -> LSTM."""

    def __init__(self, vocab_size, pred_hidden):
        super().__init__()
        self.blank_idx = vocab_size
        self.prediction = nn.ModuleDict({"embed": nn.Embedding(vocab_size + 1, pred_hidden, padding_idx=vocab_size),
                                         "dec_rnn": LSTMDropout(pred_hidden, pred_hidden)})

    def forward(self, targets, target_length):
        y = self.prediction["embed"](targets)
        B, U, H = y.shape
        y = torch.cat([torch.zeros((B, 1, H), dtype=y.dtype), y], dim=1).contiguous()
        g, _ = self.prediction["dec_rnn"](y.transpose(0, 1))
        return g.transpose(0, 1).transpose(1, 2), target_length  # (B, D, U+1)


class RNNTJoint(nn.Module):
    """A/modules/rnnt.py:1175-1710: fused joint + loss per sub-batch of `fused_batch_size`, language-selected head,
    MAS/LwF stashes.  `gpu_semantics=True` reproduces the CUDA branch (no log_softmax on the stashed tensor and on
    the loss input, rnnt.py:1651-1656)."""

    def __init__(self, enc_hidden, pred_hidden, joint_hidden, languages, vocab_per_lang, fused_batch_size=4):
        super().__init__()
        self.pred = nn.Linear(pred_hidden, joint_hidden)
        self.enc = nn.Linear(enc_hidden, joint_hidden)
        final = nn.ModuleDict({l: nn.Linear(joint_hidden, vocab_per_lang + 1) for l in languages})
        self.joint_net = nn.Sequential(nn.ReLU(inplace=True), nn.Dropout(p=0.0), final)
        self._fused_batch_size = fused_batch_size
        self.store_sub_enc = self.store_sub_logits = self.detach_sub_enc = False
        self.store_list = []
        self.gpu_semantics = True

    def joint(self, f, g, language_ids):
        inp = torch.relu(self.enc(f).unsqueeze(2) + self.pred(g).unsqueeze(1))
        if len(set(language_ids)) == 1:
            return self.joint_net[-1][language_ids[0]](inp)
        return torch.stack([self.joint_net[-1][l](x) for x, l in zip(inp, language_ids)])

    def forward(self, encoder_outputs, decoder_outputs, encoder_lengths, transcripts, transcript_lengths, loss_fn,
                language_ids):
        enc = encoder_outputs.transpose(1, 2)
        dec = decoder_outputs.transpose(1, 2)
        losses, tls, stash = [], [], []
        B = enc.size(0)
        for begin in range(0, B, self._fused_batch_size):
            end = min(begin + self._fused_batch_size, B)
            sl = slice(begin, end)
            max_t = int(encoder_lengths[sl].max())
            max_u = int(transcript_lengths[sl].max())
            sub_enc = enc[sl, :max_t]
            sub_dec = dec[sl, :max_u + 1]
            logits = self.joint(sub_enc, sub_dec, language_ids[sl])
            sub_joint = logits if self.gpu_semantics else logits.log_softmax(-1)
            if self.store_sub_enc:
                stash.append(sub_joint.detach().clone() if self.detach_sub_enc else sub_joint.clone())
            if self.store_sub_logits:
                stash.append(logits.detach().clone() if self.detach_sub_enc else logits.clone())
            losses.append(loss_fn(sub_joint, transcripts[sl, :max_u], encoder_lengths[sl], transcript_lengths[sl]))
            tls.append(transcript_lengths[sl])
        if self.store_sub_enc or self.store_sub_logits:
            self.store_list = stash
        return torch.cat(losses, 0).mean()  # A/losses/rnnt.py:424-429 'mean_batch' after concatenation


class _RNNTOracleFn(torch.autograd.Function):
    """Transducer loss through oracle/rnnt_ref.c (gradient w.r.t. logits, = CPU path of the reference:
    explicit log_softmax + CPURNNT, K/rnnt_pytorch.py:411-437)."""

    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank):
        r = rnnt_oracle.rnnt_loss(acts.detach().numpy(), labels.numpy(), act_lens.numpy(), label_lens.numpy(), blank)
        ctx.grads = torch.from_numpy(r["grads"])
        return torch.from_numpy(r["costs"])

    @staticmethod
    def backward(ctx, go):
        return ctx.grads * go.view(-1, 1, 1, 1), None, None, None, None


class ConvASRDecoder(nn.Module):
    """A/modules/conv_asr.py:402-490: Conv1d(d -> 22*256+1, k=1) over ALL languages, masked_select of the
    batch language's 257 columns, log_softmax."""

    def __init__(self, feat_in, languages, vocab_per_lang):
        super().__init__()
        n = len(languages) * vocab_per_lang + 1
        self.decoder_layers = nn.Sequential(nn.Conv1d(feat_in, n, kernel_size=1, bias=True))
        nn.init.xavier_uniform_(self.decoder_layers[0].weight, gain=1.0)  # conv_asr.py:447 init_mode
        self.language_masks = {}
        for i, l in enumerate(languages):
            m = [False] * n
            m[i * vocab_per_lang:(i + 1) * vocab_per_lang] = [True] * vocab_per_lang
            m[-1] = True
            self.language_masks[l] = m
        self.return_logits_ = False
        self.decoder_logits = None

    def forward(self, encoder_output, language_ids):
        out = self.decoder_layers(encoder_output).transpose(1, 2)
        mask = torch.tensor([self.language_masks[l] for l in language_ids], dtype=torch.bool).unsqueeze(1)
        mask = mask.repeat(1, out.shape[1], 1)
        out = torch.masked_select(out, mask).view(out.shape[0], out.shape[1], -1)
        if self.return_logits_:
            self.decoder_logits = out.clone()
        return F.log_softmax(out, dim=-1)


# ------------------------------------------------------------------------------------------------ model
class OracleHybridModel(nn.Module):
    """EncDecHybridRNNTCTCBPEModel as the CL scripts use it (A/models/hybrid_rnnt_ctc_models.py:859-930,
    rnnt_models.py:606-655, hybrid_rnnt_ctc_bpe_models.py:43-170)."""

    def __init__(self, d_model=144, n_layers=16, n_heads=4, pred_hidden=320, joint_hidden=320, languages=None,
                 vocab_per_lang=256, fused_batch_size=4, ctc_loss_weight=0.3, conv_kernel_size=31, feat_in=80):
        super().__init__()
        self.languages = list(languages or LANGS22)
        self.preprocessor = nn.Module()  # AudioToMelSpectrogramPreprocessor.featurizer (audio_preprocessing.py:88-94)
        self.preprocessor.featurizer = FilterbankFeatures(nfilt=feat_in)
        self.encoder = ConformerEncoder(feat_in, n_layers, d_model, n_heads, conv_kernel_size=conv_kernel_size)
        self.decoder = RNNTDecoder(len(self.languages) * vocab_per_lang, pred_hidden)
        self.joint = RNNTJoint(d_model, pred_hidden, joint_hidden, self.languages, vocab_per_lang, fused_batch_size)
        self.ctc_decoder = ConvASRDecoder(d_model, self.languages, vocab_per_lang)
        self.blank = vocab_per_lang
        self.ctc_loss_weight = ctc_loss_weight

    def rnnt_loss_fn(self, logits, targets, il, tl):
        return _RNNTOracleFn.apply(logits.float().contiguous(), targets.contiguous().long(), il.long(), tl.long(), self.blank)

    def forward(self, input_signal, input_signal_length, spec_aug=None, dither_noise=None):
        feats, flen = self.preprocessor.featurizer(input_signal, input_signal_length, dither_noise)
        if spec_aug is not None and self.training:
            feats = spec_augment_apply(feats, *spec_aug)
        return self.encoder(feats, flen)

    def training_step(self, batch, lang_ids, return_probs=False, spec_aug=None, dither_noise=None):
        signal, signal_len, transcript, transcript_len = batch
        encoded, encoded_len = self.forward(signal, signal_len, spec_aug, dither_noise)
        decoder, target_length = self.decoder(transcript, transcript_len)
        loss_value = self.joint(encoded, decoder, encoded_len, transcript, transcript_len, self.rnnt_loss_fn, lang_ids)
        log_probs = self.ctc_decoder(encoded, lang_ids)
        ctc = F.ctc_loss(log_probs.transpose(1, 0), transcript.long(), encoded_len.long(), transcript_len.long(),
                         blank=self.blank, reduction='none', zero_infinity=True).mean()  # A/losses/ctc.py:45-82
        monitor = {'train_rnnt_loss': loss_value.item(), 'train_ctc_loss': ctc.item()}
        loss = (1 - self.ctc_loss_weight) * loss_value + self.ctc_loss_weight * ctc  # :902
        monitor['train_loss'] = loss.item()
        if return_probs:
            return loss, monitor, log_probs
        return loss, monitor


def freeze_layer(model, n):
    """R/utils.py:246-263: whole encoder frozen, layers with index > n unfrozen again."""
    for p in model.encoder.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.encoder.layers):
        if i > n:
            for p in layer.parameters():
                p.requires_grad = True
    model.encoder.encoder_frozen_till = n


# ------------------------------------------------------------------------------------------------ CL arithmetic
def get_params(model):  # R/utils.py:273-279
    return {n: p for n, p in model.named_parameters() if p.requires_grad}


def ewc_penalty_grads(e_lambda, fish, curr, ckpt):
    """R/cl_baseline_ewc.py:69-81."""
    result, avg, n = {}, 0.0, 0
    for k in curr:
        result[k] = e_lambda * 2 * fish[k] * (curr[k] - ckpt[k])
        avg += torch.mean(torch.abs(result[k]))
        n += 1
    return result, float(avg) / n


def ewc_fisher_accumulate(fish, grads, loss):
    """R/cl_baseline_ewc.py:245-255: F += mean(loss) * g^2."""
    w = torch.mean(loss.detach().clone())
    for k in grads:
        fish[k] += w * grads[k] ** 2


def ewc_fisher_finish(main_fish, fish, total_ds, e_gamma):
    """R/cl_baseline_ewc.py:267-280."""
    for k in fish:
        fish[k] /= total_ds
    if main_fish is None:
        return fish
    for k in fish:
        main_fish[k] *= e_gamma
        main_fish[k] += fish[k]
    return main_fish


def mas_penalty(importance, params, ckpt):
    """R/cl_baseline_mas.py:70-75; the caller adds `mas_lambda * penalty` to the loss (:231-234)."""
    loss = 0
    for n, p in params.items():
        loss = loss + torch.sum(importance[n] * (p - ckpt[n]) ** 2)
    return loss


def mas_importance_loss(store_list, ctc_logits, mas_ctx):
    """R/cl_baseline_mas.py:258-265 (stashed tensors are raw logits: store_sub_logits / return_logits_)."""
    decoder_logits = (ctc_logits.flatten(end_dim=-2) ** 2).sum(dim=-1).mean()
    rnn_logits = 0
    for i in store_list:
        rnn_logits = rnn_logits + (i.flatten(end_dim=-2) ** 2).sum(dim=-1).mean()
    rnn_logits = rnn_logits / len(store_list)
    return rnn_logits * (1 - mas_ctx) + decoder_logits * mas_ctx


def mas_importance_accumulate(importance, grads):
    """R/cl_baseline_mas.py:267-270."""
    for n, g in grads.items():
        if g is not None:
            importance[n] += g.abs().detach()


def lwf_kd_loss(loss, prob, prob_, pred_store_list, store_list, kd, kd_ctx):
    """R/cl_baseline_lwf.py:242-264.  prob/prob_: student/teacher CTC log-probs [B,T',257]; store lists: the
    per-sub-batch joint tensors (raw logits under GPU semantics, rnnt.py:1651-1656)."""
    ctc_kd_loss = F.kl_div(prob, prob_.exp(), reduction='batchmean')
    rnnt_kd = 0
    for i, j in zip(store_list, pred_store_list):
        rnnt_kd = rnnt_kd + F.kl_div(j, i.exp(), reduction='batchmean')
    rnnt_kd = rnnt_kd / len(store_list)
    total = loss * (1 - kd) + kd * ((1 - kd_ctx) * rnnt_kd + kd_ctx * ctc_kd_loss)
    return total, rnnt_kd, ctc_kd_loss


# ------------------------------------------------------------------------------------------------------------------
# Greedy decoding + WER, plain per-utterance restatements for the parity tests.
def greedy_rnnt_decode_ref(model, encoded, encoded_len, lang, max_symbols=10):
    """Per-utterance, frame-by-frame restatement of GreedyBatchedRNNTInfer._greedy_decode_blank_as_pad_loop_frames
    (rnnt_greedy_decoding.py:711-909).  Processing utterances one at a time is equivalent to the batched loop: the
    per-frame blank mask is sticky per utterance and an utterance's state only advances on its own non-blank symbols.
    SOS and "no symbol yet" are the zero embedding (blank_as_pad, rnnt.py:524-792)."""
    dec, joint = model.decoder, model.joint
    head = joint.joint_net[-1][lang]
    blank = head.weight.shape[0] - 1
    H = dec.prediction["embed"].weight.shape[1]
    out = []
    with torch.no_grad():
        for b in range(encoded.shape[0]):
            f_all = joint.enc(encoded[b].transpose(0, 1).float())                    # [T, Hj]
            hidden, last, toks = None, None, []
            for t in range(int(encoded_len[b])):
                symbols = 0
                while symbols < max_symbols:
                    y = torch.zeros(1, 1, H) if last is None else dec.prediction["embed"](torch.tensor([[last]]))
                    g, hp = dec.prediction["dec_rnn"](y.transpose(0, 1), hidden)      # [1,1,H]
                    logits = head(torch.relu(f_all[t] + joint.pred(g[0, 0].float())))
                    k = int(logits.argmax())
                    if k == blank:
                        break
                    toks.append(k); last = k; hidden = hp; symbols += 1
            out.append(toks)
    return out


def greedy_ctc_decode_ref(log_probs, lengths, blank):
    out = []
    for b in range(log_probs.shape[0]):
        prev, toks = None, []
        for t in range(int(lengths[b])):
            k = int(log_probs[b, t].argmax())
            if k != blank and k != prev:
                toks.append(k)
            prev = k
        out.append(toks)
    return out
