"""Conformer encoder with the reference's module tree and parameter names (A/modules/conformer_encoder.py:259-662,
A/parts/submodules/{conformer_modules,multi_head_attention,subsampling}.py) on MI355X.

Differences in HOW (not WHAT): no [B,T',T'] mask tensor and no all_reduce(MAX)+.item() per forward
(conformer_encoder.py:664-675) -- lengths only; rel-shift is index arithmetic; frame counts are integers;
dense projections run in cfg.compute_dtype on the matrix cores, norms/softmax/BN in fp32.
"""
import math
from contextlib import nullcontext

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def subsampled_length(n):
    """Two stride-2 k=3 p=1 convolutions: floor((n-1)/2)+1 twice == subsampling.py:566-576 (float32 there;
    equality for every n up to 30 s is asserted in tests/test_oracle_step.py::test_frame_count_rule)."""
    n = (n - 1) // 2 + 1
    return (n - 1) // 2 + 1


SUBSAMPLED_EVENT = {}   # device index -> event recorded behind the subsampling of the most recent encoder forward

class FastLinear(nn.Linear):
    """nn.Linear (same parameter names) whose forward/backward run on the HIP GEMM path inside the bf16 region."""

    def forward(self, x):
        from .ops import fast
        return fast.linear(x, self.weight, self.bias)


class ConvSubsampling(nn.Module):
    def __init__(self, feat_in, feat_out, conv_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, conv_channels, 3, 2, 1), nn.ReLU(True),
                                  nn.Conv2d(conv_channels, conv_channels, 3, 2, 1), nn.ReLU(True))
        self.out = nn.Linear(conv_channels * subsampled_length(feat_in), feat_out)

    def forward(self, x, lengths):
        lengths = subsampled_length(lengths)
        x = self.conv(x.unsqueeze(1))
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).reshape(b, t, -1))
        return x, lengths


class RelPositionalEncoding(nn.Module):
    """multi_head_attention.py:935-979: table row r <-> relative position (Lmax-1-r)."""

    def __init__(self, d_model, max_len, xscale, dropout_rate, dropout_rate_emb=0.0):
        super().__init__()
        self.d_model, self.xscale, self.max_len = d_model, xscale, max_len
        self.dropout = nn.Dropout(dropout_rate)
        self.dropout_emb = nn.Dropout(dropout_rate_emb) if dropout_rate_emb > 0 else None
        self.register_buffer("pe", self._table(max_len), persistent=False)

    def _table(self, length):
        positions = torch.arange(length - 1, -length, -1, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d_model))
        pe = torch.zeros(positions.size(0), self.d_model)
        pe[:, 0::2] = torch.sin(positions * div_term)
        pe[:, 1::2] = torch.cos(positions * div_term)
        return pe.unsqueeze(0)

    def extend_pe(self, length, device):
        if self.pe.size(1) < 2 * length - 1:
            self.pe = self._table(length).to(device)

    def pos_emb_for(self, T):
        centre = self.pe.size(1) // 2 + 1
        pos_emb = self.pe[:, centre - T: centre + T - 1]
        if self.dropout_emb is not None:
            pos_emb = self.dropout_emb(pos_emb)
        return pos_emb

    def forward(self, x):
        if self.xscale:
            x = x * self.xscale
        return self.dropout(x), self.pos_emb_for(x.size(1))


class RelPositionMultiHeadAttention(nn.Module):
    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = FastLinear(n_feat, n_feat)
        self.linear_k = FastLinear(n_feat, n_feat)
        self.linear_v = FastLinear(n_feat, n_feat)
        self.linear_out = FastLinear(n_feat, n_feat)
        self.linear_pos = FastLinear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.zeros(self.h, self.d_k))
        self.dropout_rate = dropout_rate

    def forward(self, x, lens, pos_emb):
        B, T, _ = x.shape
        q = self.linear_q(x).view(B, T, self.h, self.d_k).transpose(1, 2)
        k = self.linear_k(x).view(B, T, self.h, self.d_k).transpose(1, 2)
        v = self.linear_v(x).view(B, T, self.h, self.d_k).transpose(1, 2)
        p = self.linear_pos(pos_emb.to(x.dtype)).view(-1, self.h, self.d_k).transpose(0, 1)  # [h,2T-1,dk]
        ctx = ops.rel_pos_attention(q, k, v, p, self.pos_bias_u, self.pos_bias_v, lens, self.dropout_rate, self.training)
        return self.linear_out(ctx.transpose(1, 2).reshape(B, T, self.h * self.d_k))


class ConformerFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout):
        super().__init__()
        self.linear1 = FastLinear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = FastLinear(d_ff, d_model)

    def forward(self, x):
        return self.linear2(self.dropout(F.silu(self.linear1(x))))


class CausalConv1D(nn.Conv1d):
    """Parameter holder with the reference's name (depthwise_conv.weight/bias); the arithmetic runs in
    ops.glu_dwconv_bn_silu."""

    def __init__(self, ch, k):
        super().__init__(ch, ch, k, stride=1, padding=0, groups=ch, bias=True)


class ConformerConvolution(nn.Module):
    def __init__(self, d_model, kernel_size):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, 1)
        self.depthwise_conv = CausalConv1D(d_model, kernel_size)
        self.batch_norm = nn.BatchNorm1d(d_model)
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, 1)

    def forward(self, x, pad_mask):
        # k=1 convolutions are plain projections over the feature axis: keep [B,T,d] and skip two transposes
        from .ops import fast
        x2 = fast.linear(x, self.pointwise_conv1.weight, self.pointwise_conv1.bias)
        y = ops.glu_dwconv_bn_silu(x2, pad_mask, self.depthwise_conv.weight, self.depthwise_conv.bias, self.batch_norm,
                                   self.training)
        y = y.to(x.dtype)
        return fast.linear(y, self.pointwise_conv2.weight, self.pointwise_conv2.bias)


class ConformerLayer(nn.Module):
    def __init__(self, d_model, d_ff, n_heads, conv_kernel_size, dropout, dropout_att):
        super().__init__()
        self.fc_factor = 0.5
        self.norm_feed_forward1 = nn.LayerNorm(d_model)
        self.feed_forward1 = ConformerFeedForward(d_model, d_ff, dropout)
        self.norm_conv = nn.LayerNorm(d_model)
        self.conv = ConformerConvolution(d_model, conv_kernel_size)
        self.norm_self_att = nn.LayerNorm(d_model)
        self.self_attn = RelPositionMultiHeadAttention(n_heads, d_model, dropout_att)
        self.norm_feed_forward2 = nn.LayerNorm(d_model)
        self.feed_forward2 = ConformerFeedForward(d_model, d_ff, dropout)
        self.dropout = nn.Dropout(dropout)
        self.norm_out = nn.LayerNorm(d_model)

    # ---- MI355X no-autograd path: 14 HIP launches per block instead of ~150 ATen ones (frozen prefix, teacher, eval)
    def fast_supported(self, x):
        from .ops import fast
        d = x.shape[-1]
        bn = self.conv.batch_norm
        return (x.is_cuda and not torch.is_grad_enabled() and fast.gemm_supported(d, d) and d % 8 == 0 and d <= 1024
                and fast.bn_module_ok(bn) and self.conv.depthwise_conv.weight.shape[-1] <= 31)

    def forward_fast(self, x, y, lens, pos_emb, B, T, seed, next_ln=None):
        """x: fp32 residual stream [B*T, d] (updated in place), y: bf16 LN_ff1(x) [B*T, d].  Returns (x_out fp32,
        y_next bf16 or None): x_out = norm_out(residual); y_next = next_ln(x_out) when the next block's first
        LayerNorm is given (chained in the same launch)."""
        from .ops import fast
        d = x.shape[1]
        tr = self.training
        p = self.dropout.p if tr else 0.0
        ff1, ff2, att, cv = self.feed_forward1, self.feed_forward2, self.self_attn, self.conv
        fp8 = bool(getattr(self, "fp8_projections", False))
        # projections: bf16 MFMA GEMM, or e4m3 operands with per-row scales (activations quantised per call)
        # (fp8: block-scaled MX operands on the 2x-rate MFMA where K % 128 == 0 -- d_model = 256 / 512, d_ff -- else e4m3 with
        #  per-row scales on the bf16-rate instruction: fast.fp8_weights / fast.gemm_fp8_any)
        G = fast.gemm_fp8_any if fp8 else fast.gemm
        Wt = fast.fp8_weights if fp8 else fast.bf16_shadow
        ffn_fused = (not fp8) and fast.ffn_fused_supported(d, ff1.linear1.weight.shape[0])
        # 1/2 FFN
        if ffn_fused:   # LayerNorm + both projections + residual in one row-resident launch (csrc/ffn_fused.hip)
            fast.ffn_fused(x, self.norm_feed_forward1, ff1.linear1, ff1.linear2, self.fc_factor,
                           ff1.dropout.p if tr else 0.0, seed + 1, p, seed + 2)
        else:
            if y is None:
                y = fast.layernorm(x, self.norm_feed_forward1.weight, self.norm_feed_forward1.bias, self.norm_feed_forward1.eps)
            _, h = G(y, Wt(ff1.linear1.weight), ff1.linear1.bias, act=1,
                             dropout_p=ff1.dropout.p if tr else 0.0, seed=seed + 1)
            G(h, Wt(ff1.linear2.weight), ff1.linear2.bias, dropout_p=p, seed=seed + 2, alpha=self.fc_factor,
                      residual=x, out_f32=x, want_bf16=False)
        # self-attention
        y = fast.layernorm(x, self.norm_self_att.weight, self.norm_self_att.bias, self.norm_self_att.eps)
        _, qkv = G(y, Wt(att.linear_q.weight, att.linear_k.weight, att.linear_v.weight),
                           fast.f32_cat(att.linear_q.bias, att.linear_k.bias, att.linear_v.bias))
        _, pl = fast.gemm(pos_emb, fast.bf16_shadow(att.linear_pos.weight))
        if fast.attention_flash_supported(T, att.d_k):
            ctx = fast.relpos_attention_flash(qkv, pl, att.pos_bias_u, att.pos_bias_v, lens, B, T, att.h, att.d_k,
                                              att.dropout_rate if tr else 0.0, seed + 7)
        elif fast.attention_supported(T, att.d_k):
            ctx = fast.relpos_attention(qkv, pl, att.pos_bias_u, att.pos_bias_v, lens, B, T, att.h, att.d_k,
                                        att.dropout_rate if tr else 0.0, seed + 7)
        else:  # long inputs (T' > 384) / other head sizes: ATen composition
            qkv = qkv.view(B, T, 3, att.h, att.d_k)
            q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
            ctx = ops.rel_pos_attention(q, k, v, pl.view(-1, att.h, att.d_k).transpose(0, 1), att.pos_bias_u,
                                        att.pos_bias_v, lens, att.dropout_rate, tr)
            ctx = ctx.transpose(1, 2).reshape(B * T, d).contiguous()
        G(ctx, Wt(att.linear_out.weight), att.linear_out.bias, dropout_p=p, seed=seed + 3,
                  residual=x, out_f32=x, want_bf16=False)
        # convolution module
        y = fast.layernorm(x, self.norm_conv.weight, self.norm_conv.bias, self.norm_conv.eps)
        _, c2 = G(y, Wt(cv.pointwise_conv1.weight), cv.pointwise_conv1.bias)
        c3 = fast.glu_dwconv_bn_silu_fast(c2, lens, B, T, d, cv.depthwise_conv.weight, cv.depthwise_conv.bias,
                                          cv.batch_norm, tr)
        G(c3, Wt(cv.pointwise_conv2.weight), cv.pointwise_conv2.bias, dropout_p=p, seed=seed + 4,
                  residual=x, out_f32=x, want_bf16=False)
        # 1/2 FFN
        if ffn_fused:   # ... and norm_out in the same launch
            fast.ffn_fused(x, self.norm_feed_forward2, ff2.linear1, ff2.linear2, self.fc_factor,
                           ff2.dropout.p if tr else 0.0, seed + 5, p, seed + 6, ln2=self.norm_out)
            return x, None
        y = fast.layernorm(x, self.norm_feed_forward2.weight, self.norm_feed_forward2.bias, self.norm_feed_forward2.eps)
        _, h = G(y, Wt(ff2.linear1.weight), ff2.linear1.bias, act=1,
                         dropout_p=ff2.dropout.p if tr else 0.0, seed=seed + 5)
        G(h, Wt(ff2.linear2.weight), ff2.linear2.bias, dropout_p=p, seed=seed + 6, alpha=self.fc_factor,
                  residual=x, out_f32=x, want_bf16=False)
        # norm_out (+ the next block's first LayerNorm chained in registers)
        if next_ln is not None:
            y_next = fast.layernorm(x, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, out_f32=x,
                                    g2=next_ln.weight, b2=next_ln.bias)
        else:
            fast.layernorm(x, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, out_f32=x, want_bf16=False)
            y_next = None
        return x, y_next

    def forward(self, x, lens, pos_emb, pad_mask):
        residual = x
        residual = residual + self.dropout(self.feed_forward1(self.norm_feed_forward1(x))) * self.fc_factor
        residual = residual + self.dropout(self.self_attn(self.norm_self_att(residual), lens, pos_emb))
        residual = residual + self.dropout(self.conv(self.norm_conv(residual), pad_mask))
        residual = residual + self.dropout(self.feed_forward2(self.norm_feed_forward2(residual))) * self.fc_factor
        return self.norm_out(residual)


class ConformerEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.d_model
        self.d_model = d
        self.pre_encode = ConvSubsampling(cfg.feat_in, d, d)
        self.pos_enc = RelPositionalEncoding(d, cfg.pos_emb_max_len, math.sqrt(d), cfg.dropout_pre_encoder, cfg.dropout_emb)
        self.layers = nn.ModuleList([ConformerLayer(d, cfg.d_ff, cfg.n_heads, cfg.conv_kernel_size, cfg.dropout,
                                                    cfg.dropout_att) for _ in range(cfg.n_layers)])
        self.encoder_frozen_till = -1  # the reference's custom attribute (conformer_encoder.py:447)
        self.use_fast_path = True
        self.use_fused_blocks = True   # trainable blocks as single autograd nodes on the HIP kernels
        self.fast_seed = 0             # per-step dropout seed for the fused path (set by the model)
        self._pos_cache = {}           # T -> bf16 images of the position table slice (plain dict: not a buffer, not copied state)

    def _amp(self, x):
        if self.cfg.compute_dtype == "bf16" and x.is_cuda:
            return torch.autocast(device_type="cuda", dtype=torch.bfloat16)
        return nullcontext()

    def forward(self, audio_signal, length, subsampled_len=None):
        """audio_signal [B,feat,Tm] f32, length [B] i64 -> (encoded [B,d,T'], encoded_len [B] i64).
        `subsampled_len`: optional precomputed [B] i64 device tensor of the output lengths (the caller knew the lengths on
        the host: saves the half-dozen tiny integer kernels of the frame-count rule)."""
        from . import cl
        if cl._PENDING_OPTIMIZERS and any(p.requires_grad for p in self.pre_encode.parameters()):
            cl.flush_pending_updates()   # deferred data-parallel update: the subsampling weights are about to be read
        with self._amp(audio_signal):
            with (torch.no_grad() if self.encoder_frozen_till > 0 else nullcontext()):
                from .ops import fast
                pe = self.pre_encode
                scaled = False
                if (self.use_fast_path and self.cfg.compute_dtype == "bf16" and audio_signal.is_cuda
                        and not (torch.is_grad_enabled() and any(p.requires_grad for p in pe.parameters()))
                        and fast.subsample_supported(pe.conv[0].weight.shape[0], self.d_model, self.cfg.feat_in)):
                    # RelPositionalEncoding's x * sqrt(d) and its dropout ride in the epilogue of the subsampling's Linear
                    pen = self.pos_enc
                    p_pre = float(pen.dropout.p) if (self.training and pen.training) else 0.0
                    x = fast.conv_subsampling(audio_signal, pe.conv[0], pe.conv[2], pe.out, alpha=pen.xscale or 1.0, dropout_p=p_pre,
                                              seed=(self.fast_seed * 40503 + 977) & 0x7FFFFFFF)
                    scaled = True
                    length = subsampled_length(length) if subsampled_len is None else subsampled_len
                else:
                    x = audio_signal.transpose(1, 2)
                    x, length = pe(x, length)
                    if subsampled_len is not None:
                        length = subsampled_len
                length = length.to(torch.int64)
                T = x.size(1)
                if x.is_cuda:   # lets the caller start side-stream work behind the subsampling instead of beside it (model.training_step)
                    ev = SUBSAMPLED_EVENT[x.device.index] = torch.cuda.Event()   # (module-level: models stay deep-copyable)
                    ev.record(torch.cuda.current_stream(x.device))
                if scaled:
                    pos_emb = self.pos_enc.pos_emb_for(T)
                else:
                    x, pos_emb = self.pos_enc(x)
                pad_mask = None    # (only the ATen composition below reads it: built there)
            lth = 0
            n_layers = len(self.layers)
            # no-autograd prefix (frozen layers, or everything under torch.no_grad()): fused HIP path
            if self.use_fast_path and self.cfg.compute_dtype == "bf16" and x.is_cuda:
                n_fast = n_layers if not torch.is_grad_enabled() else max(0, min(n_layers, self.encoder_frozen_till))
                # layer `frozen_till` runs with autograd on in the reference but has frozen weights and a no-grad
                # input (R/utils.py:250-253 vs conformer_encoder.py:577): nothing to differentiate -> fused path too
                while (n_fast < n_layers and n_fast == self.encoder_frozen_till and not x.requires_grad
                       and not any(p.requires_grad for p in self.layers[n_fast].parameters())):
                    n_fast += 1
                if n_fast > 0:
                    with torch.no_grad():
                        if self.layers[0].fast_supported(x):
                            x, lth = self._fast_prefix(x, length, pos_emb, n_fast)
            # a deferred data-parallel optimizer update (cl.FusedAdamW.step) lands here: everything above read frozen
            # weights only, so the gradient all-reduce ran under it
            if cl._PENDING_OPTIMIZERS:
                cl.flush_pending_updates()
            # trainable suffix: one autograd node per block on the HIP kernels (ops/block.py), ATen composition otherwise
            blk_ok = False
            if (self.use_fast_path and self.use_fused_blocks and self.cfg.compute_dtype == "bf16" and x.is_cuda
                    and torch.is_grad_enabled() and lth < n_layers and self.training):
                from .ops import block
                B_, T_, d_ = x.shape
                blk_ok = all(block.block_supported(self.layers[l], x, T_) and self.encoder_frozen_till <= l
                             for l in range(lth, n_layers))
            if blk_ok:
                xr = x.float().reshape(B_ * T_, d_).contiguous()
                pe16 = self._pos_bf16(pos_emb, d_)[1]
                base = (self.fast_seed * 2654435761) & 0x7FFFFFFF
                for l in range(lth, n_layers):
                    xr = block.conformer_block(xr, self.layers[l], length, pe16, B_, T_, base + 16 * l)
                x = xr.view(B_, T_, d_)
            else:
                if lth < n_layers:
                    pad_mask = torch.arange(T, device=x.device)[None, :] >= length[:, None]
                for l in range(lth, n_layers):
                    with (torch.no_grad() if self.encoder_frozen_till > l else nullcontext()):
                        x = self.layers[l](x, length, pos_emb, pad_mask)
        return x.transpose(1, 2), length

    def _pos_bf16(self, pos_emb, d):
        """(bf16 [2T-1, d], bf16 zero-padded to a multiple of 8 rows) of the position table slice: a pure function of T (and of
        the table buffer), cached -- the cast and the padded copy used to be three launches per step."""
        if self.pos_enc.dropout_emb is not None and self.training:
            from .ops import block
            return pos_emb.reshape(-1, d).to(torch.bfloat16).contiguous(), block.pad_pos_emb(pos_emb, d)
        key = (pos_emb.shape[1], self.pos_enc.pe.data_ptr(), str(pos_emb.device))
        hit = self._pos_cache.get(key)
        if hit is None:
            from .ops import block
            if len(self._pos_cache) >= 8:
                self._pos_cache.pop(next(iter(self._pos_cache)))
            with torch.no_grad():
                padded = block.pad_pos_emb(pos_emb, d)
                hit = self._pos_cache[key] = (padded[:pos_emb.shape[1]], padded)
        return hit

    def _fast_prefix(self, x, length, pos_emb, n_fast):
        from .ops import fast
        B, T, d = x.shape
        xr = x.float().reshape(B * T, d).contiguous()
        pe = self._pos_bf16(pos_emb, d)[0]
        base = (self.fast_seed * 2654435761) & 0x7FFFFFFF
        l0 = self.layers[0]
        bn_ok = all(l.conv.batch_norm.track_running_stats for l in self.layers[:n_fast])
        same_mode = all(l.training == l0.training for l in self.layers[:n_fast])
        fp8 = bool(getattr(self.cfg, "fp8_frozen_prefix", False)) and d % 16 == 0
        for l in self.layers[:n_fast]:
            l.fp8_projections = fp8
        # native executor: one C call for the whole prefix; with SyncBatchNorm over several ranks one call per block boundary,
        # the all-reduce of the BatchNorm sums in between (ops/fast.conformer_prefix).  fp8 projections: per-op path below.
        if ((fast.attention_flash_supported(T, l0.self_attn.d_k) or fast.attention_supported(T, l0.self_attn.d_k)) and bn_ok
                and same_mode and not fp8):
            # native executor: one C call enqueues the 14 kernels of every block (csrc/block_exec.hip)
            fast.conformer_prefix(list(self.layers[:n_fast]), xr, pe, length, B, T, base, 16, l0.training)
            return xr.view(B, T, d), n_fast
        y = None   # (the first module of a block applies its LayerNorm itself unless the previous block chained it)
        for l in range(n_fast):
            nxt = self.layers[l + 1].norm_feed_forward1 if l + 1 < n_fast else None
            xr, y = self.layers[l].forward_fast(xr, y, length, pe, B, T, base + 16 * l, nxt)
        return xr.view(B, T, d), n_fast
