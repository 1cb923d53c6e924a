"""One trainable Conformer block as a single autograd node on the HIP kernels.

Forward = the fused no-autograd path of encoder.ConformerLayer.forward_fast (GEMMs with fused epilogues, LayerNorm,
GLU+depthwise conv+BatchNorm+SiLU, rel-pos attention) keeping the intermediates the backward needs; backward = manual
chain rule: bf16 library GEMMs for the data gradients, batched split-K GEMMs for the weight gradients, and the
kernels of csrc/encoder_bwd.hip for everything in between; the attention core is the key-tiled pair
csrc/attention_flash.hip / attention_flash_bwd.hip (any T, no [T,T] matrices in HBM).

Semantics: ConformerLayer.forward, A/parts/submodules/conformer_modules.py:141-214 (bf16 projections, fp32 residual
stream / norms / BatchNorm statistics), dropout masks are counter-based and regenerated in the backward.
"""
import torch

from .. import _lib
from . import fast


def _ptr(t):
    return _lib.ptr(t)


def _ln_bwd(x, ln, dy_f32=None, dy_bf16=None, dx_in=None):
    N, d = x.shape
    dev = x.device
    dx = torch.empty(N, d, dtype=torch.float32, device=dev)
    dgb = torch.empty(2, d, dtype=torch.float32, device=dev)
    dy = dy_f32 if dy_f32 is not None else dy_bf16
    L = _lib.lib()
    st = L.ia_layernorm_bwd(_ptr(x), x.stride(0), _ptr(dy_f32), _ptr(dy_bf16), dy.stride(0), N, d, _ptr(ln.weight),
                            float(ln.eps), _ptr(dx_in), _ptr(dx), d, _ptr(dgb[0]), _ptr(dgb[1]),
                            _ptr(fast.scratch(dev, L.ia_layernorm_bwd_scratch_elems(N, d))), _lib.stream_ptr())
    _lib.check(st, "ia_layernorm_bwd")
    return dx, dgb[0], dgb[1]


def _branch_grad(dxr, alpha, p, seed):
    N, d = dxr.shape
    out = torch.empty(N, d, dtype=torch.bfloat16, device=dxr.device)
    st = _lib.lib().ia_scale_dropout_bf16(_ptr(dxr), N, d, float(alpha), float(p), int(seed) & 0xFFFFFFFF, _ptr(out),
                                          _lib.stream_ptr())
    _lib.check(st, "ia_scale_dropout_bf16")
    return out


def _lin_bwd(dyb, xb, wb, need_dx=True):
    """dyb [M,n] bf16, xb [M,k] bf16, wb [n,k] bf16 -> (dx bf16 [M,k] or None, dW f32 [n,k], db f32 [n]).
    Data gradient: library GEMM; weight + bias gradient: csrc/gemm_tn.hip (transposing LDS reads, split-K partial tiles)."""
    M, n = dyb.shape
    k = xb.shape[1]
    dx = torch.mm(dyb, wb) if need_dx else None
    L = _lib.lib()
    buf = torch.empty(n * k + n, dtype=torch.float32, device=dyb.device)  # dW | db contiguous: one finishing pass
    dW, db = buf[:n * k].view(n, k), buf[n * k:]
    st = L.ia_gemm_tn_bf16(_ptr(dyb), dyb.stride(0), _ptr(xb), xb.stride(0), M, n, k, _ptr(dW), _ptr(db),
                           _ptr(fast.scratch(dyb.device, L.ia_gemm_tn_scratch_elems(M, n, k))), _lib.stream_ptr())
    _lib.check(st, "ia_gemm_tn_bf16")
    return dx, dW, db


def _silu_dropout(h_pre, p, seed):
    M, N = h_pre.shape
    out = torch.empty_like(h_pre)
    _lib.check(_lib.lib().ia_silu_dropout(_ptr(h_pre), M, N, float(p), int(seed) & 0xFFFFFFFF, _ptr(out), _lib.stream_ptr()),
               "ia_silu_dropout")
    return out


def _silu_dropout_bwd(h_pre, dh, p, seed):
    M, N = h_pre.shape
    out = torch.empty_like(h_pre)
    _lib.check(_lib.lib().ia_silu_dropout_bwd(_ptr(h_pre), _ptr(dh), M, N, float(p), int(seed) & 0xFFFFFFFF, _ptr(out),
                                              _lib.stream_ptr()), "ia_silu_dropout_bwd")
    return out


# True: the block's backward adds parameter gradients into existing fp32 .grad buffers itself (and reports None to
# autograd).  Set False when differentiating with torch.autograd.grad(..., block parameters).
DIRECT_ACCUMULATE = True


def block_supported(layer, x2d, T):
    d = x2d.shape[-1]
    bn = layer.conv.batch_norm
    return (x2d.is_cuda and d % 8 == 0 and d <= 1024 and fast.bn_module_ok(bn)
            and layer.conv.depthwise_conv.weight.shape[-1] <= 31 and fast.attention_flash_supported(T, layer.self_attn.d_k))


class _ConformerBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, lens, pe, B, T, seed, *params):
        L = _lib.lib()
        N, d = x.shape
        dev = x.device
        tr = layer.training
        p = float(layer.dropout.p) if tr else 0.0
        ff1, ff2, att, cv = layer.feed_forward1, layer.feed_forward2, layer.self_attn, layer.conv
        pff1 = float(ff1.dropout.p) if tr else 0.0
        pff2 = float(ff2.dropout.p) if tr else 0.0
        patt = float(att.dropout_rate) if tr else 0.0
        W = dict(w1=fast.bf16_shadow(ff1.linear1.weight), w2=fast.bf16_shadow(ff1.linear2.weight),
                 wqkv=fast.bf16_shadow(att.linear_q.weight, att.linear_k.weight, att.linear_v.weight),
                 wpos=fast.bf16_shadow(att.linear_pos.weight), wo=fast.bf16_shadow(att.linear_out.weight),
                 wp1=fast.bf16_shadow(cv.pointwise_conv1.weight), wp2=fast.bf16_shadow(cv.pointwise_conv2.weight),
                 w1b=fast.bf16_shadow(ff2.linear1.weight), w2b=fast.bf16_shadow(ff2.linear2.weight))
        new = lambda: torch.empty(N, d, dtype=torch.float32, device=dev)
        x0 = x.contiguous()
        # 1/2 FFN
        y1 = fast.layernorm(x0, layer.norm_feed_forward1.weight, layer.norm_feed_forward1.bias, layer.norm_feed_forward1.eps)
        _, h1p = fast.gemm(y1, W["w1"], ff1.linear1.bias)
        h1 = _silu_dropout(h1p, pff1, seed + 1)
        x1, _ = fast.gemm(h1, W["w2"], ff1.linear2.bias, dropout_p=p, seed=seed + 2, alpha=layer.fc_factor, residual=x0,
                          out_f32=new(), want_bf16=False)
        # self-attention
        y2 = fast.layernorm(x1, layer.norm_self_att.weight, layer.norm_self_att.bias, layer.norm_self_att.eps)
        _, qkv = fast.gemm(y2, W["wqkv"], fast.f32_cat(att.linear_q.bias, att.linear_k.bias, att.linear_v.bias))
        _, pl = fast.gemm(pe, W["wpos"])
        ctxv, lse = fast.relpos_attention_flash(qkv, pl, att.pos_bias_u, att.pos_bias_v, lens, B, T, att.h, att.d_k, patt, seed + 7,
                                                want_lse=True)
        x2, _ = fast.gemm(ctxv, W["wo"], att.linear_out.bias, dropout_p=p, seed=seed + 3, residual=x1, out_f32=new(),
                          want_bf16=False)
        # convolution module
        y3 = fast.layernorm(x2, layer.norm_conv.weight, layer.norm_conv.bias, layer.norm_conv.eps)
        _, c2 = fast.gemm(y3, W["wp1"], cv.pointwise_conv1.bias)
        bn = cv.batch_norm
        use_batch = bool(tr or not bn.track_running_stats)
        c3, z, sums = fast.glu_dwconv_bn_silu_fast(c2, lens, B, T, d, cv.depthwise_conv.weight, cv.depthwise_conv.bias, bn, tr,
                                                   keep=True)   # (SyncBatchNorm: sums all-reduced inside)
        x3, _ = fast.gemm(c3, W["wp2"], cv.pointwise_conv2.bias, dropout_p=p, seed=seed + 4, residual=x2, out_f32=new(),
                          want_bf16=False)
        # 1/2 FFN
        y4 = fast.layernorm(x3, layer.norm_feed_forward2.weight, layer.norm_feed_forward2.bias, layer.norm_feed_forward2.eps)
        _, h4p = fast.gemm(y4, W["w1b"], ff2.linear1.bias)
        h4 = _silu_dropout(h4p, pff2, seed + 5)
        x4, _ = fast.gemm(h4, W["w2b"], ff2.linear2.bias, dropout_p=p, seed=seed + 6, alpha=layer.fc_factor, residual=x3,
                          out_f32=new(), want_bf16=False)
        out = new()
        fast.layernorm(x4, layer.norm_out.weight, layer.norm_out.bias, layer.norm_out.eps, out_f32=out, want_bf16=False)
        if not use_batch:
            raise RuntimeError("trainable fused block expects train-mode BatchNorm (batch statistics)")
        ctx.S = dict(x0=x0, y1=y1, h1p=h1p, h1=h1, x1=x1, y2=y2, qkv=qkv, pl=pl, ctxv=ctxv, lse=lse, x2=x2, y3=y3, c2=c2, z=z, sums=sums,
                     c3=c3, x3=x3, y4=y4, h4p=h4p, h4=h4, x4=x4, W=W, pe=pe, lens=lens)
        ctx.meta = (layer, B, T, seed, p, pff1, pff2, patt, [n for n, _ in layer.named_parameters()],
                    [q.requires_grad for q in params], params)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        S = ctx.S
        ctx.S = None
        layer, B, T, seed, p, pff1, pff2, patt, names, req, params = ctx.meta
        W = S["W"]
        ff1, ff2, att, cv = layer.feed_forward1, layer.feed_forward2, layer.self_attn, layer.conv
        d = S["x0"].shape[1]
        N = B * T
        dev = dout.device
        G = {}
        dout = dout.float().contiguous()
        # norm_out
        dx4, G["norm_out.weight"], G["norm_out.bias"] = _ln_bwd(S["x4"], layer.norm_out, dy_f32=dout)
        # feed_forward2
        dB = _branch_grad(dx4, layer.fc_factor, p, seed + 6)
        dh, G["feed_forward2.linear2.weight"], G["feed_forward2.linear2.bias"] = _lin_bwd(dB, S["h4"], W["w2b"])
        dhp = _silu_dropout_bwd(S["h4p"], dh, pff2, seed + 5)
        dy, G["feed_forward2.linear1.weight"], G["feed_forward2.linear1.bias"] = _lin_bwd(dhp, S["y4"], W["w1b"])
        dx3, G["norm_feed_forward2.weight"], G["norm_feed_forward2.bias"] = _ln_bwd(S["x3"], layer.norm_feed_forward2,
                                                                                  dy_bf16=dy, dx_in=dx4)
        del dx4, dh, dhp
        # convolution module
        dB = _branch_grad(dx3, 1.0, p, seed + 4)
        dc3, dWp2, G["conv.pointwise_conv2.bias"] = _lin_bwd(dB, S["c3"], W["wp2"])
        G["conv.pointwise_conv2.weight"] = dWp2.unsqueeze(-1)
        bn = cv.batch_norm
        S12 = torch.empty(2, d, dtype=torch.float32, device=dev)
        dz = torch.empty(N, d, dtype=torch.float32, device=dev)
        sums = S["sums"]
        group = fast.bn_sync_group(bn)
        bn_scr = _ptr(fast.scratch(dev, L.ia_bn_silu_bwd_scratch_elems(N, d)))
        if group is None:
            _lib.check(L.ia_bn_silu_bwd(_ptr(S["z"]), _ptr(dc3), N, d, _ptr(sums[:d]), _ptr(sums[d:2 * d]), _ptr(bn.weight),
                                        _ptr(bn.bias), float(bn.eps), _ptr(S12[0]), _ptr(S12[1]), _ptr(dz), bn_scr,
                                        _lib.stream_ptr()), "ia_bn_silu_bwd")
        else:
            # SyncBatchNorm: the backward's own exchange -- local S1 | S2 are this rank's d beta | d gamma; dz needs the global
            # sums, brought to the n_local scale the kernels divide by
            import torch.distributed as dist
            _lib.check(L.ia_bn_silu_bwd_reduce(_ptr(S["z"]), _ptr(dc3), N, d, _ptr(sums[:d]), _ptr(sums[d:2 * d]), _ptr(bn.weight),
                                               _ptr(bn.bias), float(bn.eps), _ptr(S12[0]), _ptr(S12[1]), bn_scr,
                                               _lib.stream_ptr()), "ia_bn_silu_bwd_reduce")
            Sg = torch.cat([S12.reshape(-1), torch.full((1,), float(N), dtype=torch.float32, device=dev)])
            dist.all_reduce(Sg, group=group)
            Sg = Sg[:2 * d] * (float(N) / Sg[2 * d])
            _lib.check(L.ia_bn_silu_bwd_apply(_ptr(S["z"]), _ptr(dc3), N, d, _ptr(sums[:d]), _ptr(sums[d:2 * d]), _ptr(bn.weight),
                                              _ptr(bn.bias), float(bn.eps), _ptr(Sg[:d]), _ptr(Sg[d:]), _ptr(dz),
                                              _lib.stream_ptr()), "ia_bn_silu_bwd_apply")
        G["conv.batch_norm.bias"], G["conv.batch_norm.weight"] = S12[0], S12[1]
        ksz = cv.depthwise_conv.weight.shape[-1]
        w2 = cv.depthwise_conv.weight.detach().float().reshape(d, ksz).contiguous()
        dG = torch.empty(N, d, dtype=torch.float32, device=dev)
        _lib.check(L.ia_dwconv_time(_ptr(dz), B, T, d, ksz, _ptr(w2), None, 1, _ptr(dG), _lib.stream_ptr()), "ia_dwconv_time")
        Gm = torch.empty(N, d, dtype=torch.float32, device=dev)
        _lib.check(L.ia_glu_mask(_ptr(S["c2"]), _ptr(S["lens"]), B, T, d, _ptr(Gm), _lib.stream_ptr()), "ia_glu_mask")
        dwd = torch.empty(d, ksz, dtype=torch.float32, device=dev)
        dbd = torch.empty(d, dtype=torch.float32, device=dev)
        _lib.check(L.ia_dwconv_time_wgrad(_ptr(Gm), _ptr(dz), B, T, d, ksz, _ptr(dwd), _ptr(dbd),
                                          _ptr(fast.scratch(dev, L.ia_dwconv_scratch_elems(B, T, d, ksz))), _lib.stream_ptr()),
                   "ia_dwconv_time_wgrad")
        G["conv.depthwise_conv.weight"], G["conv.depthwise_conv.bias"] = dwd.view(d, 1, ksz), dbd
        dc2 = torch.empty(N, 2 * d, dtype=torch.bfloat16, device=dev)
        _lib.check(L.ia_glu_bwd(_ptr(S["c2"]), _ptr(dG), _ptr(S["lens"]), B, T, d, _ptr(dc2), _lib.stream_ptr()), "ia_glu_bwd")
        dy, dWp1, G["conv.pointwise_conv1.bias"] = _lin_bwd(dc2, S["y3"], W["wp1"])
        G["conv.pointwise_conv1.weight"] = dWp1.unsqueeze(-1)
        dx2, G["norm_conv.weight"], G["norm_conv.bias"] = _ln_bwd(S["x2"], layer.norm_conv, dy_bf16=dy, dx_in=dx3)
        del dx3, dz, dG, Gm, dc2, dc3
        # self-attention
        dB = _branch_grad(dx2, 1.0, p, seed + 3)
        dctx, G["self_attn.linear_out.weight"], G["self_attn.linear_out.bias"] = _lin_bwd(dB, S["ctxv"], W["wo"])
        dqkv, dpl, du, dv = fast.relpos_attention_flash_bwd(S["qkv"], S["pl"], att.pos_bias_u, att.pos_bias_v, S["lens"], S["ctxv"],
                                                            dctx, S["lse"], B, T, att.h, att.d_k, patt, seed + 7)
        G["self_attn.pos_bias_u"], G["self_attn.pos_bias_v"] = du, dv
        dy, dWqkv, dbqkv = _lin_bwd(dqkv, S["y2"], W["wqkv"])
        for i, nm in enumerate(("q", "k", "v")):
            G[f"self_attn.linear_{nm}.weight"] = dWqkv[i * d:(i + 1) * d]
            G[f"self_attn.linear_{nm}.bias"] = dbqkv[i * d:(i + 1) * d]
        # (bias-free projection of the position table: same split-K kernel, the library needs 36 us for this 0.1 GFLOP TN GEMM)
        G["self_attn.linear_pos.weight"] = _lin_bwd(dpl, S["pe"], None, need_dx=False)[1]
        dx1, G["norm_self_att.weight"], G["norm_self_att.bias"] = _ln_bwd(S["x1"], layer.norm_self_att, dy_bf16=dy, dx_in=dx2)
        del dx2, dqkv, dctx
        # feed_forward1
        dB = _branch_grad(dx1, layer.fc_factor, p, seed + 2)
        dh, G["feed_forward1.linear2.weight"], G["feed_forward1.linear2.bias"] = _lin_bwd(dB, S["h1"], W["w2"])
        dhp = _silu_dropout_bwd(S["h1p"], dh, pff1, seed + 1)
        dy, G["feed_forward1.linear1.weight"], G["feed_forward1.linear1.bias"] = _lin_bwd(dhp, S["y1"], W["w1"])
        dx0, G["norm_feed_forward1.weight"], G["norm_feed_forward1.bias"] = _ln_bwd(S["x0"], layer.norm_feed_forward1,
                                                                                  dy_bf16=dy, dx_in=dx1)
        # parameter gradients: added to existing .grad buffers in ONE multi-tensor launch (autograd's AccumulateGrad
        # would issue one small add per parameter: 40 launches per block); parameters without a .grad get theirs returned
        outs, dst, src = [], [], []
        for n, r, q in zip(names, req, params):
            if not r:
                outs.append(None)
            elif DIRECT_ACCUMULATE and q.grad is not None and q.grad.dtype == torch.float32:
                dst.append(q.grad); src.append(G[n].reshape(q.grad.shape)); outs.append(None)
            else:
                outs.append(G[n].reshape(q.shape))
        if dst:
            torch._foreach_add_(dst, src)
        return (dx0, None, None, None, None, None, None) + tuple(outs)


def pad_pos_emb(pos_emb, d):
    """[1, 2T-1, d] -> bf16 [ceil8(2T-1), d] with zero rows appended (row count a multiple of 8 keeps the bf16 GEMMs that
    touch it -- linear_pos forward and its weight gradient -- on aligned fast paths)."""
    pe = pos_emb.detach().reshape(-1, d)
    rows = (pe.shape[0] + 7) // 8 * 8
    out = torch.zeros(rows, d, dtype=torch.bfloat16, device=pe.device)
    out[:pe.shape[0]] = pe
    return out


def conformer_block(x2d, layer, lens, pe_bf16, B, T, seed):
    """x2d [B*T, d] f32 residual stream -> [B*T, d] f32 (autograd-connected to x2d and the block's parameters).
    pe_bf16: pad_pos_emb(pos_emb) (>= 2T-1 rows)."""
    if USE_NATIVE_BLOCKS and layer.training:
        att = layer.self_attn
        if _lib.lib().ia_conformer_block_supported(x2d.shape[-1], layer.feed_forward1.linear1.weight.shape[0], att.h,
                                                   layer.conv.depthwise_conv.weight.shape[-1], T):
            return conformer_block_native(x2d, layer, lens, pe_bf16, B, T, seed)
    return _ConformerBlockFn.apply(x2d, layer, lens, pe_bf16, B, T, seed, *list(layer.parameters()))


# ---------------------------------------------------------------------------------------------------------------------
# Native executor (csrc/block_train.hip): the same block as ONE C call forward and TWO C calls backward around the
# attention core's backward.  The node above issued ~40 + ~70 launches per block through ctypes (~17 us of host time
# each: 4.5 ms of a 12 ms step on the host); here Python only carves two arenas and calls three functions.
USE_NATIVE_BLOCKS = True

_SAVED_FIELDS = ("y1", "h1p", "h1", "x1", "y2", "qkv", "pl", "ctxv", "x2", "y3", "c2", "z", "sums", "c3", "x3", "y4", "h4p", "h4", "x4", "lse")


def _saved_layout(N, d, dff, pos_rows, H):
    """(field -> (byte offset, bytes)), total bytes: every buffer 256-byte aligned."""
    sizes = dict(y1=N * d * 2, h1p=N * dff * 2, h1=N * dff * 2, x1=N * d * 4, y2=N * d * 2, qkv=N * 3 * d * 2, pl=pos_rows * d * 2,
                 ctxv=N * d * 2, x2=N * d * 4, y3=N * d * 2, c2=N * 2 * d * 2, z=N * d * 4, sums=(2 * d + 64) * 4, c3=N * d * 2,
                 x3=N * d * 4, y4=N * d * 2, h4p=N * dff * 2, h4=N * dff * 2, x4=N * d * 4, lse=N * H * 4)
    lay, o = {}, 0
    for f in _SAVED_FIELDS:
        lay[f] = (o, sizes[f])
        o = (o + sizes[f] + 255) // 256 * 256
    return lay, o


class _BlockRuntime:
    """Per-layer persistent state of the native path: gradient arena (f32), the ia_block_grads struct pointing into it and
    the device table {dst .grad pointer, src arena pointer, n} of the one multi-tensor add that ends the backward."""

    def __init__(self, layer, dev):
        import ctypes
        d = layer.norm_out.weight.shape[0]
        dff = layer.feed_forward1.linear1.weight.shape[0]
        ksz = layer.conv.depthwise_conv.weight.shape[-1]
        self.d, self.dff, self.ksz = d, dff, ksz
        # arena layout: weight | bias pairs contiguous (one finishing pass of the split-K weight-gradient kernel)
        blocks = [("w_ff1a", dff * d), ("b_ff1a", dff), ("w_ff1b", d * dff), ("b_ff1b", d), ("w_qkv", 3 * d * d), ("b_qkv", 3 * d),
                  ("w_pos", d * d), ("w_out", d * d), ("b_out", d), ("w_pw1", 2 * d * d), ("b_pw1", 2 * d), ("w_pw2", d * d), ("b_pw2", d),
                  ("w_ff2a", dff * d), ("b_ff2a", dff), ("w_ff2b", d * dff), ("b_ff2b", d)]
        for nm in ("ln_ff1", "ln_att", "ln_conv", "ln_ff2", "ln_out"):
            blocks += [(nm + "_g", d), (nm + "_b", d)]
        blocks += [("dw_w", d * ksz), ("dw_b", d), ("bn_g", d), ("bn_b", d), ("pos_u", d), ("pos_v", d)]
        off, o = {}, 0
        for nm, n in blocks:
            off[nm] = (o, n)
            o += (n + 3) // 4 * 4       # keep every block 16-byte aligned
        self.arena = torch.zeros(o, dtype=torch.float32, device=dev)
        self.off = off
        base = self.arena.data_ptr()
        self.grads = _lib.BlockGrads()
        for nm, _ in _lib.BlockGrads._fields_:
            setattr(self.grads, nm, base + 4 * off[nm][0])
        self.pos_u = self.arena[off["pos_u"][0]:off["pos_u"][0] + d]
        self.pos_v = self.arena[off["pos_v"][0]:off["pos_v"][0] + d]
        att, cv, ff1, ff2 = layer.self_attn, layer.conv, layer.feed_forward1, layer.feed_forward2
        self.param_src = [   # (parameter, arena block, element offset inside the block)
            (ff1.linear1.weight, "w_ff1a", 0), (ff1.linear1.bias, "b_ff1a", 0), (ff1.linear2.weight, "w_ff1b", 0), (ff1.linear2.bias, "b_ff1b", 0),
            (att.linear_q.weight, "w_qkv", 0), (att.linear_k.weight, "w_qkv", d * d), (att.linear_v.weight, "w_qkv", 2 * d * d),
            (att.linear_q.bias, "b_qkv", 0), (att.linear_k.bias, "b_qkv", d), (att.linear_v.bias, "b_qkv", 2 * d),
            (att.linear_pos.weight, "w_pos", 0), (att.linear_out.weight, "w_out", 0), (att.linear_out.bias, "b_out", 0),
            (att.pos_bias_u, "pos_u", 0), (att.pos_bias_v, "pos_v", 0),
            (cv.pointwise_conv1.weight, "w_pw1", 0), (cv.pointwise_conv1.bias, "b_pw1", 0), (cv.pointwise_conv2.weight, "w_pw2", 0),
            (cv.pointwise_conv2.bias, "b_pw2", 0), (cv.depthwise_conv.weight, "dw_w", 0), (cv.depthwise_conv.bias, "dw_b", 0),
            (cv.batch_norm.weight, "bn_g", 0), (cv.batch_norm.bias, "bn_b", 0),
            (ff2.linear1.weight, "w_ff2a", 0), (ff2.linear1.bias, "b_ff2a", 0), (ff2.linear2.weight, "w_ff2b", 0), (ff2.linear2.bias, "b_ff2b", 0),
            (layer.norm_feed_forward1.weight, "ln_ff1_g", 0), (layer.norm_feed_forward1.bias, "ln_ff1_b", 0),
            (layer.norm_self_att.weight, "ln_att_g", 0), (layer.norm_self_att.bias, "ln_att_b", 0),
            (layer.norm_conv.weight, "ln_conv_g", 0), (layer.norm_conv.bias, "ln_conv_b", 0),
            (layer.norm_feed_forward2.weight, "ln_ff2_g", 0), (layer.norm_feed_forward2.bias, "ln_ff2_b", 0),
            (layer.norm_out.weight, "ln_out_g", 0), (layer.norm_out.bias, "ln_out_b", 0)]
        self._table_key, self._table = None, None

    def add_table(self):
        """Device table of the final multi-tensor add; rebuilt only when a .grad buffer moves."""
        key = tuple((q.grad.data_ptr() if (q.requires_grad and q.grad is not None) else 0) for q, _, _ in self.param_src)
        if key != self._table_key:
            rows = []
            base = self.arena.data_ptr()
            for (q, blk, sub), gp in zip(self.param_src, key):
                if gp:
                    rows.append((gp, base + 4 * (self.off[blk][0] + sub), q.numel()))
            self._table = torch.tensor(rows, dtype=torch.int64).to(self.arena.device) if rows else None
            self._table_key = key
            self._n_rows = len(rows)
        return self._table, (self._n_rows if self._table is not None else 0)


_RUNTIMES = {}
_BWD_WS = {}


def _runtime(layer, dev):
    rt = _RUNTIMES.get(id(layer))
    if rt is None or rt.arena.device != dev:
        import weakref
        rt = _RUNTIMES[id(layer)] = _BlockRuntime(layer, dev)
        weakref.finalize(layer, _RUNTIMES.pop, id(layer), None)
    return rt


class _ConformerBlockNativeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, lens, pe, B, T, seed, *params):
        import ctypes
        L = _lib.lib()
        N, d = x.shape
        dev = x.device
        rt = _runtime(layer, dev)
        tr = layer.training
        if not tr:
            raise RuntimeError("trainable fused block expects train-mode BatchNorm (batch statistics)")
        bp = fast._block_params(layer)
        x0 = x.contiguous()
        lay, nbytes = _saved_layout(N, d, rt.dff, pe.shape[0], layer.self_attn.h)
        arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        saved = _lib.BlockSaved()
        base = arena.data_ptr()
        for f in _SAVED_FIELDS:
            setattr(saved, f, base + lay[f][0])
        out = torch.empty(N, d, dtype=torch.float32, device=dev)
        att = layer.self_attn
        nvt = L.ia_attn_vt_elems(B, T, att.h)
        vt = fast._VT.get((dev.index, nvt))
        if vt is None:
            vt = fast._VT[(dev.index, nvt)] = torch.empty(nvt, dtype=torch.bfloat16, device=dev)
        dw_scr = fast.scratch(dev, L.ia_dwconv_scratch_elems(B, T, d, rt.ksz))
        bn = layer.conv.batch_norm
        group = fast.bn_sync_group(bn)
        fwd_args = (ctypes.addressof(bp), _ptr(x0), _ptr(pe), pe.shape[0], _ptr(lens), B, T, int(seed) & 0xFFFFFFFF,
                    ctypes.addressof(saved), _ptr(out), _ptr(vt), _ptr(dw_scr))
        if group is None:
            _lib.check(L.ia_conformer_block_fwd(*fwd_args, _lib.stream_ptr()), "ia_conformer_block_fwd")
        else:   # SyncBatchNorm over several ranks: split at the exchange of the BatchNorm sums
            _lib.check(L.ia_conformer_block_fwd_phase(*fwd_args, 1, _lib.stream_ptr()), "ia_conformer_block_fwd_phase")
            o = lay["sums"][0]
            fast.bn_sync_sums(arena[o:o + (2 * d + 1) * 4].view(torch.float32), N, d, bn, group)
            _lib.check(L.ia_conformer_block_fwd_phase(*fwd_args, 2, _lib.stream_ptr()), "ia_conformer_block_fwd_phase")
        ctx.group = group
        ctx.keep = (x0, arena, saved, lay, pe, lens, bp)
        ctx.meta = (layer, B, T, seed, [q.requires_grad for q in params], params)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        if ctx.keep is None:
            raise RuntimeError("fused Conformer block: trying to backward through the graph a second time (its saved "
                               "activations were released by the first backward)")
        L = _lib.lib()
        x0, arena, saved, lay, pe, lens, bp = ctx.keep
        ctx.keep = None
        layer, B, T, seed, req, params = ctx.meta
        N, d = x0.shape
        dev = x0.device
        rt = _runtime(layer, dev)
        att = layer.self_attn
        n_ws = L.ia_conformer_block_bwd_ws_bytes(B, T, d, rt.dff, rt.ksz)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = _BWD_WS.get(key)
        if ws is None or ws.numel() < n_ws:
            ws = _BWD_WS[key] = torch.empty(n_ws, dtype=torch.uint8, device=dev)
        dout = dout.float().contiguous()
        dx2_p, dctx_p = ctypes.c_void_p(), ctypes.c_void_p()
        sp = _lib.stream_ptr()
        a_args = (ctypes.addressof(bp), ctypes.addressof(saved), ctypes.addressof(rt.grads), _ptr(dout), _ptr(lens), B, T,
                  int(seed) & 0xFFFFFFFF, _ptr(ws), n_ws, ctypes.addressof(dx2_p), ctypes.addressof(dctx_p))
        if ctx.group is None:
            _lib.check(L.ia_conformer_block_bwd_a(*a_args, sp), "ia_conformer_block_bwd_a")
        else:   # SyncBatchNorm: the backward's own exchange (local S1 | S2 = this rank's d beta | d gamma stay in the gradient arena)
            import torch.distributed as dist
            _lib.check(L.ia_conformer_block_bwd_a_phase(*a_args, 1, None, sp), "ia_conformer_block_bwd_a_phase")
            ob, og = rt.off["bn_b"][0], rt.off["bn_g"][0]
            Sg = torch.cat([rt.arena[ob:ob + d], rt.arena[og:og + d], torch.full((1,), float(N), dtype=torch.float32, device=dev)])
            fast.SYNC_BN_COLLECTIVES += 1
            dist.all_reduce(Sg, group=ctx.group)
            Sg = (Sg[:2 * d] * (float(N) / Sg[2 * d])).contiguous()
            _lib.check(L.ia_conformer_block_bwd_a_phase(*a_args, 2, _ptr(Sg), sp), "ia_conformer_block_bwd_a_phase")
        # attention core: key-tiled backward (ops/fast.relpos_attention_flash_bwd) on views of the saved arena / workspace
        def view(buf, off, shape, dtype):
            n = 1
            for s_ in shape:
                n *= s_
            return buf[off:off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)
        qkv = view(arena, lay["qkv"][0], (N, 3 * d), torch.bfloat16)
        pl = view(arena, lay["pl"][0], (pe.shape[0], d), torch.bfloat16)
        ctxv = view(arena, lay["ctxv"][0], (N, d), torch.bfloat16)
        dctx = view(ws, dctx_p.value - ws.data_ptr(), (N, d), torch.bfloat16)
        lse = view(arena, lay["lse"][0], (B * att.h, T), torch.float32)
        patt = float(att.dropout_rate)
        dqkv, dpl, du, dv = fast.relpos_attention_flash_bwd(qkv, pl, att.pos_bias_u, att.pos_bias_v, lens, ctxv, dctx, lse, B, T, att.h,
                                                            att.d_k, patt, seed + 7, dub_out=(rt.pos_u, rt.pos_v))
        dx0 = torch.empty(N, d, dtype=torch.float32, device=dev)
        table, n_rows = rt.add_table() if DIRECT_ACCUMULATE else (None, 0)
        st = L.ia_conformer_block_bwd_b(ctypes.addressof(bp), ctypes.addressof(saved), ctypes.addressof(rt.grads), _ptr(x0), _ptr(pe), pe.shape[0],
                                        _ptr(dqkv), _ptr(dpl), B, T, int(seed) & 0xFFFFFFFF, _ptr(ws), n_ws, _ptr(dx0), _ptr(table),
                                        n_rows, sp)
        _lib.check(st, "ia_conformer_block_bwd_b")
        outs = []
        by_id = {id(q): (blk, sub) for q, blk, sub in rt.param_src}
        for r, q in zip(req, params):
            if not r:
                outs.append(None)
            elif DIRECT_ACCUMULATE and q.grad is not None and q.grad.dtype == torch.float32:
                outs.append(None)            # added to .grad by the multi-tensor launch of bwd_b
            else:
                blk, sub = by_id[id(q)]
                o = rt.off[blk][0] + sub
                outs.append(rt.arena[o:o + q.numel()].view(q.shape).clone())
        return (dx0, None, None, None, None, None, None) + tuple(outs)


def conformer_block_native(x2d, layer, lens, pe_bf16, B, T, seed):
    return _ConformerBlockNativeFn.apply(x2d, layer, lens, pe_bf16, B, T, seed, *list(layer.parameters()))
