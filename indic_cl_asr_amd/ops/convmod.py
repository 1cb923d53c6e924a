"""Conformer convolution-module core: GLU -> pad mask -> depthwise conv k -> BatchNorm (batch stats incl. the
zeroed padded frames, as the reference) -> SiLU  (conformer_modules.py:340-366), autograd path of the trainable blocks.

Time-major throughout ([B,T,d], channel = lane): the depthwise conv and its two gradients are HIP kernels
(csrc/dwconv.hip: ia_dwconv_time / ia_dwconv_time_wgrad) -- MIOpen has no tuned depthwise-1D kernels on gfx950
and falls back to naive ones (1.6 ms per weight gradient at bs32 x 15 s); no [B,d,T] transposes are made.
"""
import torch
import torch.nn.functional as F

from .. import _lib


class _DepthwiseConvTime(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        B, T, d = x.shape
        k = w.shape[-1]
        x = x.float().contiguous()
        w2 = w.detach().float().reshape(d, k).contiguous()
        y = torch.empty_like(x)
        st = _lib.lib().ia_dwconv_time(_lib.ptr(x), B, T, d, k, _lib.ptr(w2), _lib.ptr(b.detach().float().contiguous()), 0,
                                       _lib.ptr(y), _lib.stream_ptr())
        _lib.check(st, "ia_dwconv_time")
        ctx.save_for_backward(x, w2)
        ctx.wshape, ctx.wdtype, ctx.bdtype = w.shape, w.dtype, b.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, T, d = x.shape
        k = w2.shape[1]
        L = _lib.lib()
        dy = dy.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.ia_dwconv_time(_lib.ptr(dy), B, T, d, k, _lib.ptr(w2), None, 1, _lib.ptr(dx), _lib.stream_ptr()),
                       "ia_dwconv_time")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            from . import fast
            dw = torch.empty(d, k, dtype=torch.float32, device=x.device)
            db = torch.empty(d, dtype=torch.float32, device=x.device)
            _lib.check(L.ia_dwconv_time_wgrad(_lib.ptr(x), _lib.ptr(dy), B, T, d, k, _lib.ptr(dw), _lib.ptr(db),
                                              _lib.ptr(fast.scratch(x.device, L.ia_dwconv_scratch_elems(B, T, d, k))),
                                              _lib.stream_ptr()), "ia_dwconv_time_wgrad")
            dw = dw.view(ctx.wshape).to(ctx.wdtype)
            db = db.to(ctx.bdtype)
        return dx, dw, db


def dwconv_supported(x, k):
    return x.is_cuda and x.shape[-1] <= 1024 and k <= 31 and (k & 1) == 1


def glu_dwconv_bn_silu(x2, pad_mask, dw_weight, dw_bias, bn, training):
    """x2: [B,T,2d] output of pointwise_conv1 (time-major); pad_mask [B,T] True at padding; bn: the module's
    BatchNorm1d / SyncBatchNorm (called as a module so SyncBatchNorm.convert_sync_batchnorm keeps working,
    R/cl_baseline.py:133).  Returns [B,T,d] f32."""
    x = F.glu(x2, dim=-1)
    x = x.float().masked_fill(pad_mask.unsqueeze(-1), 0.0)
    B, T, d = x.shape
    k = dw_weight.shape[-1]
    if dwconv_supported(x, k):
        x = _DepthwiseConvTime.apply(x, dw_weight, dw_bias)
    else:
        with torch.autocast(device_type=x.device.type, enabled=False):
            x = F.conv1d(F.pad(x.transpose(1, 2), ((k - 1) // 2, (k - 1) // 2)), dw_weight.float(), dw_bias.float(),
                         groups=d).transpose(1, 2)
    x = bn(x.reshape(B * T, d)).view(B, T, d)
    return F.silu(x)
