"""The step's tail on lean autograd nodes: CTC head + loss on raw logits, the loss combination with the monitor's values,
the prediction network's input embedding, and one-launch accumulation of parameter gradients.

Each node here replaces a chain of small ATen launches around HIP kernels that were already there (index_select + zeros +
copy + cast, log_softmax forward / backward, mul + mul + add + stack, the sort-based embedding backward, one AccumulateGrad
add per parameter): at 4-5 us of stream time and ~10 us of host time per launch they cost more than the arithmetic.
Reference semantics: ConvASRDecoder.forward (A/modules/conv_asr.py:459-490) + CTCLoss.forward (A/losses/ctc.py:68-82);
the loss combination of hybrid_rnnt_ctc_models.py:899-913; RNNTDecoder.predict (A/modules/rnnt.py:734-792).
"""
import ctypes
import weakref

import torch

from .. import _lib
from . import fast

# True: parameter gradients are ADDED into existing fp32 `.grad` buffers by the nodes themselves (cl.FlatParams keeps every
# .grad as a view of one flat buffer) and autograd is told None -- one launch instead of one AccumulateGrad add per tensor.
# torch.autograd.grad(...) callers set this False (ops/block.DIRECT_ACCUMULATE is the blocks' switch of the same kind).
DIRECT_ACCUMULATE = True


def _direct(p):
    return DIRECT_ACCUMULATE and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()


class _AxpyRow(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("n", ctypes.c_longlong), ("scale", ctypes.c_float),
                ("pad", ctypes.c_int)]


def multi_axpy(pairs):
    """[(dst f32 tensor, src f32 tensor, scale), ...]: dst += scale * src for all of them in ONE launch (csrc/tail_ops.hip)."""
    if not pairs:
        return
    n = len(pairs)
    arr = (_AxpyRow * n)()
    keep = []
    big = 0
    for i, (dst, src, scale) in enumerate(pairs):
        src = src.contiguous()
        keep.append(src)
        assert dst.is_contiguous() and dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.numel() == src.numel()
        arr[i].dst, arr[i].src, arr[i].n, arr[i].scale = dst.data_ptr(), src.data_ptr(), dst.numel(), float(scale)
        big = max(big, dst.numel())
    blocks = max(1, min(256, (big // 4 + 255) // 256))
    _lib.check(_lib.lib().ia_multi_axpy(ctypes.addressof(arr), n, blocks, _lib.stream_ptr()), "ia_multi_axpy")


def accumulate_or_return(params_and_grads):
    """[(parameter, gradient tensor | None, scale)] -> tuple of what the autograd node should return for them: None where the
    gradient was added into `.grad` directly (one launch for all of those), the (scaled) tensor otherwise."""
    outs, pairs = [], []
    for p, g, scale in params_and_grads:
        if g is None or not p.requires_grad:
            outs.append(None)
        elif _direct(p):
            pairs.append((p.grad, g.reshape(p.grad.shape), scale))
            outs.append(None)
        else:
            outs.append((g if scale == 1.0 else g * scale).reshape(p.shape).to(p.dtype))
    multi_axpy(pairs)
    return tuple(outs)


# ----------------------------------------------------------------------------------------------------------------------
_SHARED = {}   # device index -> (data_ptr, shape, bf16 image): ONE activation published by model.training_step for the step


def share_bf16(x):
    """Cast the contiguous fp32 activation `x` to bf16 ONCE (on the current stream) and publish the image for the consumers
    of this step (the encoder output feeds both the joint's encoder projection and the CTC head).  The caller keeps `x` alive
    and calls unshare() when the step's forward is over -- while the entry exists no other tensor can own that address, so a
    (pointer, shape) match is an identity match.  A consumer on another stream must be ordered behind this call."""
    x2d = x.detach().reshape(-1, x.shape[-1])
    xb = x2d if x2d.dtype == torch.bfloat16 else x2d.to(torch.bfloat16)
    # (the weak reference makes the entry die with `x` even when the step is left by an exception: a dead source's address
    #  may be handed to another tensor, a live one's cannot)
    _SHARED[x2d.device.index] = (x2d.data_ptr(), tuple(x2d.shape), xb, weakref.ref(x))
    return xb


def unshare(device):
    _SHARED.pop(device.index if device.index is not None else torch.cuda.current_device(), None)


def bf16_of(x):
    """bf16 image [rows, last dim] of a CONTIGUOUS activation: the published image when `x` is the step's shared activation
    (share_bf16), a fresh cast otherwise."""
    x2d = x.detach().reshape(-1, x.shape[-1])
    if x2d.dtype == torch.bfloat16:
        return x2d
    hit = _SHARED.get(x2d.device.index)
    if hit is not None and hit[3]() is not None and hit[0] == x2d.data_ptr() and hit[1] == tuple(x2d.shape):
        return hit[2]
    return x2d.to(torch.bfloat16)


class _CtcHeadLoss(torch.autograd.Function):
    """nll[b] of the language-restricted CTC head on x [B,T,d]: select the language's rows of the 5633-wide Conv1d(k=1) head
    -> GEMM -> raw logits [B*T, 264] -> per-frame lse -> alpha/beta on (logit - lse).  Backward: (softmax - occupancy) written
    as the bf16 GEMM operand -> dX on the HIP GEMM against the selected rows' transpose, dW | db on csrc/gemm_tn.hip, scattered
    into the head's gradient rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, targets, in_lens, tg_lens, row0, nrows, extra_row, blank, zero_infinity, keep):
        L = _lib.lib()
        B, T, d = x.shape
        M = B * T
        dev = x.device
        n_all = weight.shape[0]
        V = nrows + (1 if extra_row >= 0 else 0)
        Vp = (V + 7) // 8 * 8
        xb = bf16_of(x if x.is_contiguous() else x.contiguous())
        w2 = weight.detach().view(n_all, d)
        wsel = torch.empty(Vp, d, dtype=torch.bfloat16, device=dev)
        wselT = torch.empty(d, Vp, dtype=torch.bfloat16, device=dev)
        bsel = torch.empty(Vp, dtype=torch.float32, device=dev)
        _lib.check(L.ia_select_rows_cast(_lib.ptr(w2), d, _lib.ptr(bias.detach()), int(row0), int(nrows), int(extra_row), d, Vp, 1.0, 0,
                                         _lib.ptr(wsel), _lib.ptr(wselT), Vp, _lib.ptr(bsel), _lib.stream_ptr()), "ia_select_rows_cast")
        logits = torch.empty(M, Vp, dtype=torch.float32, device=dev)
        fast.gemm(xb, wsel, bsel, out_f32=logits, want_bf16=False)
        lse = torch.empty(M, dtype=torch.float32, device=dev)
        _lib.check(L.ia_ctc_row_lse(_lib.ptr(logits), Vp, M, V, _lib.ptr(lse), _lib.stream_ptr()), "ia_ctc_row_lse")
        tg = targets.contiguous()
        S = tg.shape[1]
        n = L.ia_ctc_workspace_bytes(B, T, S)
        if n == 0:
            raise RuntimeError("CTC (HIP): target length beyond the kernel's limit (S <= 255)")
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        _lib.check(L.ia_ctc_forward_logits(_lib.ptr(logits), Vp, _lib.ptr(lse), _lib.ptr(tg), _lib.ptr(in_lens), _lib.ptr(tg_lens), B, T, V, S,
                                           int(blank), int(bool(zero_infinity)), _lib.ptr(nll), _lib.ptr(ws), n, _lib.stream_ptr()),
                   "ia_ctc_forward_logits")
        ctx.saved = (xb, wselT, logits, lse, tg, in_lens, tg_lens, ws, n)
        ctx.meta = (B, T, d, V, Vp, S, int(blank), int(row0), int(nrows), int(extra_row), x.dtype, weight, bias)
        if keep is not None:   # greedy CTC decoding of the step's WER reads the logits (argmax is the log-probs' argmax)
            keep["logits"], keep["lse"], keep["V"] = logits.view(B, T, Vp), lse.view(B, T), V
        return nll

    @staticmethod
    def backward(ctx, gnll):
        if ctx.saved is None:
            raise RuntimeError("CTC head + loss (HIP): trying to backward through the graph a second time -- its lattice workspace "
                               "was released by the first backward (retain_graph=True is not supported; run the forward again)")
        L = _lib.lib()
        xb, wselT, logits, lse, tg, il, tl, ws, n = ctx.saved
        ctx.saved = None
        B, T, d, V, Vp, S, blank, row0, nrows, extra_row, xdt, weight, bias = ctx.meta
        dev = xb.device
        M = B * T
        from . import joint as _joint
        if _joint.LAST_GRAD_KERNEL_EVENT is not None:   # on a side stream under the joint's backward: start after its
            torch.cuda.current_stream(dev).wait_event(_joint.LAST_GRAD_KERNEL_EVENT)   # HBM-bound gradient kernel
        g = gnll.reshape(-1).float().contiguous()
        dyb = torch.empty(M, Vp, dtype=torch.bfloat16, device=dev)
        _lib.check(L.ia_ctc_backward_logits(_lib.ptr(logits), Vp, _lib.ptr(lse), _lib.ptr(tg), _lib.ptr(il), _lib.ptr(tl), B, T, V, S, blank,
                                            _lib.ptr(g), 1.0, _lib.ptr(dyb), Vp, _lib.ptr(ws), n, _lib.stream_ptr()), "ia_ctc_backward_logits")
        dx = None
        if ctx.needs_input_grad[0]:
            if xdt == torch.float32:
                dx = torch.empty(M, d, dtype=torch.float32, device=dev)
                fast.gemm(dyb, wselT, out_f32=dx, want_bf16=False)
            else:
                dx = fast.gemm(dyb, wselT)[1].to(xdt)
            dx = dx.view(B, T, d)
        dW = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dWs, dbs = fast.gemm_tn(dyb, xb)          # [Vp, d], [Vp] f32
            n_all = weight.shape[0]
            wg, bg = weight.grad, bias.grad
            if (_direct(weight) and _direct(bias) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]):
                _lib.check(L.ia_rows_scatter_add(_lib.ptr(wg), d, _lib.ptr(dWs), d, row0, nrows, extra_row, d, 1.0, _lib.ptr(bg),
                                                 _lib.ptr(dbs), _lib.stream_ptr()), "ia_rows_scatter_add")
            else:
                dWf = torch.zeros(n_all, d, dtype=torch.float32, device=dev)
                dbf = torch.zeros(n_all, dtype=torch.float32, device=dev)
                _lib.check(L.ia_rows_scatter_add(_lib.ptr(dWf), d, _lib.ptr(dWs), d, row0, nrows, extra_row, d, 1.0, _lib.ptr(dbf),
                                                 _lib.ptr(dbs), _lib.stream_ptr()), "ia_rows_scatter_add")
                dW, db = dWf.view(weight.shape).to(weight.dtype), dbf.to(bias.dtype)
        return dx, dW, db, None, None, None, None, None, None, None, None, None


def ctc_head_loss(x_btd, weight, bias, targets, in_lens, tg_lens, row0, nrows, extra_row, blank, zero_infinity=True, keep=None):
    """x [B,T,d] -> nll [B] f32 of the CTC head restricted to rows [row0, row0 + nrows) + extra_row of `weight` [n, d(,1)]."""
    return _CtcHeadLoss.apply(x_btd, weight, bias, targets.long(), in_lens.long().contiguous(), tg_lens.long().contiguous(),
                              row0, nrows, extra_row, blank, zero_infinity, keep)


def ctc_head_loss_supported(x, targets, d):
    return x.is_cuda and targets.dim() == 2 and targets.shape[1] <= 255 and d % 8 == 0


# ----------------------------------------------------------------------------------------------------------------------
class _LossCombine(torch.autograd.Function):
    """total = (1 - w) mean(costs) + w mean(nll) and the monitor's [rnnt, ctc, total, lstm timeout flag] in one launch
    (hybrid_rnnt_ctc_models.py:899-913 issues three .item() reads and two scalar kernels per term)."""

    @staticmethod
    def forward(ctx, costs, nll, w, flag_words):
        L = _lib.lib()
        B = costs.shape[0]
        dev = costs.device
        vals = torch.empty(4, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        fl = [_lib.ptr(t) for t in flag_words[:4]] + [None] * (4 - min(4, len(flag_words)))
        c = costs.detach().float().contiguous()
        k = nll.detach().float().contiguous() if nll is not None else None
        _lib.check(L.ia_loss_combine(_lib.ptr(c), _lib.ptr(k), B, float(w), fl[0], fl[1], fl[2], fl[3], _lib.ptr(vals), _lib.ptr(total),
                                     _lib.stream_ptr()), "ia_loss_combine")
        ctx.meta = (B, float(w), costs.dtype, nll.dtype if nll is not None else None)
        ctx.mark_non_differentiable(vals)
        return total, vals

    @staticmethod
    def backward(ctx, gtotal, _gvals):
        B, w, cdt, ndt = ctx.meta
        dev = gtotal.device
        gc = torch.empty(B, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gn = torch.empty(B, dtype=torch.float32, device=dev) if (ndt is not None and ctx.needs_input_grad[1]) else None
        g = gtotal.detach().float().contiguous()
        if gc is not None or gn is not None:
            _lib.check(_lib.lib().ia_loss_combine_bwd(_lib.ptr(g), B, w, _lib.ptr(gc), _lib.ptr(gn), _lib.stream_ptr()), "ia_loss_combine_bwd")
        return (gc.to(cdt) if gc is not None else None), (gn.to(ndt) if gn is not None else None), None, None


def loss_combine(costs, nll, ctc_weight, flag_words=()):
    """(total 0-dim f32 with autograd, vals [4] f32 = [mean costs, mean nll, total, #timeout flags])."""
    return _LossCombine.apply(costs, nll, ctc_weight, tuple(flag_words))


# ----------------------------------------------------------------------------------------------------------------------
class _EmbedSOS(torch.autograd.Function):
    """Prediction-network input for the persistent LSTM: [U+1, B, H] bf16, row 0 the zero SOS step, row u the embedding of
    token u-1 (rnnt.py:734-751 does embed -> cat(zeros) -> transpose).  Backward: deterministic row sums added into the
    embedding's gradient (nn.Embedding's dense backward sorts the indices: ~12 launches)."""

    @staticmethod
    def forward(ctx, emb_weight, tokens, scan_rows, pad_row):
        B, U = tokens.shape
        n_rows, H = emb_weight.shape
        out = torch.empty(U + 1, B, H, dtype=torch.bfloat16, device=emb_weight.device)
        tk = tokens.contiguous()
        _lib.check(_lib.lib().ia_embed_sos(_lib.ptr(emb_weight.detach()), _lib.ptr(tk), B, U, H, n_rows, 1, 1, _lib.ptr(out),
                                           _lib.stream_ptr()), "ia_embed_sos")
        ctx.tok = tk
        ctx.meta = (B, U, H, n_rows, int(scan_rows), int(pad_row), emb_weight)
        return out

    @staticmethod
    def backward(ctx, dX):
        B, U, H, n_rows, scan_rows, pad_row, emb = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        is16 = dX.dtype == torch.bfloat16
        dXc = dX.contiguous() if is16 else dX.float().contiguous()
        direct = _direct(emb)
        dE = emb.grad if direct else torch.zeros(n_rows, H, dtype=torch.float32, device=dX.device)
        _lib.check(_lib.lib().ia_embed_sos_bwd(_lib.ptr(dXc), int(is16), _lib.ptr(ctx.tok), B, U, H, min(n_rows, scan_rows), pad_row, 1,
                                               1.0, _lib.ptr(dE), _lib.stream_ptr()), "ia_embed_sos_bwd")
        return (None if direct else dE.to(emb.dtype)), None, None, None


def embed_sos(emb_weight, tokens, scan_rows=None, pad_row=-1):
    """tokens [B,U] i64 -> [U+1,B,H] bf16.  `scan_rows`: only embedding rows below it can be referenced (the multilingual
    model feeds language-local ids 0..255: the backward then launches 256 workgroups instead of 5633)."""
    n_rows = emb_weight.shape[0]
    return _EmbedSOS.apply(emb_weight, tokens.long(), n_rows if scan_rows is None else int(scan_rows), pad_row)
