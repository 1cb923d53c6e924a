"""CTC loss with the reference's wrapper semantics (A/losses/ctc.py:45-82): blank = num_classes,
zero_infinity, 'mean_batch' = mean over the per-utterance losses."""
import torch
import torch.nn as nn

from .. import _lib


class _CTCHip(torch.autograd.Function):
    """nn.CTCLoss(reduction='none') on csrc/ctc.hip; the gradient kernel runs in backward() with the upstream
    per-utterance gradient folded in (one write of the [B,T,V] gradient)."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank, zero_infinity):
        L = _lib.lib()
        B, T, V = log_probs.shape
        S = targets.shape[1]
        lp = log_probs.detach().float().contiguous()
        tg = targets.contiguous()
        n = L.ia_ctc_workspace_bytes(B, T, S)
        ws = torch.empty(n, dtype=torch.uint8, device=lp.device)
        nll = torch.empty(B, dtype=torch.float32, device=lp.device)
        st = L.ia_ctc_forward(_lib.ptr(lp), _lib.ptr(tg), _lib.ptr(input_lengths), _lib.ptr(target_lengths), B, T, V, S,
                              int(blank), int(bool(zero_infinity)), _lib.ptr(nll), _lib.ptr(ws), n, _lib.stream_ptr())
        _lib.check(st, "ia_ctc_forward")
        ctx.saved = (lp, tg, input_lengths, target_lengths, ws, n, int(blank), log_probs.dtype)
        return nll

    @staticmethod
    def backward(ctx, gnll):
        if ctx.saved is None:
            raise RuntimeError("CTC loss (HIP): trying to backward through the graph a second time -- its lattice workspace "
                               "was released by the first backward (retain_graph=True is not supported; run the forward again)")
        lp, tg, il, tl, ws, n, blank, dt = ctx.saved
        ctx.saved = None
        B, T, V = lp.shape
        from ..ops import joint as _joint
        if _joint.LAST_GRAD_KERNEL_EVENT is not None:   # on a side stream under the joint's backward: start after its
            torch.cuda.current_stream(lp.device).wait_event(_joint.LAST_GRAD_KERNEL_EVENT)   # HBM-bound gradient kernel
        grad = torch.empty_like(lp)
        g = gnll.reshape(-1).float().contiguous()
        st = _lib.lib().ia_ctc_backward(_lib.ptr(lp), _lib.ptr(tg), _lib.ptr(il), _lib.ptr(tl), B, T, V, tg.shape[1], blank,
                                        _lib.ptr(g), _lib.ptr(grad), _lib.ptr(ws), n, _lib.stream_ptr())
        _lib.check(st, "ia_ctc_backward")
        return grad.to(dt), None, None, None, None, None


def ctc_hip_supported(log_probs, targets):
    return log_probs.is_cuda and targets.dim() == 2 and targets.shape[1] <= 255



class CTCLoss(nn.Module):
    def __init__(self, num_classes, zero_infinity=False, reduction='mean_batch'):
        super().__init__()
        if reduction not in ['none', 'mean', 'sum', 'mean_batch', 'mean_volume']:
            raise ValueError('`reduction` must be one of [mean, sum, mean_batch, mean_volume]')
        self._blank = num_classes
        self.zero_infinity = zero_infinity
        self.config_reduction = reduction
        self._apply_reduction = reduction in ('mean_batch', 'mean_volume')
        self._ctc_reduction = 'none' if self._apply_reduction else reduction

    def reduce(self, losses, target_lengths):
        if self.config_reduction == 'mean_batch':
            losses = losses.mean()
        elif self.config_reduction == 'mean_volume':
            losses = losses.sum() / target_lengths.sum()
        return losses

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        input_lengths = input_lengths.long()
        target_lengths = target_lengths.long()
        targets = targets.long()
        if ctc_hip_supported(log_probs, targets):
            loss = _CTCHip.apply(log_probs, targets, input_lengths.contiguous(), target_lengths.contiguous(), self._blank,
                                 self.zero_infinity)
            if self._ctc_reduction == 'sum':
                loss = loss.sum()
            elif self._ctc_reduction == 'mean':
                loss = (loss / target_lengths.clamp(min=1)).mean()
        else:
            loss = torch.nn.functional.ctc_loss(log_probs.transpose(1, 0), targets, input_lengths, target_lengths,
                                                blank=self._blank, reduction=self._ctc_reduction,
                                                zero_infinity=self.zero_infinity)
        if self._apply_reduction:
            loss = self.reduce(loss, target_lengths)
        return loss
