"""Transducer loss on the HIP kernels (csrc/rnnt_loss.hip) behind the reference's loss surface.

Mirrors, name for name:
  RNNTLossHIP ....... RNNTLossNumba           K/rnnt_pytorch.py:386-437 (blank, reduction, fastemit_lambda, clamp)
  _RNNTHip .......... _RNNTNumba              K/rnnt_pytorch.py:40-91   (grads computed in forward, scaled in backward)
  certify_inputs .... certify_inputs          K/rnnt_pytorch.py:599-632 (TypeError / ValueError behaviour)
  RNNTLoss .......... RNNTLoss                A/losses/rnnt.py:333-508  (fp32 up-cast, narrowing, `reduce`)
(K/ = NeMo/nemo/collections/asr/parts/numba/rnnt_loss/, A/ = NeMo/nemo/collections/asr/)

GPU semantics of the reference are kept: `acts` are raw joint logits, the log-softmax is fused into the
kernels, gradients are w.r.t. the logits (gpu_rnnt_kernel.py:351-403).  There is no CPU path here.
"""
from typing import List, Optional

import torch

from .. import _lib


def _check_type(var, t, name):
    if var.dtype is not t:
        raise TypeError("{} must be {}".format(name, t))


def _check_contiguous(var, name):
    if not var.is_contiguous():
        raise ValueError("{} must be contiguous".format(name))


def _check_dim(var, dim, name):
    if len(var.shape) != dim:
        raise ValueError("{} must be {}D".format(name, dim))


def certify_inputs(log_probs, labels, lengths, label_lengths, max_T=None, max_U=None):
    """Same checks and messages as the reference.  `max_T`/`max_U` may be passed as host ints by callers that
    already know them (the model does) so that no device->host sync happens on the hot path."""
    _check_type(labels, torch.int64, "labels")
    _check_type(label_lengths, torch.int64, "label_lengths")
    _check_type(lengths, torch.int64, "lengths")
    _check_contiguous(log_probs, "log_probs")
    _check_contiguous(labels, "labels")
    _check_contiguous(label_lengths, "label_lengths")
    _check_contiguous(lengths, "lengths")
    if lengths.shape[0] != log_probs.shape[0]:
        raise ValueError(
            f"Must have a length per example. Given lengths dim: {lengths.shape[0]}, Log probs dim : {log_probs.shape[0]}")
    if label_lengths.shape[0] != log_probs.shape[0]:
        raise ValueError("Must have a label length per example. "
                         f"Given label lengths dim : {label_lengths.shape[0]}, Log probs dim : {log_probs.shape[0]}")
    _check_dim(log_probs, 4, "log_probs")
    _check_dim(labels, 2, "labels")
    _check_dim(lengths, 1, "lenghts")
    _check_dim(label_lengths, 1, "label_lenghts")
    if max_T is None:
        max_T = int(torch.max(lengths))
    if max_U is None:
        max_U = int(torch.max(label_lengths))
    T, U = log_probs.shape[1:3]
    if T != max_T:
        raise ValueError(f"Input length mismatch! Given T: {T}, Expected max T from input lengths: {max_T}")
    if U != max_U + 1:
        raise ValueError(f"Output length mismatch! Given U: {U}, Expected max U from target lengths: {max_U} + 1")


# Optional profiling hook used by bench.py: callable(B, T, U1, V) -> (start_event_ptr, stop_event_ptr) or None; the
# events are recorded around the gradient kernel of that launch.
PROFILE_HOOK = None


def _workspace(acts, workspace=None):
    B, T, U1, V = acts.shape
    nbytes = _lib.lib().ia_rnnt_workspace_bytes(B, T, U1)
    if nbytes == 0:
        raise RuntimeError(f"ia_rnnt_workspace_bytes({B},{T},{U1}) unsupported (U1 <= 1024)")
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=acts.device)
    return workspace, nbytes


def _check_acts(acts):
    if not acts.is_cuda:
        raise RuntimeError("rnnt loss: acts must live on the MI355X (no CPU path in the product)")
    if acts.dtype != torch.float32:
        raise TypeError("rnnt loss: acts must be float32 (A/losses/rnnt.py:449-468 forces fp32)")


def rnnt_forward_hip(acts, labels, act_lens, label_lens, blank, fastemit_lambda=0.0, need_backward=True, workspace=None):
    """ia_rnnt_forward: costs [B] + the state the backward needs (kept in `workspace`)."""
    _check_acts(acts)
    B, T, U1, V = acts.shape
    workspace, nbytes = _workspace(acts, workspace)
    costs = torch.empty(B, dtype=torch.float32, device=acts.device)
    st = _lib.lib().ia_rnnt_forward(_lib.ptr(acts), _lib.ptr(labels), _lib.ptr(act_lens), _lib.ptr(label_lens), B, T,
                                    U1, V, int(blank), float(fastemit_lambda), int(bool(need_backward)), _lib.ptr(costs),
                                    _lib.ptr(workspace), nbytes, _lib.stream_ptr())
    _lib.check(st, "ia_rnnt_forward")
    return costs, workspace


def rnnt_backward_hip(acts, labels, act_lens, label_lens, blank, workspace, cost_grad=None, fastemit_lambda=0.0,
                      clamp=0.0, inplace=False):
    """ia_rnnt_backward: d/d(logits) with the upstream per-utterance gradient folded into the single write."""
    B, T, U1, V = acts.shape
    _, nbytes = _workspace(acts, workspace)
    grads = acts if inplace else torch.empty_like(acts)
    post_scale = None
    if clamp > 0.0 and cost_grad is not None:
        post_scale, cost_grad = cost_grad, None
    if cost_grad is not None:
        cost_grad = cost_grad.to(torch.float32).contiguous()
    hook = PROFILE_HOOK(B, T, U1, V) if PROFILE_HOOK is not None else None
    ev0, ev1 = hook if hook is not None else (None, None)
    st = _lib.lib().ia_rnnt_backward(_lib.ptr(acts), _lib.ptr(labels), _lib.ptr(act_lens), _lib.ptr(label_lens), B, T,
                                     U1, V, int(blank), float(fastemit_lambda), float(clamp), _lib.ptr(cost_grad),
                                     _lib.ptr(grads), _lib.ptr(workspace), nbytes, _lib.stream_ptr(), ev0, ev1)
    _lib.check(st, "ia_rnnt_backward")
    if post_scale is not None:
        grads.mul_(post_scale.view(-1, 1, 1, 1).to(grads))
    return grads


def rnnt_loss_hip(acts, labels, act_lens, label_lens, blank, fastemit_lambda=0.0, clamp=0.0, want_grads=True,
                  inplace=False, workspace=None):
    """Raw call through the C ABI (ia_rnnt_loss).  Returns (costs[B], grads or None, workspace)."""
    _check_acts(acts)
    B, T, U1, V = acts.shape
    workspace, nbytes = _workspace(acts, workspace)
    costs = torch.empty(B, dtype=torch.float32, device=acts.device)
    grads = None
    if want_grads:
        grads = acts if inplace else torch.empty_like(acts)
    st = _lib.lib().ia_rnnt_loss(_lib.ptr(acts), _lib.ptr(labels), _lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, V,
                                 int(blank), float(fastemit_lambda), float(clamp), _lib.ptr(costs), _lib.ptr(grads),
                                 _lib.ptr(workspace), nbytes, _lib.stream_ptr())
    _lib.check(st, "ia_rnnt_loss")
    return costs, grads, workspace


def rnnt_alphas_betas(workspace, act_lens, label_lens, B, T, U1):
    """Test helper: dense [B,T,U1] forward/backward variables of the last ia_rnnt_loss call on `workspace`."""
    L = _lib.lib()
    alphas = torch.empty(B, T, U1, dtype=torch.float32, device=workspace.device)
    betas = torch.empty_like(alphas)
    st = L.ia_rnnt_export_alphas_betas(_lib.ptr(workspace), workspace.numel(), _lib.ptr(act_lens),
                                       _lib.ptr(label_lens), B, T, U1, _lib.ptr(alphas), _lib.ptr(betas),
                                       _lib.stream_ptr())
    _lib.check(st, "ia_rnnt_export_alphas_betas")
    return alphas, betas


class _RNNTHip(torch.autograd.Function):
    """_RNNTNumba (K/rnnt_pytorch.py:40-91) with the gradient kernel deferred to backward(): the upstream gradient
    is known there, so the [B,T,U,V] gradient is written once, already scaled."""

    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, clamp, max_T, max_U,
                own_acts):
        certify_inputs(acts, labels, act_lens, label_lens, max_T, max_U)
        if clamp < 0:
            raise ValueError("`clamp` must be 0.0 or positive float value.")
        need = acts.requires_grad
        a = acts.detach()
        costs, ws = rnnt_forward_hip(a, labels, act_lens, label_lens, blank, fastemit_lambda, need_backward=need)
        ctx.cfg = (blank, reduction, fastemit_lambda, clamp, bool(own_acts), acts.size(0))
        if need:
            ctx.state = (a, labels, act_lens, label_lens, ws)
        if reduction in ['sum', 'mean']:
            costs = costs.sum().unsqueeze_(-1)
            if reduction == 'mean':
                costs /= acts.size(0)
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        state = getattr(ctx, "state", None)
        if grad_output is None or state is None:
            return (None,) * 11
        a, labels, act_lens, label_lens, ws = state
        blank, reduction, fastemit_lambda, clamp, own, B = ctx.cfg
        go = grad_output.reshape(-1).float()
        if reduction in ['sum', 'mean']:
            go = go.expand(B) / (B if reduction == 'mean' else 1)
        g = rnnt_backward_hip(a, labels, act_lens, label_lens, blank, ws, cost_grad=go.contiguous(),
                              fastemit_lambda=fastemit_lambda, clamp=clamp, inplace=own)
        ctx.state = None
        return (g,) + (None,) * 10


class RNNTLossHIP(torch.nn.Module):
    """Drop-in for RNNTLossNumba on MI355X."""

    def __init__(self, blank=0, reduction='mean', fastemit_lambda: float = 0.0, clamp: float = -1):
        super().__init__()
        self.blank = blank
        self.fastemit_lambda = fastemit_lambda
        self.clamp = float(clamp) if clamp > 0 else 0.0
        self.reduction = reduction
        self.loss = _RNNTHip.apply

    def forward(self, acts, labels, act_lens, label_lens, max_T=None, max_U=None, own_acts=False):
        """`own_acts=True`: the caller promises nothing else reads `acts` after this loss, so backward() writes the
        gradient in place over it (saves one lattice-sized allocation)."""
        return self.loss(acts, labels, act_lens, label_lens, self.blank, self.reduction, self.fastemit_lambda,
                         self.clamp, max_T, max_U, own_acts)


class RNNTLoss(torch.nn.Module):
    """A/losses/rnnt.py:333-508 restated for the HIP loss: int64 casts, fp32 up-cast, narrowing to the
    batch maxima, dynamic `reduction` with `reduce()` over (lists of) per-utterance losses."""

    def __init__(self, num_classes, reduction: str = 'mean_batch', loss_name: str = "default", loss_kwargs=None):
        super().__init__()
        if reduction not in [None, 'mean', 'sum', 'mean_batch', 'mean_volume']:
            raise ValueError('`reduction` must be one of [mean, sum, mean_batch, mean_volume]')
        if loss_name not in ("default", "warprnnt_numba", "hip"):
            raise NotImplementedError(f"loss_name={loss_name}: only the standard transducer loss is on the hot path")
        kw = dict(loss_kwargs or {})
        self._blank = num_classes
        self.reduction = reduction
        self._loss = RNNTLossHIP(blank=self._blank, reduction='none', fastemit_lambda=kw.get('fastemit_lambda', 0.0),
                                 clamp=kw.get('clamp', -1.0))
        self._force_float32 = True

    def reduce(self, losses, target_lengths):
        if isinstance(losses, List):
            losses = torch.cat(losses, 0)
            target_lengths = torch.cat(target_lengths, 0)
        if self.reduction == 'mean_batch':
            losses = losses.mean()
        elif self.reduction == 'mean':
            losses = torch.div(losses, target_lengths).mean()
        elif self.reduction == 'sum':
            losses = losses.sum()
        elif self.reduction == 'mean_volume':
            losses = losses.sum() / target_lengths.sum()
        return losses

    def forward(self, log_probs, targets, input_lengths, target_lengths, max_T: Optional[int] = None,
                max_U: Optional[int] = None):
        targets = targets.long()
        input_lengths = input_lengths.long()
        target_lengths = target_lengths.long()
        if max_T is None:
            max_T = int(input_lengths.max())
        if max_U is None:
            max_U = int(target_lengths.max())
        own = False
        if log_probs.dtype != torch.float32:
            log_probs = log_probs.float()  # private fp32 copy: the gradient may overwrite it in place
            own = True
        if log_probs.shape[1] != max_T:
            log_probs = log_probs.narrow(dim=1, start=0, length=max_T).contiguous()
        if not targets.is_contiguous():
            targets = targets.contiguous()
        if targets.shape[1] != max_U:
            targets = targets.narrow(dim=1, start=0, length=max_U).contiguous()
        loss = self._loss(log_probs, targets, input_lengths, target_lengths, max_T, max_U, own)
        if self.reduction is not None:
            loss = self.reduce(loss, target_lengths)
        return loss
