"""Continual-learning state and arithmetic on flat device buffers (EWC / MAS / LwF + AdamW + DP exchange).

Reference call sites restated here (R/ = reference repo root):
  get_params / get_params_clone / get_zero_params / get_grads / set_grads ... R/utils.py:273-321
  get_penalty_grads (EWC) ............................................... R/cl_baseline_ewc.py:69-81
  Fisher epoch ........................................................... R/cl_baseline_ewc.py:245-282
  penalty (MAS) + importance epoch ...................................... R/cl_baseline_mas.py:70-75,212-288
  LwF distillation ....................................................... R/cl_baseline_lwf.py:212-264
  AdamW(lr) / zero_grad / step ........................................... R/cl_baseline.py:137,187-196

MI355X design: every trainable parameter lives in ONE flat fp32 buffer (`FlatParams.theta`), `.grad`s are views
of ONE flat gradient buffer, so the penalty, Fisher/omega accumulation, AdamW and the RCCL all-reduce are each
a single launch / a single collective over ~N*4 bytes instead of ~300 per-tensor kernels.
The dict-of-tensors API of the reference is kept: the dicts handed out are `FlatDict`s (name -> view).
"""
import math
from contextlib import contextmanager
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib

_ALIGN = 64  # floats: every tensor starts on a 256-byte boundary inside the flat buffers


class FlatDict(dict):
    """name -> view into `.flat` (a 1-D fp32 device tensor laid out by `.layout`)."""

    def __init__(self, layout: "FlatParams", flat: torch.Tensor):
        super().__init__()
        self.layout, self.flat = layout, flat
        for name, off, numel, shape in layout.entries:
            self[name] = flat[off:off + numel].view(shape)


class FlatParams:
    def __init__(self, model: torch.nn.Module):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("model has no trainable parameters")
        named = _qkv_adjacent(named)
        dev = named[0][1].device
        self.model = model
        self.entries, off = [], 0
        for n, p in named:
            if p.dtype != torch.float32:
                raise TypeError(f"{n}: master parameters must be float32")
            self.entries.append((n, off, p.numel(), tuple(p.shape)))
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        self.theta = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self._theta_views = FlatDict(self, self.theta)
        self._grad_views = FlatDict(self, self.grad)
        with torch.no_grad():
            for (n, o, k, shape), p in zip(self.entries, self.params):
                self._theta_views[n].copy_(p.data)
                p.data = self._theta_views[n]
                p.grad = self._grad_views[n]
        # chunk table for ia_cl_penalty (chunks never straddle tensors)
        ch = _lib.lib().ia_cl_chunk_elems() if dev.type == "cuda" else 4096
        rows = []
        for seg, (n, o, k, shape) in enumerate(self.entries):
            for c0 in range(0, k, ch):
                rows.append((o + c0, min(ch, k - c0), seg, 0))
        self.chunk_table = torch.tensor(rows, dtype=torch.int32, device=dev)
        self.seg_inv_numel = torch.tensor([1.0 / e[2] for e in self.entries], dtype=torch.float32, device=dev)
        # True once something gave EVERY trainable tensor a gradient since the last zero_grad (a pre-loaded EWC penalty /
        # the MAS penalty: R/utils.py:316-321, R/cl_baseline_mas.py:231-234) -- torch.optim.AdamW then updates them all
        self.all_grads_live = False
        model._ia_flat = self

    # -- buffers ---------------------------------------------------------------------------------
    def zeros(self) -> FlatDict:
        return FlatDict(self, torch.zeros_like(self.theta))

    def clone_theta(self) -> FlatDict:
        flush_pending_updates()
        return FlatDict(self, self.theta.clone())

    def params_dict(self) -> FlatDict:
        flush_pending_updates()
        return self._theta_views

    def grads_dict(self) -> FlatDict:
        return self._grad_views

    def attach_grads(self):
        for n, p in zip(self.names, self.params):
            p.grad = self._grad_views[n]

    def zero_grad(self):
        self.grad.zero_()
        self.all_grads_live = False
        self.attach_grads()

    @contextmanager
    def weights(self, other: FlatDict):
        """Run with the parameters temporarily pointing at another flat buffer (LwF teacher forward) -- no
        torch.save/torch.load ping-pong and no barriers (R/cl_baseline_lwf.py:220-234 does both per batch)."""
        from .ops import fast
        flush_pending_updates()
        try:
            for n, p in zip(self.names, self.params):
                p.data = other[n]
            fast.bump_weight_epoch()
            yield
        finally:
            for n, p in zip(self.names, self.params):
                p.data = self._theta_views[n]
            fast.bump_weight_epoch()


def _qkv_adjacent(named):
    """Layout order of the flat buffers: as named_parameters(), except that each attention module's linear_q | linear_k | linear_v
    weights (and then their biases) are placed back to back, so the [3d, d] operand of the fused Q/K/V projection and its bias
    are VIEWS of the flat weight / bf16 shadow buffers (ops/fast.bf16_shadow, f32_cat) instead of three-way concatenations
    rebuilt after every optimizer step.  The dict-of-names surface (FlatDict) does not depend on the order."""
    by_name = dict(named)
    order, done = [], set()
    for n, p in named:
        if n in done:
            continue
        if n.endswith("linear_q.weight") or n.endswith("linear_q.bias"):
            kind = n.rsplit(".", 1)[1]
            stem = n[: -len("linear_q." + kind)]
            group = [stem + f"linear_{x}.{kind}" for x in ("q", "k", "v")]
            if all(g in by_name for g in group):
                for g in group:
                    order.append((g, by_name[g])); done.add(g)
                continue
        order.append((n, p)); done.add(n)
    return order


def flat_of(model) -> FlatParams:
    m = getattr(model, "module", model)
    f = getattr(m, "_ia_flat", None)
    if f is None:
        f = FlatParams(m)
    return f


# ----------------------------------------------------------------------------- R/utils.py:273-321 drop-ins
def get_params(model) -> FlatDict:
    return flat_of(model).params_dict()


def get_params_clone(model) -> FlatDict:
    return flat_of(model).clone_theta()


def get_zero_params(model, device=None) -> FlatDict:
    return flat_of(model).zeros()


def get_grads(model) -> FlatDict:
    return flat_of(model).grads_dict()


def set_grads(model, grad_dict):
    """R/utils.py:316-321.  A FlatDict produced by get_penalty_grads already IS the flat gradient buffer."""
    flush_pending_updates()
    flush_pending_updates()
    f = flat_of(model)
    if isinstance(grad_dict, FlatDict) and grad_dict.flat.data_ptr() != f.grad.data_ptr():
        f.grad.copy_(grad_dict.flat)
        grad_dict = f.grads_dict()
    f.all_grads_live = True
    for name, p in getattr(model, "module", model).named_parameters():
        p.grad = grad_dict[name] if name in grad_dict else None


def save_model(model, path):
    """R/utils.py:265-271: trainable-only state dict (same interchange format)."""
    flush_pending_updates()
    torch.save({n: p.data.clone() for n, p in getattr(model, "module", model).named_parameters() if p.requires_grad}, path)


# ----------------------------------------------------------------------------- EWC
def ewc_penalty_into_grads(flat: FlatParams, fisher: FlatDict, checkpoint: FlatDict, e_lambda: float,
                           monitor_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grad <- 2*lambda*F*(theta-theta*) (pre-load, autograd accumulates on top: R/cl_baseline_ewc.py:228-231);
    returns a 0-dim device tensor = mean_k mean|penalty_k| (the 'ewc_penalty' monitor, :74-80)."""
    flush_pending_updates()
    L = _lib.lib()
    seg = torch.zeros(len(flat.entries), dtype=torch.float32, device=flat.theta.device)
    st = L.ia_cl_penalty(_lib.ptr(flat.theta), _lib.ptr(checkpoint.flat), _lib.ptr(fisher.flat), 2.0 * float(e_lambda),
                         _lib.ptr(flat.grad), 0, _lib.ptr(flat.chunk_table), flat.chunk_table.shape[0],
                         _lib.ptr(flat.seg_inv_numel), _lib.ptr(seg), None, _lib.stream_ptr())
    _lib.check(st, "ia_cl_penalty")
    flat.attach_grads()
    flat.all_grads_live = True
    return seg.mean()


def get_penalty_grads(config, fish: FlatDict, curr_checkpoint: FlatDict, checkpoint: FlatDict):
    """Signature of R/cl_baseline_ewc.py:69: returns (grad dict, python float)."""
    flat = fish.layout
    avg = ewc_penalty_into_grads(flat, fish, checkpoint, config.cl_config.e_lambda)
    return flat.grads_dict(), avg.item()


def fisher_accumulate(flat: FlatParams, fish: FlatDict, loss: torch.Tensor):
    """fish += mean(loss) * grad**2 (R/cl_baseline_ewc.py:245-255), loss stays on the device."""
    s = loss.detach().float().mean().reshape(1).contiguous()
    st = _lib.lib().ia_cl_fisher_accumulate(_lib.ptr(fish.flat), _lib.ptr(flat.grad), _lib.ptr(s), flat.numel,
                                            _lib.stream_ptr())
    _lib.check(st, "ia_cl_fisher_accumulate")


def fisher_finish(main_fish: Optional[FlatDict], fish: FlatDict, total_ds: int, e_gamma: float,
                  group=None) -> FlatDict:
    """fish /= N; main = gamma*main + fish (R/cl_baseline_ewc.py:267-280).  Under DP the rank-local Fisher sums and
    sample counts are all-reduced first (one RCCL collective over the flat buffer) so the result equals the
    1-GPU Fisher over the union of the shards (the reference keeps rank-local dicts)."""
    n = torch.tensor([float(total_ds)], device=fish.flat.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(fish.flat, group=group)
        dist.all_reduce(n, group=group)
    fish.flat.div_(n)
    if main_fish is None:
        return fish
    main_fish.flat.mul_(e_gamma).add_(fish.flat)
    return main_fish


# ----------------------------------------------------------------------------- MAS
def mas_penalty_add_grads(flat: FlatParams, importance: FlatDict, checkpoint: FlatDict, mas_lambda: float):
    """Adds d/dtheta [mas_lambda * sum omega (theta-theta*)^2] to the flat gradient buffer and returns the
    un-weighted penalty value as a 0-dim device tensor ('mass_loss', R/cl_baseline_mas.py:70-75,231-234).
    The reference obtains the same gradient through autograd on `loss + mass_loss*mas_lambda`."""
    flush_pending_updates()
    val = torch.zeros(1, dtype=torch.float32, device=flat.theta.device)
    st = _lib.lib().ia_cl_penalty(_lib.ptr(flat.theta), _lib.ptr(checkpoint.flat), _lib.ptr(importance.flat),
                                  2.0 * float(mas_lambda), _lib.ptr(flat.grad), 1, _lib.ptr(flat.chunk_table),
                                  flat.chunk_table.shape[0], None, None, _lib.ptr(val), _lib.stream_ptr())
    _lib.check(st, "ia_cl_penalty")
    flat.all_grads_live = True
    return val[0]


def penalty(model, main_importance: FlatDict, prev_params: FlatDict):
    """Signature of R/cl_baseline_mas.py:70.  Returns the penalty VALUE (no autograd graph); pair it with
    mas_penalty_add_grads() after backward, or use MASRegulariser which does both."""
    flush_pending_updates()
    flat = flat_of(model)
    val = torch.zeros(1, dtype=torch.float32, device=flat.theta.device)
    st = _lib.lib().ia_cl_penalty(_lib.ptr(flat.theta), _lib.ptr(prev_params.flat), _lib.ptr(main_importance.flat), 0.0,
                                  None, 0, _lib.ptr(flat.chunk_table), flat.chunk_table.shape[0], None, None,
                                  _lib.ptr(val), _lib.stream_ptr())
    _lib.check(st, "ia_cl_penalty")
    return val[0]


def mas_importance_loss(model, mas_ctx: float):
    """R/cl_baseline_mas.py:258-265 on the stashed raw logits (joint.store_list, ctc_decoder.decoder_logits)."""
    m = getattr(model, "module", model)
    from .ops.joint import LatticeStash, lattice_sumsq_term
    decoder_logits = (m.ctc_decoder.decoder_logits.flatten(end_dim=-2).float() ** 2).sum(dim=-1).mean()
    if isinstance(m.joint.store_list, LatticeStash):   # fused joint: one streaming pass over the f16 lattice
        rnn_logits = lattice_sumsq_term(m.joint.store_list)
    else:
        rnn_logits = 0
        for i in m.joint.store_list:
            rnn_logits = rnn_logits + (i.flatten(end_dim=-2).float() ** 2).sum(dim=-1).mean()
        rnn_logits = rnn_logits / len(m.joint.store_list)
    return rnn_logits * (1 - mas_ctx) + decoder_logits * mas_ctx


def importance_accumulate(flat: FlatParams, importance: FlatDict):
    st = _lib.lib().ia_cl_abs_accumulate(_lib.ptr(importance.flat), _lib.ptr(flat.grad), flat.numel, _lib.stream_ptr())
    _lib.check(st, "ia_cl_abs_accumulate")


def importance_finish(importance: FlatDict, n_batches: int, group=None) -> FlatDict:
    """omega /= #batches (R/cl_baseline_mas.py:284-287; overwrites the previous task's omega as the reference
    does).  Under DP: all-reduce(SUM) of omega and of the batch counts first."""
    n = torch.tensor([float(n_batches)], device=importance.flat.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(importance.flat, group=group)
        dist.all_reduce(n, group=group)
    importance.flat.div_(n)
    return importance


# ----------------------------------------------------------------------------- LwF
def lwf_kd_loss(loss, prob, prob_, pred_store_list, store_list, knowledge_distillation: float, kd_ctx: float):
    """R/cl_baseline_lwf.py:242-264.  Returns (total loss, rnnt_kd, ctc_kd) -- device tensors."""
    F = torch.nn.functional
    # fp32 arithmetic whatever the stash dtype (the bf16 path stashes bf16 lattices; kl_div evaluated in bf16 rounds
    # log(exp(i)) - j, a difference of nearly equal numbers, to 8 bits: observed 4.5 % off the fp32 value)
    ctc_kd_loss = F.kl_div(prob.float(), prob_.float().exp(), reduction='batchmean')
    from .ops.joint import LatticeStash, lattice_kd_term
    if isinstance(store_list, LatticeStash) and isinstance(pred_store_list, LatticeStash):
        rnnt_kd = lattice_kd_term(pred_store_list, store_list)   # fused joint: both lattices stay f16 in HBM, two streaming passes
    else:
        if isinstance(store_list, LatticeStash) or isinstance(pred_store_list, LatticeStash):
            raise ValueError("lwf_kd_loss: one of the two stashes is a fused-joint lattice and the other a list of tensors; "
                             "run the teacher and the student passes on the same joint path (joint.use_fused)")
        assert len(store_list) == len(pred_store_list)
        rnnt_kd = 0
        for i, j in zip(store_list, pred_store_list):
            rnnt_kd = rnnt_kd + F.kl_div(j.float(), i.float().exp(), reduction='batchmean')
        rnnt_kd = rnnt_kd / len(store_list)
    total = loss * (1 - knowledge_distillation) + knowledge_distillation * ((1 - kd_ctx) * rnnt_kd + kd_ctx * ctc_kd_loss)
    return total, rnnt_kd, ctc_kd_loss


def lwf_teacher_forward(model, flat: FlatParams, teacher: FlatDict, batch, lang_ids, host_lengths=None):
    """Teacher pass with the previous task's weights resident in HBM (R/cl_baseline_lwf.py:213-232 semantics:
    no_grad, store_sub_enc + detach)."""
    flush_pending_updates()
    m = getattr(model, "module", model)
    with torch.no_grad(), flat.weights(teacher):
        m.joint.store_sub_enc, m.joint.detach_sub_enc = True, True
        step = m._step
        _, _, prob_ = m.training_step(batch, lang_ids, return_probs=True, host_lengths=host_lengths, compute_wer=False)
        m._step = step  # same SpecAugment/dither draw for the student pass
        store_list = m.joint.store_list
    return prob_, store_list


# ----------------------------------------------------------------------------- optimizer + DP
class FusedAdamW:
    """torch.optim.AdamW(model.parameters(), lr) of R/cl_baseline.py:137 as one launch over the flat buffers.
    step() first averages the flat gradient across ranks (RCCL all-reduce over xGMI) when a process group
    exists -- the reference wraps the model in DDP but never arms its reducer."""

    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, group=None,
                 bf16_shadow=None, defer_update=True, grad_exchange_dtype=None):
        """`grad_exchange_dtype="bf16"` ("fp32 or bf16"): the data-parallel exchange all-reduces a bf16 image of the
        flat gradient (half the bytes over xGMI: 80 instead of 160 MB per step at 40 M trainable parameters); every rank
        then applies AdamW to the same bf16-rounded sum, so the weights stay identical across ranks.  None: fp32 exchange."""
        self.flat = model_or_flat if isinstance(model_or_flat, FlatParams) else flat_of(model_or_flat)
        if grad_exchange_dtype not in (None, "fp32", "bf16"):
            raise ValueError("grad_exchange_dtype: None | 'fp32' | 'bf16'")
        self.grad_exchange_dtype = None if grad_exchange_dtype == "fp32" else grad_exchange_dtype
        self._g16 = None
        self.profile_exchange = False     # bench.py: HIP events around the wait for the exchange (exposed time per step)
        self.exchange_events = []
        self.exchange_bytes = 0           # bytes handed to the last all-reduce
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat.theta)
        self.exp_avg_sq = torch.zeros_like(self.flat.theta)
        self.step_count = 0
        # torch.optim.AdamW keeps one step counter per parameter and skips parameters whose .grad is None: per-tensor
        # counters + "received a gradient" flags live on the device (ia_adamw_step_segmented)
        nseg = len(self.flat.entries)
        self.seg_step = torch.zeros(nseg, dtype=torch.int32, device=self.flat.theta.device)
        self.seg_active = torch.zeros(nseg, dtype=torch.int32, device=self.flat.theta.device)
        self.group = group
        if (group is None and defer_update and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            # A communicator executes its collectives in issue order: on the default group the deferred 160 MB gradient
            # all-reduce (launched at step(), meant to run under the NEXT forward's frozen prefix) would sit in front of
            # that prefix's small SyncBatchNorm all-reduces and stall the first block until it has finished.  Its own
            # communicator (every rank constructs its optimizer at the same point of the program) lets both run side by side.
            self.group = dist.new_group()
        if bf16_shadow is None:  # the HIP GEMM paths consume bf16 weights: let the optimizer kernel emit them (one launch)
            bf16_shadow = self.flat.theta.is_cuda
        self.shadow = self.flat.theta.to(torch.bfloat16) if bf16_shadow else None
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=self.flat.params)]
        self.defer_update = defer_update   # data parallel only: overlap the gradient all-reduce with the next forward
        self._pending, self._zero_after_flush = None, False

    def zero_grad(self, set_to_none: bool = False):
        if self._pending is not None:
            self._zero_after_flush = True   # the deferred update still has to consume these gradients
        else:
            self.flat.zero_grad()

    def _exchange_buffer(self):
        """The tensor the all-reduce runs on: the flat fp32 gradient itself, or its bf16 image (copied back by _exchange_done)."""
        if self.grad_exchange_dtype == "bf16":
            if self._g16 is None:
                self._g16 = torch.empty_like(self.flat.grad, dtype=torch.bfloat16)
            self._g16.copy_(self.flat.grad)
            buf = self._g16
        else:
            buf = self.flat.grad
        self.exchange_bytes = buf.numel() * buf.element_size()
        return buf

    def _exchange_done(self):
        if self.grad_exchange_dtype == "bf16":
            self.flat.grad.copy_(self._g16)

    def allreduce_grads(self):
        if dist.is_available() and dist.is_initialized():
            ws = dist.get_world_size(self.group)
            if ws > 1:
                dist.all_reduce(self._exchange_buffer(), group=self.group)
                self._exchange_done()
                return 1.0 / ws
        return 1.0

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def step(self, grad_scale: Optional[float] = None):
        """Single process: AdamW now.  Data parallel with `defer_update` (default): the all-reduce of the flat gradient is
        launched asynchronously and the AdamW kernel is deferred to flush(), which the model calls right before the first
        module that reads trainable weights in the NEXT forward -- the exchange over xGMI then runs under the front end,
        subsampling and the frozen encoder prefix, none of which read trainable weights, so the result is bit-identical to
        updating immediately.  Everything here that reads or swaps the flat weights flushes first."""
        self.flush()
        ws = self._world()
        gs = 1.0 if grad_scale is None else grad_scale
        if ws > 1 and self.defer_update:
            work = dist.all_reduce(self._exchange_buffer(), group=self.group, async_op=True)
            self._pending = (work, gs / ws, self.flat.all_grads_live)
            _PENDING_OPTIMIZERS.add(self)
            return
        self._apply(self.allreduce_grads() * gs, self.flat.all_grads_live)

    def flush(self):
        if self._pending is None:
            return
        work, scale, live = self._pending
        self._pending = None
        _PENDING_OPTIMIZERS.discard(self)
        if self.profile_exchange and self.flat.theta.is_cuda:   # how long the compute stream stalls for the exchange
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            self.exchange_events.append((e0, e1))
        else:
            work.wait()
        self._exchange_done()
        self._apply(scale, live)
        if self._zero_after_flush:
            self._zero_after_flush = False
            self.flat.zero_grad()

    def _apply(self, scale, all_live=False):
        self.step_count += 1
        g = self.param_groups[0]
        f = self.flat
        st = _lib.lib().ia_adamw_step_segmented(
            _lib.ptr(f.theta), _lib.ptr(f.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), _lib.ptr(f.chunk_table),
            f.chunk_table.shape[0], _lib.ptr(self.seg_active), _lib.ptr(self.seg_step), len(f.entries), int(bool(all_live)),
            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
            float(scale), _lib.ptr(self.shadow), _lib.stream_ptr())
        _lib.check(st, "ia_adamw_step_segmented")
        if self.flat.theta.is_cuda:
            global LAST_UPDATE_EVENT
            LAST_UPDATE_EVENT = torch.cuda.Event()
            LAST_UPDATE_EVENT.record()
        from .ops import fast
        fast.bump_weight_epoch()  # the kernel rewrote theta by raw pointer: bf16 weight shadows are stale now
        if self.shadow is not None:  # ... and were re-made by the same kernel: hand the views to the shadow cache
            for (n, o, k, shape), q in zip(self.flat.entries, self.flat.params):
                if q.dim() >= 2:
                    fast.register_flat_shadow(q, self.shadow[o:o + k].view(shape[0], -1))


_PENDING_OPTIMIZERS = set()
LAST_UPDATE_EVENT = None   # recorded on the stream of the latest AdamW launch (side streams wait for it)


def flush_pending_updates():
    """Apply every deferred optimizer update (FusedAdamW.step under data parallelism).  Called by the model before the
    first module that reads trainable weights, and by every helper here that reads or swaps the flat weights."""
    for opt in list(_PENDING_OPTIMIZERS):
        opt.flush()
