"""Fused joint + transducer loss on the matrix cores (csrc/joint_fwd.hip, joint_bwd.hip, rnnt_loss.hip).

fused_joint_rnnt(f, g, W, bias, ...) -> per-utterance costs [B], with gradients to f, g, W, bias.
Semantics = RNNTJoint.joint_after_projection (A/modules/rnnt.py:1587-1665) followed by the transducer loss on the
accelerator branch (raw logits in, fused log-softmax gradient out), computed like the reference's AMP mode: f16
operands, f32 accumulation, logits rounded to f16, loss in f32.
"""
import math

import os

import torch

from .. import _lib

JOINT_MAX_V = 272
# csrc/joint_dh.hip fuses the dH GEMM with its mask and reductions (1.4 ms against 1.4 + 0.6 ms for the library GEMM +
# ia_joint_dh_reduce at bs32 x 15 s); False selects the unfused path (also the fallback for H % 80 != 0).
MFMA_PROFILE_HOOK = None   # bench.py: callable(flops) -> (start, stop) torch events recorded around the hidden-gradient kernel
USE_FUSED_DH = True
# csrc/joint_dw.hip regenerates the hidden tile in LDS and contracts it with G row-major (transposing LDS reads): no
# hidden^T tensor, no transposed copy of G.  False selects hidden^T + the batched split-K library GEMM.
DW_SKIP_DEAD_FRAMES = os.environ.get("IA_DW_SKIP", "1") != "0"   # the weight-gradient kernel skips the steps behind each utterance's last frame
USE_FUSED_DW = True
# Recorded right after the HBM-bound gradient kernel of the latest backward: work that other streams run under the joint's
# backward (the CTC branch, losses/ctc.py) waits for it, so that kernel has the memory system to itself and the MFMA-bound
# hidden- / weight-gradient kernels are the ones that share the GPU.
LAST_GRAD_KERNEL_EVENT = None


def fused_joint_supported(H, V, device):
    return device.type == "cuda" and H % 64 == 0 and H >= 64 and V <= JOINT_MAX_V


def _kappa_for(scale_hint):
    s = abs(float(scale_hint)) if scale_hint else 1.0
    return float(2.0 ** math.floor(-math.log2(s))) if s > 0 else 1.0


# ------------------------------------------------------------------------------------------------------------------------
# The sub-batch logits stash of the CL scripts (joint.store_list, A/modules/rnnt.py:1463-1496) on the fused path.
def _range_push(stash, kind, sums4):
    """max |z| / max t of a term's reduction -> pinned host memory behind an event (read when the backward sizes kappa)."""
    host = torch.empty(4, dtype=torch.float32, pin_memory=True)
    host.copy_(sums4.detach(), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    stash.ranges[kind] = (host, ev)


class LatticeStash:
    """What `joint.store_list` holds after a fused forward with store_sub_enc / store_sub_logits set: the whole batch's raw
    logits as ONE f16 lattice [B*T*U1, LD] (the buffer the loss read -- no clones) plus, per utterance, the extent of
    the sub-batch box the reference would have stashed it in.  len() = number of sub-batches; iterating materialises the
    reference's per-sub-batch tensors [b, max_t, max_u+1, V] (debugging / third-party code: detached copies).
    cl.mas_importance_loss / cl.lwf_kd_loss evaluate their terms on the lattice itself (csrc/joint_extra.hip)."""

    def __init__(self, logits, geom, boxes, weights, n_sub, sub_size, token, detached):
        self.logits = logits                  # [cells, LD] f16; None once the joint's backward has consumed it
        self.B, self.T, self.U1, self.V, self.LD = geom
        self.box_t, self.box_u, self.box_um1 = boxes          # device int64 [B]
        self.w_sq, self.w_kd = weights                        # device f32 [B]
        self.n_sub, self.sub_size = n_sub, sub_size
        self.token = token                    # autograd handle of the lattice (None when detached / no_grad)
        self.detached = detached
        self.pending_E = None                 # kappa * d(extra terms)/d logits, f16, accumulated by the terms' backward
        self.kappa = None
        self.scale_hint = 1.0                 # |d loss / d cost_b| the transducer part of the gradient is expected to carry
        self.host_boxes = None                # (box_t, box_u) host lists
        self.host_weights = (0.0, 0.0)        # (max w_sq, max w_kd)
        self.ranges = {}                      # kind -> (pinned [4] f32, event) of the terms evaluated on this lattice

    def __len__(self):
        return self.n_sub

    def _live(self):
        if self.logits is None:
            raise RuntimeError("joint.store_list: this lattice was overwritten by the joint's backward (the fused path keeps "
                               "one f16 lattice per forward and turns it into its gradient in place); evaluate "
                               "mas_importance_loss / lwf_kd_loss before calling backward")
        return self.logits

    def __iter__(self):
        z = self._live().view(self.B, self.T, self.U1, self.LD)
        bt, bu = self.host_boxes
        for b0 in range(0, self.B, self.sub_size):
            yield z[b0:b0 + self.sub_size, :bt[b0], :bu[b0], :self.V].detach().clone()

    def same_geometry(self, other):
        return (self.B, self.T, self.U1, self.V, self.LD, self.n_sub) == (other.B, other.T, other.U1, other.V, other.LD, other.n_sub)


class _LatticeTerm(torch.autograd.Function):
    """One weighted reduction over a stash (kind 0: sum of squares, kind 1: KL against a teacher lattice).  Its backward
    does not return a gradient tensor for the lattice (that would be a second lattice-sized round trip): it leaves
    kappa * d term / d logits in stash.pending_E, which the joint's backward -- ordered after this node by `token` --
    adds to the transducer gradient before its MFMA kernels run."""

    @staticmethod
    def forward(ctx, token, stash, teacher, kind):
        L = _lib.lib()
        z = stash._live()
        dev = z.device
        sums = torch.empty(4, dtype=torch.float32, device=dev)
        scr = torch.empty(L.ia_joint_extra_scratch_elems(), dtype=torch.float32, device=dev)
        t = teacher._live() if teacher is not None else None
        st = L.ia_joint_extra_reduce(_lib.ptr(z), _lib.ptr(t), _lib.ptr(stash.box_t), _lib.ptr(stash.box_u),
                                     _lib.ptr(stash.w_sq) if kind == 0 else None, _lib.ptr(stash.w_kd) if kind == 1 else None,
                                     stash.B, stash.T, stash.U1, stash.V, stash.LD, _lib.ptr(sums), _lib.ptr(scr),
                                     _lib.stream_ptr())
        _lib.check(st, "ia_joint_extra_reduce")
        ctx.stash, ctx.teacher, ctx.kind = stash, teacher, kind
        _range_push(stash, kind, sums)
        return sums[kind].clone()

    @staticmethod
    def backward(ctx, gval):
        L = _lib.lib()
        stash, teacher, kind = ctx.stash, ctx.teacher, ctx.kind
        z = stash._live()
        dev = z.device
        if stash.kappa is None:
            # power-of-two gradient scale: the largest of |d rnnt|, |d sq| = w 2|z|, |d kd| = w e^t lands near 2^6
            # (the one host wait of a MAS / LwF step: the reductions of THIS forward have to have finished.  A stale
            #  reading from an earlier step would do most of the time and fail badly when the workload changes.)
            big = abs(float(stash.scale_hint))
            for k, (host, ev) in stash.ranges.items():
                ev.synchronize()
                big = max(big, stash.host_weights[0] * 2.0 * float(host[2]) if k == 0
                          else stash.host_weights[1] * math.exp(min(float(host[3]), 80.0)))
            stash.kappa = float(2.0 ** math.floor(math.log2(64.0 / max(big, 1e-30))))
        ups = torch.zeros(2, dtype=torch.float32, device=dev)
        ups[kind] = gval.float() * stash.kappa
        t = teacher._live() if teacher is not None else None
        E = torch.empty_like(z)
        st = L.ia_joint_extra_grad(_lib.ptr(z), _lib.ptr(t), _lib.ptr(stash.box_t), _lib.ptr(stash.box_u),
                                   _lib.ptr(stash.w_sq) if kind == 0 else None, _lib.ptr(stash.w_kd) if kind == 1 else None,
                                   _lib.ptr(ups), stash.B, stash.T, stash.U1, stash.V, stash.LD, _lib.ptr(E), _lib.stream_ptr())
        _lib.check(st, "ia_joint_extra_grad")
        if stash.pending_E is None:
            stash.pending_E = E
        else:
            st = L.ia_lattice_add_f16(_lib.ptr(stash.pending_E), _lib.ptr(E), E.numel(), _lib.stream_ptr())
            _lib.check(st, "ia_lattice_add_f16")
        return gval.new_zeros(()), None, None, None


def lattice_sumsq_term(stash: LatticeStash):
    """(1/n_sub) sum_sub mean_{cells of the sub-batch box} sum_v z^2  (R/cl_baseline_mas.py:260-263), differentiable."""
    if stash.token is None:
        tok = torch.zeros((), dtype=torch.float32, device=stash._live().device)
    else:
        tok = stash.token
    return _LatticeTerm.apply(tok, stash, None, 0)


def lattice_kd_term(stash: LatticeStash, teacher: LatticeStash):
    """(1/n_sub) sum_sub F.kl_div(z_sub, exp(t_sub), 'batchmean')  (R/cl_baseline_lwf.py:249-257), differentiable in z."""
    if not stash.same_geometry(teacher):
        raise ValueError("lwf: teacher and student lattices come from different batches")
    tok = stash.token if stash.token is not None else torch.zeros((), dtype=torch.float32, device=stash._live().device)
    return _LatticeTerm.apply(tok, stash, teacher, 1)


class _FusedJointRNNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, g, W, bias, labels, act_lens, label_lens, blank, dropout_p, seed, fastemit, scale_hint, stash_req,
                We=None, be=None, Wg=None, bg=None):
        """Projection mode (We .. bg given): `f` is the encoder output x [B,T,d] (f32, contiguous) and `g` the prediction
        network's output [B,U1,Hp] (f32; a batch-major VIEW of the LSTM's time-major [U1,B,Hp] buffer, or contiguous); the
        joint's two projections (rnnt.py:1590-1596) run inside this node on the HIP GEMM with f16 outputs, so that no cast /
        gather / transpose launches sit between them and the lattice kernels, forward or backward."""
        from ..losses import rnnt as rl
        from . import fast
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        proj = We is not None
        dev = f.device
        need = any(ctx.needs_input_grad[:4]) or (proj and any(ctx.needs_input_grad[13:17]))
        p = float(dropout_p)
        V = W.shape[0]
        ctx.proj = None
        if proj:
            from . import tail
            B, T, d = f.shape
            U1, Hp = g.shape[1], g.shape[2]
            H = We.shape[0]
            xb = tail.bf16_of(f if f.is_contiguous() else f.contiguous())                       # [B*T, d] (shared with the CTC head)
            gd = g.detach()
            tm = gd.transpose(0, 1)
            if (not gd.is_contiguous()) and tm.is_contiguous() and gd.dtype in (torch.float32, torch.bfloat16):
                gb, g_time_major = fast.swap01_cast(tm, torch.bfloat16).view(B * U1, Hp), True    # [U1,B,Hp] -> [B,U1,Hp] bf16
            else:
                gb, g_time_major = gd.contiguous().to(torch.bfloat16).view(B * U1, Hp), False
            We16, Wg16 = fast.bf16_shadow(We), fast.bf16_shadow(Wg)
            f16 = fast.gemm(xb, We16, be.detach(), out_f16=True)[1].view(B, T, H)
            g16 = fast.gemm(gb, Wg16, bg.detach(), out_f16=True)[1].view(B, U1, H)
            ctx.proj = (xb, gb, We16, Wg16, g_time_major, d, Hp, We, be, Wg, bg)
        else:
            B, T, H = f.shape
            U1 = g.shape[1]
            f16 = f.detach().to(torch.float16).contiguous()
            g16 = g.detach().to(torch.float16).contiguous()
        LD = L.ia_joint_ld(V)
        # head image: rows padded to 272, dropout scale folded in, f16 -- and its transpose for the hidden-gradient kernel -- by ONE
        # launch (was zeros + div + cast + copy here and zeros + transposed copy in the backward)
        Wp = torch.empty(JOINT_MAX_V, H, dtype=torch.float16, device=dev)
        dhk = L.ia_joint_dh_k()
        Wt = torch.empty(H, dhk, dtype=torch.float16, device=dev) if (need and dhk >= JOINT_MAX_V) else None
        W32 = W.detach() if (W.dtype == torch.float32 and W.is_contiguous()) else W.detach().float().contiguous()
        _lib.check(L.ia_select_rows_cast(_lib.ptr(W32), H, None, 0, V, -1, H, JOINT_MAX_V, 1.0 / (1.0 - p) if p > 0 else 1.0, 1,
                                         _lib.ptr(Wp), _lib.ptr(Wt), dhk, None, _lib.stream_ptr()), "ia_select_rows_cast")
        bias32 = bias.detach().float().contiguous()
        nbytes = L.ia_rnnt_workspace_bytes(B, T, U1)
        if nbytes == 0:
            raise RuntimeError("fused joint: unsupported lattice size (U1 <= 1024)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        logits = torch.empty(B * T * U1, LD, dtype=torch.float16, device=dev)
        stash = None
        if stash_req is not None:
            stash = _make_stash(stash_req, logits, (B, T, U1, V, LD), dev)
            stash.scale_hint = scale_hint if scale_hint else 1.0
            stash_req["out"] = stash
        st = L.ia_joint_fwd_box(_lib.ptr(f16), _lib.ptr(g16), _lib.ptr(Wp), _lib.ptr(bias32), _lib.ptr(labels),
                                _lib.ptr(act_lens), _lib.ptr(label_lens), _lib.ptr(stash.box_t) if stash else None,
                                _lib.ptr(stash.box_u) if stash else None, B, T, U1, H, V, int(blank), p,
                                int(seed) & 0xFFFFFFFF, _lib.ptr(logits), LD, _lib.ptr(ws), nbytes, _lib.stream_ptr())
        _lib.check(st, "ia_joint_fwd_box")
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        st = L.ia_rnnt_lattice(_lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, float(fastemit), int(need),
                               _lib.ptr(costs), _lib.ptr(ws), nbytes, _lib.stream_ptr())
        _lib.check(st, "ia_rnnt_lattice")
        ctx.stash = stash
        if need:
            ctx.saved = (f16, g16, Wp, logits, ws, labels, act_lens, label_lens)
            ctx.Wt = Wt
            ctx.head = (W, bias)
            ctx.meta = (B, T, U1, H, V, LD, int(blank), p, int(seed) & 0xFFFFFFFF, float(fastemit),
                        _kappa_for(scale_hint), f.dtype, g.dtype, W.dtype, bias.dtype, nbytes)
        # the second output is the autograd handle of the lattice: the CL terms hang their nodes on it (_LatticeTerm)
        return costs, (torch.zeros((), dtype=torch.float32, device=dev) if stash is not None else None)

    @staticmethod
    def _rnnt_gradient(ctx, L, rl, gcosts, logits, ws, labels, act_lens, label_lens, fused_dw, kappa, skip_dead=False):
        """logits -> kappa * d loss / d logits in place (csrc/joint_bwd.hip); returns (G, kappa*dbias | None, GT, S, Kc)."""
        B, T, U1, H, V, LD, blank, p, seed, fastemit, _, fdt, gdt, wdt, bdt, nbytes = ctx.meta
        dev = logits.device
        cells = B * T * U1
        cg = gcosts.reshape(-1).float().contiguous()
        GT, S, Kc = None, 0, 0
        if not fused_dw:
            # split-K layout for the library weight-gradient GEMM: S chunks of Kc lattice cells, both operands K-contiguous
            S = 64
            Kc = ((cells + S - 1) // S + 63) // 64 * 64
            GT = torch.empty(S, LD, Kc, dtype=torch.float16, device=dev)
        hook = None
        if rl.PROFILE_HOOK is not None:   # bench.py: HIP events around the gradient kernel on its launch stream
            hook = rl.PROFILE_HOOK(B, T, U1, V, 2, 2 if fused_dw else 3, "joint_grad_h_db_kernel" if fused_dw else "joint_grad_h_t_kernel",
                                   bool(skip_dead))
        ev0, ev1 = hook if hook is not None else (None, None)
        dbk = dbscr = None
        if fused_dw:   # the gradient kernel also returns kappa * dbias (register column sums, partial rows in dbscr)
            dbk = torch.empty(LD, dtype=torch.float32, device=dev)
            dbscr = torch.empty(L.ia_joint_backward_g_dbias_scratch_elems(LD), dtype=torch.float32, device=dev)
        # tiles behind an utterance's last frame: not touched when nothing reads them (fused hidden- and weight-gradient kernels
        # that know the frame counts, no continual-learning term to be added over the sub-batch boxes)
        st = L.ia_joint_backward_g_skip(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, V,
                                        LD, blank, fastemit, _lib.ptr(cg), kappa, _lib.ptr(GT), S, Kc, _lib.ptr(dbk), _lib.ptr(dbscr),
                                        _lib.ptr(ws), nbytes, int(bool(skip_dead)), _lib.stream_ptr(), ev0, ev1)
        _lib.check(st, "ia_joint_backward_g_skip")
        global LAST_GRAD_KERNEL_EVENT
        LAST_GRAD_KERNEL_EVENT = torch.cuda.Event()
        LAST_GRAD_KERNEL_EVENT.record()
        G = logits  # [cells, LD] f16, = kappa * dL/dlogits
        return G, dbk, GT, S, Kc

    @staticmethod
    def backward(ctx, gcosts, gtoken):
        from ..losses import rnnt as rl
        L = _lib.lib()
        stash = ctx.stash
        E = stash.pending_E if (stash is not None and gtoken is not None) else None
        if gcosts is None and E is None:
            return (None,) * 17
        if ctx.saved is None:   # the gradient overwrites the saved logits in place: one backward per forward
            raise RuntimeError("fused joint+loss: trying to backward through the graph a second time -- the saved lattice "
                               "was overwritten in place by the first backward (the fused path does not support "
                               "retain_graph=True; run the forward again)")
        f16, g16, Wp, logits, ws, labels, act_lens, label_lens = ctx.saved
        B, T, U1, H, V, LD, blank, p, seed, fastemit, kappa, fdt, gdt, wdt, bdt, nbytes = ctx.meta
        ctx.saved = None
        dev = f16.device
        cells = B * T * U1
        fused_dw = USE_FUSED_DW and cells < 2 ** 31 and L.ia_joint_dw_fused_supported(U1, H, LD)
        if stash is not None:
            stash.logits = None   # consumed below
        if E is not None:
            kappa = stash.kappa
            stash.pending_E = None
            if not (fused_dw and USE_FUSED_DH and L.ia_joint_dh_fused_supported(U1, H, LD)):
                raise RuntimeError("fused joint: the continual-learning terms need the fused hidden- and weight-gradient "
                                   "kernels (H % 320 == 0); set joint.use_fused = False for this shape")
        have_rnnt = gcosts is not None
        dbk = None
        if not have_rnnt:
            # importance pass (R/cl_baseline_mas.py:258-266): only the extra terms are differentiated -- their gradient IS
            # the lattice gradient, the transducer's gradient kernel does not run
            G = E
            dbk = torch.sum(E, 0, dtype=torch.float32)
            GT, S, Kc = None, 0, 0
        else:
            skip_dead = (E is None and fused_dw and DW_SKIP_DEAD_FRAMES and USE_FUSED_DH
                         and bool(L.ia_joint_dh_fused_supported(U1, H, LD)))
            G, dbk, GT, S, Kc = _FusedJointRNNT._rnnt_gradient(ctx, L, rl, gcosts, logits, ws, labels, act_lens, label_lens,
                                                               fused_dw, kappa, skip_dead)
            if E is not None:
                st = L.ia_lattice_add_f16(_lib.ptr(G), _lib.ptr(E), G.numel(), _lib.stream_ptr())
                _lib.check(st, "ia_lattice_add_f16")
                dbk = dbk + torch.sum(E, 0, dtype=torch.float32)
        if E is not None:   # the extra terms live on the sub-batch boxes, not only on the valid lattice
            act_lens, label_lens = stash.box_t, stash.box_um1
        del E
        LDH = H + 8
        dh_fused = bool(USE_FUSED_DH and L.ia_joint_dh_fused_supported(U1, H, LD))
        proj_mode = ctx.proj is not None
        dfb = dgb = None
        if dh_fused and proj_mode:
            # projection mode: only the bf16 images are needed (operands of the projections' backward GEMMs); the finishing
            # kernel zeroes the dead rows itself: no fp32 d f / d g tensors, no memsets, no casts
            df = dg = None
            dfb = torch.empty(B * T, H, dtype=torch.bfloat16, device=dev)
            dgb = torch.empty(B * U1, H, dtype=torch.bfloat16, device=dev)
        else:
            df = torch.zeros(B, T, H, dtype=torch.float32, device=dev)
            dg = torch.zeros(B, U1, H, dtype=torch.float32, device=dev)
        if dh_fused:
            # dH = G @ W, relu/dropout mask and both reductions in one MFMA kernel (dH never reaches memory)
            Wt = ctx.Wt
            if Wt is None:
                Wt = torch.zeros(H, L.ia_joint_dh_k(), dtype=torch.float16, device=dev)
                Wt[:, :JOINT_MAX_V] = Wp.t()
            scr = torch.empty(L.ia_joint_dh_fused_scratch_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
            ev = MFMA_PROFILE_HOOK(2.0 * cells * H * LD) if MFMA_PROFILE_HOOK is not None else None   # bench.py: torch events
            if ev is not None:
                ev[0].record()
            st = L.ia_joint_dh_fused_ex(_lib.ptr(G), _lib.ptr(Wt), _lib.ptr(f16), _lib.ptr(g16), _lib.ptr(act_lens),
                                        _lib.ptr(label_lens), _lib.ptr(df), _lib.ptr(dg), _lib.ptr(dfb), _lib.ptr(dgb),
                                        1 if dfb is not None else 0, B, T, U1, H, LD, 1.0 / kappa, p, seed, _lib.ptr(scr),
                                        _lib.stream_ptr())
            _lib.check(st, "ia_joint_dh_fused_ex")
            if ev is not None:
                ev[1].record()
        else:
            dH = torch.mm(G, Wp[:LD])  # [cells, H] f16 (plain library GEMM)
            scr = torch.empty(L.ia_joint_dh_reduce_scratch_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
            st = L.ia_joint_dh_reduce(_lib.ptr(dH), _lib.ptr(f16), _lib.ptr(g16), _lib.ptr(act_lens), _lib.ptr(label_lens),
                                      _lib.ptr(df), _lib.ptr(dg), B, T, U1, H, 1.0 / kappa, p, seed, _lib.ptr(scr),
                                      _lib.stream_ptr())
            _lib.check(st, "ia_joint_dh_reduce")
            del dH
        if fused_dw:
            dWk = torch.empty(LD, H, dtype=torch.float32, device=dev)
            scr = torch.empty(L.ia_joint_dw_fused_scratch_elems(B, T, U1, H, LD), dtype=torch.float32, device=dev)
            # dead frames AND dead labels skipped (8 x 8-cell tiles over each utterance's live box) when the lengths are passed
            st = L.ia_joint_dw_fused_ex(_lib.ptr(G), _lib.ptr(f16), _lib.ptr(g16), _lib.ptr(act_lens) if DW_SKIP_DEAD_FRAMES else None,
                                        _lib.ptr(label_lens) if DW_SKIP_DEAD_FRAMES else None, B, T, U1, H, LD, p, seed,
                                        _lib.ptr(dWk), _lib.ptr(scr), _lib.stream_ptr())
            _lib.check(st, "ia_joint_dw_fused_ex")
            dW_src, dW_scale = dWk[:V], 1.0 / (kappa * (1.0 - p))
            db_src, db_scale = dbk[:V], 1.0 / kappa
        else:
            HT = torch.empty(S, LDH, Kc, dtype=torch.float16, device=dev)
            st = L.ia_joint_hidden_t(_lib.ptr(f16), _lib.ptr(g16), _lib.ptr(HT), B, T, U1, H, LDH, S, Kc, p, seed,
                                     _lib.stream_ptr())
            _lib.check(st, "ia_joint_hidden_t")
            # dW (+dbias in column H): batched split-K library GEMM over the chunks, f32 partials summed
            dWx = torch.bmm(GT, HT.transpose(1, 2), out_dtype=torch.float32).sum(0)  # [LD, LDH]
            dW_src, dW_scale = dWx[:V, :H].contiguous(), 1.0 / (kappa * (1.0 - p))
            db_src, db_scale = dWx[:V, H].contiguous(), 1.0 / kappa
        from . import fast, tail
        Wh, bh = ctx.head
        need = ctx.needs_input_grad
        plist = [(Wh, dW_src if need[2] else None, dW_scale), (bh, db_src if need[3] else None, db_scale)]
        if ctx.proj is None:
            outs = tail.accumulate_or_return(plist)
            return (df.to(fdt), dg.to(gdt)) + outs + (None,) * 13
        # ---- projection mode: the two projections' backward on the HIP GEMMs, every parameter gradient in one accumulation launch
        xb, gb, We16, Wg16, g_time_major, d, Hp, We, be, Wg, bg = ctx.proj
        ctx.proj = None
        if dfb is None:
            dfb = df.view(B * T, H).to(torch.bfloat16)
            dgb = dg.view(B * U1, H).to(torch.bfloat16)
        WeT, WgT = fast.transpose16_multi([We16, Wg16])                  # [d, H], [Hp, H]: "weights" of dX = dY W
        dx = dgin = None
        if need[0]:
            dx = torch.empty(B * T, d, dtype=torch.float32, device=dev)
            fast.gemm(dfb, WeT, out_f32=dx, want_bf16=False)
            dx = dx.view(B, T, d).to(fdt)
        if need[1]:
            dgm = fast.gemm(dgb, WgT)[1].view(B, U1, Hp)                    # batch-major bf16
            if g_time_major:   # hand the gradient back in the layout the prediction network's output has: [U1,B,Hp] storage
                dgin = fast.swap01_cast(dgm, torch.float32 if gdt == torch.float32 else torch.bfloat16).transpose(0, 1).to(gdt)
            else:
                dgin = dgm.to(gdt)
        (dWe, dbe), (dWg, dbg) = fast.gemm_tn_grouped([(dfb, xb), (dgb, gb)])
        plist += [(We, dWe if need[13] else None, 1.0), (be, dbe if need[14] else None, 1.0),
                  (Wg, dWg if need[15] else None, 1.0), (bg, dbg if need[16] else None, 1.0)]
        outs = tail.accumulate_or_return(plist)
        return (dx, dgin) + outs[:2] + (None,) * 9 + outs[2:]


def _make_stash(req, logits, geom, dev):
    """Box extents and weights of the reference's sub-batch loop (A/modules/rnnt.py:1436-1447: each sub-batch is narrowed
    to its own max T / max U) from the host-side lengths: utterance b of sub-batch s gets box (max_t_s, max_u_s + 1),
    w_sq = 1 / (n_sub * b_s * max_t_s * (max_u_s+1)) (mean over cells, mean over sub-batches) and w_kd = 1 / (n_sub * b_s)
    ('batchmean', mean over sub-batches).  One pinned H2D copy each for the integers and the floats."""
    import numpy as np
    B, T, U1, V, LD = geom
    sub, h_enc, h_tgt = int(req["sub"]), req["h_enc"], req["h_tgt"]
    n_sub = (B + sub - 1) // sub
    ints = np.empty((3, B), dtype=np.int64)
    w = np.empty((2, B), dtype=np.float32)
    for b0 in range(0, B, sub):
        b1 = min(b0 + sub, B)
        mt, mu1 = min(max(h_enc[b0:b1]), T), min(max(h_tgt[b0:b1]) + 1, U1)
        ints[0, b0:b1], ints[1, b0:b1], ints[2, b0:b1] = mt, mu1, mu1 - 1
        w[0, b0:b1] = 1.0 / (n_sub * (b1 - b0) * mt * mu1)
        w[1, b0:b1] = 1.0 / (n_sub * (b1 - b0))
    di = torch.from_numpy(ints).pin_memory().to(dev, non_blocking=True)
    dw = torch.from_numpy(w).pin_memory().to(dev, non_blocking=True)
    st = LatticeStash(logits, geom, (di[0], di[1], di[2]), (dw[0], dw[1]), n_sub, sub, None, bool(req.get("detach")))
    st.host_boxes = (ints[0].tolist(), ints[1].tolist())
    st.host_weights = (float(w[0].max()), float(w[1].max()))
    return st


def fused_joint_rnnt(f, g, W, bias, labels, act_lens, label_lens, blank, dropout_p=0.0, seed=0, fastemit_lambda=0.0,
                     scale_hint=1.0, stash_req=None, enc_proj=None, pred_proj=None):
    """f [B,T,H], g [B,U1,H] (any float dtype), W [V,H], bias [V] -> costs [B] f32 (differentiable).
    stash_req = {"sub": fused_batch_size, "h_enc": [...], "h_tgt": [...], "detach": bool}: also keeps the logits of every
    sub-batch box and leaves a LatticeStash in stash_req["out"].
    enc_proj = (weight [H,d], bias [H]) and pred_proj = (weight [H,Hp], bias [H]): projection mode -- `f` is then the encoder
    output x [B,T,d] and `g` the prediction network's output [B,U1,Hp]; the joint's enc / pred Linear layers run inside the node."""
    extra = (None, None, None, None)
    if enc_proj is not None:
        extra = (enc_proj[0], enc_proj[1], pred_proj[0], pred_proj[1])
    costs, token = _FusedJointRNNT.apply(f, g, W, bias, labels.contiguous(), act_lens.contiguous(), label_lens.contiguous(),
                                         blank, dropout_p, seed, fastemit_lambda, scale_hint, stash_req, *extra)
    if stash_req is not None and not stash_req.get("detach") and token is not None and token.requires_grad:
        stash_req["out"].token = token
    return costs
