"""Fused joint + transducer loss on the matrix cores (csrc/joint_fwd.hip, joint_bwd.hip, rnnt_loss.hip).

fused_joint_rnnt(f, g, W, bias, ...) -> per-utterance costs [B], with gradients to f, g, W, bias.
Semantics = RNNTJoint.joint_after_projection (A/modules/rnnt.py:1587-1665) followed by the transducer loss on the
accelerator branch (raw logits in, fused log-softmax gradient out), computed like the reference's AMP mode: f16
operands, f32 accumulation, logits rounded to f16, loss in f32.
"""
import math

import torch

from .. import _lib

JOINT_MAX_V = 272
# csrc/joint_dh.hip fuses the dH GEMM with its mask and reductions (1.4 ms against 1.4 + 0.6 ms for the library GEMM +
# ia_joint_dh_reduce at bs32 x 15 s); False selects the unfused path (also the fallback for H % 80 != 0).
MFMA_PROFILE_HOOK = None   # bench.py: callable(flops) -> (start, stop) torch events recorded around the hidden-gradient kernel
USE_FUSED_DH = True
# csrc/joint_dw.hip regenerates the hidden tile in LDS and contracts it with G row-major (transposing LDS reads): no
# hidden^T tensor, no transposed copy of G.  False selects hidden^T + the batched split-K library GEMM.
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


class _FusedJointRNNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, g, W, bias, labels, act_lens, label_lens, blank, dropout_p, seed, fastemit, scale_hint):
        from ..losses import rnnt as rl
        L = _lib.lib()
        B, T, H = f.shape
        U1 = g.shape[1]
        V = W.shape[0]
        dev = f.device
        need = any(ctx.needs_input_grad[:4])
        f16 = f.detach().to(torch.float16).contiguous()
        g16 = g.detach().to(torch.float16).contiguous()
        p = float(dropout_p)
        LD = L.ia_joint_ld(V)
        Wp = torch.zeros(JOINT_MAX_V, H, dtype=torch.float16, device=dev)
        Wp[:V] = (W.detach().float() / (1.0 - p)).to(torch.float16) if p > 0 else W.detach().to(torch.float16)
        bias32 = bias.detach().float().contiguous()
        nbytes = L.ia_rnnt_workspace_bytes(B, T, U1)
        if nbytes == 0:
            raise RuntimeError("fused joint: unsupported lattice size (U1 <= 1024)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        logits = torch.empty(B * T * U1, LD, dtype=torch.float16, device=dev)
        st = L.ia_joint_fwd(_lib.ptr(f16), _lib.ptr(g16), _lib.ptr(Wp), _lib.ptr(bias32), _lib.ptr(labels),
                            _lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, H, V, int(blank), p, int(seed) & 0xFFFFFFFF,
                            _lib.ptr(logits), LD, _lib.ptr(ws), nbytes, _lib.stream_ptr())
        _lib.check(st, "ia_joint_fwd")
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        st = L.ia_rnnt_lattice(_lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, float(fastemit), int(need),
                               _lib.ptr(costs), _lib.ptr(ws), nbytes, _lib.stream_ptr())
        _lib.check(st, "ia_rnnt_lattice")
        if need:
            ctx.saved = (f16, g16, Wp, logits, ws, labels, act_lens, label_lens)
            ctx.meta = (B, T, U1, H, V, LD, int(blank), p, int(seed) & 0xFFFFFFFF, float(fastemit),
                        _kappa_for(scale_hint), f.dtype, g.dtype, W.dtype, bias.dtype, nbytes)
        return costs

    @staticmethod
    def backward(ctx, gcosts):
        from ..losses import rnnt as rl
        L = _lib.lib()
        if ctx.saved is None:   # the gradient overwrites the saved logits in place: one backward per forward
            raise RuntimeError("fused joint+loss: trying to backward through the graph a second time -- the saved lattice "
                               "was overwritten in place by the first backward (the fused path does not support "
                               "retain_graph=True; run the forward again)")
        f16, g16, Wp, logits, ws, labels, act_lens, label_lens = ctx.saved
        B, T, U1, H, V, LD, blank, p, seed, fastemit, kappa, fdt, gdt, wdt, bdt, nbytes = ctx.meta
        ctx.saved = None
        dev = f16.device
        cg = gcosts.reshape(-1).float().contiguous()
        cells = B * T * U1
        fused_dw = USE_FUSED_DW and cells < 2 ** 31 and L.ia_joint_dw_fused_supported(U1, H, LD)
        GT, S, Kc = None, 0, 0
        if not fused_dw:
            # split-K layout for the library weight-gradient GEMM: S chunks of Kc lattice cells, both operands K-contiguous
            S = 64
            Kc = ((cells + S - 1) // S + 63) // 64 * 64
            LDH = H + 8
            GT = torch.empty(S, LD, Kc, dtype=torch.float16, device=dev)
        hook = None
        if rl.PROFILE_HOOK is not None:   # bench.py: HIP events around the gradient kernel on its launch stream
            hook = rl.PROFILE_HOOK(B, T, U1, V, 2, 2 if fused_dw else 3, "joint_grad_h_db_kernel" if fused_dw else "joint_grad_h_t_kernel")
        ev0, ev1 = hook if hook is not None else (None, None)
        dbk = dbscr = None
        if fused_dw:   # the gradient kernel also returns kappa * dbias (register column sums, partial rows in dbscr)
            dbk = torch.empty(LD, dtype=torch.float32, device=dev)
            dbscr = torch.empty(L.ia_joint_backward_g_dbias_scratch_elems(LD), dtype=torch.float32, device=dev)
        st = L.ia_joint_backward_g(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(act_lens), _lib.ptr(label_lens), B, T, U1, V,
                                   LD, blank, fastemit, _lib.ptr(cg), kappa, _lib.ptr(GT), S, Kc, _lib.ptr(dbk), _lib.ptr(dbscr),
                                   _lib.ptr(ws), nbytes, _lib.stream_ptr(), ev0, ev1)
        _lib.check(st, "ia_joint_backward_g")
        global LAST_GRAD_KERNEL_EVENT
        LAST_GRAD_KERNEL_EVENT = torch.cuda.Event()
        LAST_GRAD_KERNEL_EVENT.record()
        G = logits  # [cells, LD] f16, = kappa * dL/dlogits
        df = torch.zeros(B, T, H, dtype=torch.float32, device=dev)
        dg = torch.zeros(B, U1, H, dtype=torch.float32, device=dev)
        if USE_FUSED_DH and L.ia_joint_dh_fused_supported(U1, H, LD):
            # dH = G @ W, relu/dropout mask and both reductions in one MFMA kernel (dH never reaches memory)
            Wt = torch.zeros(H, L.ia_joint_dh_k(), dtype=torch.float16, device=dev)
            Wt[:, :JOINT_MAX_V] = Wp.t()
            scr = torch.empty(L.ia_joint_dh_fused_scratch_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
            ev = MFMA_PROFILE_HOOK(2.0 * cells * H * LD) if MFMA_PROFILE_HOOK is not None else None   # bench.py: torch events
            if ev is not None:
                ev[0].record()
            st = L.ia_joint_dh_fused(_lib.ptr(G), _lib.ptr(Wt), _lib.ptr(f16), _lib.ptr(g16), _lib.ptr(act_lens),
                                     _lib.ptr(label_lens), _lib.ptr(df), _lib.ptr(dg), B, T, U1, H, LD, 1.0 / kappa, p, seed,
                                     _lib.ptr(scr), _lib.stream_ptr())
            _lib.check(st, "ia_joint_dh_fused")
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
            st = L.ia_joint_dw_fused(_lib.ptr(G), _lib.ptr(f16), _lib.ptr(g16), B, T, U1, H, LD, p, seed, _lib.ptr(dWk),
                                     _lib.ptr(scr), _lib.stream_ptr())
            _lib.check(st, "ia_joint_dw_fused")
            dW = dWk[:V] * (1.0 / (kappa * (1.0 - p)))
            db = dbk[:V] * (1.0 / kappa)
        else:
            HT = torch.empty(S, LDH, Kc, dtype=torch.float16, device=dev)
            st = L.ia_joint_hidden_t(_lib.ptr(f16), _lib.ptr(g16), _lib.ptr(HT), B, T, U1, H, LDH, S, Kc, p, seed,
                                     _lib.stream_ptr())
            _lib.check(st, "ia_joint_hidden_t")
            # dW (+dbias in column H): batched split-K library GEMM over the chunks, f32 partials summed
            dWx = torch.bmm(GT, HT.transpose(1, 2), out_dtype=torch.float32).sum(0)  # [LD, LDH]
            dW = dWx[:V, :H] * (1.0 / (kappa * (1.0 - p)))
            db = dWx[:V, H] * (1.0 / kappa)
        return df.to(fdt), dg.to(gdt), dW.to(wdt), db.to(bdt), None, None, None, None, None, None, None, None


def fused_joint_rnnt(f, g, W, bias, labels, act_lens, label_lens, blank, dropout_p=0.0, seed=0, fastemit_lambda=0.0,
                     scale_hint=1.0):
    """f [B,T,H], g [B,U1,H] (any float dtype), W [V,H], bias [V] -> costs [B] f32 (differentiable)."""
    return _FusedJointRNNT.apply(f, g, W, bias, labels.contiguous(), act_lens.contiguous(), label_lens.contiguous(),
                                 blank, dropout_p, seed, fastemit_lambda, scale_hint)
