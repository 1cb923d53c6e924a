"""Prediction-network LSTM on the persistent HIP kernels (csrc/lstm.hip); autograd wrapper.

Same arithmetic as torch.nn.LSTM(num_layers=1) with zero initial state (gate order i,f,g,o): bf16 operands for the
two projections (like the rest of the bf16 path), fp32 gates / cell state / outputs.
"""
import torch

from .. import _lib
from . import fast

MAXB = 32


_LDS_LIMIT = 160 * 1024


def lstm_supported(x, H):
    """Shapes the persistent kernels take (mirrors the launchers' checks, so that anything else falls back to nn.LSTM
    instead of raising IA_UNSUPPORTED): the W_hh slices of forward and backward must fit the 160 KiB LDS."""
    if not (x.is_cuda and H % 64 == 0 and H <= 4096):
        return False
    L = _lib.lib()
    return max(L.ia_lstm_lds_bytes(H, 0), L.ia_lstm_lds_bytes(H, 1)) <= _LDS_LIMIT


_SCRATCH = {}        # (device index, stream, direction) -> persistent scratch (hand-off buffers + sync words)
_STICKY_WORD = 32    # csrc/lstm.hip LS_STICKY_WORD: set by a kernel whose bounded spin gave up, never cleared by a launch


def _scratch(B, H, dev, which=0):
    """One scratch buffer per (device, stream, direction), kept alive so that its sticky timeout word can be polled
    after the step (timeout_flags); launches on one stream are ordered, so they may share it."""
    n = _lib.lib().ia_lstm_scratch_bytes(B, H)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, which)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = _SCRATCH[key] = torch.zeros(n, dtype=torch.uint8, device=dev)
    return buf, n


def timeout_flags(dev):
    """0-dim float tensor on `dev`: number of scratch buffers whose sticky timeout word is set (None when the persistent
    LSTM never ran there).  model.training_step ships it to the host with the loss values; StepMonitor raises on it."""
    words = [b.view(torch.int32)[_STICKY_WORD] for k, b in _SCRATCH.items() if k[0] == dev.index]
    if not words:
        return None
    return torch.stack(words).ne(0).sum().float()


def timeout_words(dev):
    """The sticky timeout words themselves (0-dim int32 views) of the scratch buffers on `dev` (at most 4): the loss
    combination kernel counts the non-zero ones (ops/tail.loss_combine) -- no stack / ne / sum launches."""
    return [b.view(torch.int32)[_STICKY_WORD] for k, b in _SCRATCH.items() if k[0] == dev.index][:4]


def raise_if_timed_out(dev=None):
    """Synchronous check (tests, end of an epoch): RuntimeError if a hand-off spin of the persistent LSTM ever gave up."""
    for k, b in _SCRATCH.items():
        if dev is not None and k[0] != torch.device(dev).index:
            continue
        if int(b.view(torch.int32)[_STICKY_WORD].item()) != 0:
            b.view(torch.int32)[_STICKY_WORD] = 0
            raise RuntimeError("persistent LSTM (csrc/lstm.hip): a workgroup hand-off timed out -- the prediction network's "
                               "outputs / gradients of that launch are invalid (results of this step must be discarded)")


class _LSTMHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        L = _lib.lib()
        U, B, H = x.shape
        dev = x.device
        xb = x.detach().to(torch.bfloat16).contiguous()            # (no launch when ops/tail.embed_sos produced it)
        wih, whh = fast.bf16_shadow(w_ih), fast.bf16_shadow(w_hh)  # the optimizer kernel's bf16 images: no cast launches
        bias = (b_ih.detach().float() + b_hh.detach().float()).contiguous()
        Gx = torch.empty(U * B, 4 * H, dtype=torch.float32, device=dev)
        fast.gemm(xb.view(U * B, H), wih, bias, out_f32=Gx, want_bf16=False)
        Gx = Gx.view(U, B, 4 * H)
        need = any(ctx.needs_input_grad)
        Hout = torch.empty(U, B, H, dtype=torch.float32, device=dev)
        gates = torch.empty(U, B, 4 * H, dtype=torch.float32, device=dev) if need else None
        Cs = torch.empty(U, B, H, dtype=torch.float32, device=dev) if need else None
        for b0 in range(0, B, MAXB):
            b1 = min(B, b0 + MAXB)
            sl = slice(b0, b1)
            whole = (b0 == 0 and b1 == B)
            gx = Gx if whole else Gx[:, sl].contiguous()
            ho = Hout if whole else torch.empty(U, b1 - b0, H, dtype=torch.float32, device=dev)
            ga = gates if (whole or not need) else torch.empty(U, b1 - b0, 4 * H, dtype=torch.float32, device=dev)
            cs = Cs if (whole or not need) else torch.empty(U, b1 - b0, H, dtype=torch.float32, device=dev)
            sc, n = _scratch(b1 - b0, H, dev, 0)
            st = L.ia_lstm_forward(_lib.ptr(gx), _lib.ptr(whh), _lib.ptr(ho), _lib.ptr(ga), _lib.ptr(cs), U, b1 - b0, H,
                                   _lib.ptr(sc), n, _lib.stream_ptr())
            _lib.check(st, "ia_lstm_forward")
            if not whole:
                Hout[:, sl] = ho
                if need:
                    gates[:, sl] = ga
                    Cs[:, sl] = cs
        if need:
            ctx.save_for_backward(xb, wih, whh, Hout, gates, Cs)
            ctx.dt = (x.dtype, w_ih.dtype, w_hh.dtype, b_ih.dtype, b_hh.dtype)
            ctx.params = (w_ih, w_hh, b_ih, b_hh)
        return Hout

    @staticmethod
    def backward(ctx, dH):
        L = _lib.lib()
        xb, wih, whh, Hout, gates, Cs = ctx.saved_tensors
        U, B, H = xb.shape
        dev = xb.device
        dH = dH.float().contiguous()
        whhT, wihT = fast.transpose16_multi([whh, wih])   # one launch: W_hh^T for the recurrence, W_ih^T for dX = dG W_ih
        dG = torch.empty(U, B, 4 * H, dtype=torch.float32, device=dev)
        for b0 in range(0, B, MAXB):
            b1 = min(B, b0 + MAXB)
            sl = slice(b0, b1)
            whole = (b0 == 0 and b1 == B)
            args = [t if whole else t[:, sl].contiguous() for t in (dH, gates, Cs)]
            out = dG if whole else torch.empty(U, b1 - b0, 4 * H, dtype=torch.float32, device=dev)
            sc, n = _scratch(b1 - b0, H, dev, 1)
            st = L.ia_lstm_backward(_lib.ptr(args[0]), _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(whhT), _lib.ptr(out), U,
                                    b1 - b0, H, _lib.ptr(sc), n, _lib.stream_ptr())
            _lib.check(st, "ia_lstm_backward")
            if not whole:
                dG[:, sl] = out
        xdt, wihdt, whhdt, bihdt, bhhdt = ctx.dt
        dGb = dG.view(U * B, 4 * H).to(torch.bfloat16)
        # the three dense contractions on the HIP GEMMs (a library TN GEMM took 0.29 ms for dW_ih alone: 16-40 workgroups)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = fast.gemm(dGb, wihT)[1].view(U, B, H).to(xdt)   # [H, 4H] weight image: dX = dG W_ih = dG (W_ih^T)^T
        dWih, db = fast.gemm_tn(dGb, xb.view(U * B, H))      # bias gradient = the column sums the same kernel returns
        if U > 1:
            hprev = Hout[:-1].reshape((U - 1) * B, H).to(torch.bfloat16)
            dWhh, _ = fast.gemm_tn(dGb[B:], hprev)
        else:
            dWhh = torch.zeros(4 * H, H, dtype=torch.float32, device=dev)
        # parameter gradients: added into the flat .grad buffers by ONE launch where they exist (ops/tail.py), returned otherwise
        from . import tail
        w_ih, w_hh, b_ih, b_hh = ctx.params
        need = ctx.needs_input_grad
        outs = tail.accumulate_or_return([(w_ih, dWih if need[1] else None, 1.0), (w_hh, dWhh if need[2] else None, 1.0),
                                          (b_ih, db if need[3] else None, 1.0), (b_hh, db if need[4] else None, 1.0)])
        return (dx,) + outs


def lstm_forward(x, lstm: torch.nn.LSTM):
    """x [U,B,H] -> (y [U,B,H] f32, (h_n, c_n) = None: the training step does not use the final state)."""
    return _LSTMHip.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
