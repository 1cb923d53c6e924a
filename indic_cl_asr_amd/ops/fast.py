"""Python faces of the Conformer-block HIP kernels (csrc/gemm_bf16.hip, csrc/encoder_ops.hip) and the bf16 weight
shadow cache used by the no-autograd (frozen-prefix / teacher / eval) encoder path."""
import ctypes
import weakref

import torch

from .. import _lib

# Caches below are keyed by id(parameter); every entry also holds weak references to its source parameters and is
# (a) only served while `ref() is param` for each of them -- a new Parameter that reuses a dead one's id() / storage /
# version can never be handed the dead model's bf16 weights -- and (b) dropped when a source parameter is collected, so
# teacher copies and per-task reloads do not pin HBM.  Frozen weights edited through `.data` do not move `_version`:
# call invalidate_weight_caches() after such an edit (checkpoint.load_weights does).
_SHADOW = {}
# Bumped whenever trainable weights change behind autograd's back (the fused AdamW kernel writes the flat buffer by
# raw pointer, so tensor._version does not move; cl.FlatParams.weights() re-points .data).
WEIGHT_EPOCH = 0


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def invalidate_weight_caches():
    """Forget every cached bf16 / concatenated / re-laid-out weight image (after in-place edits through `.data`)."""
    _SHADOW.clear(); _FLAT16.clear(); _BLOCK_PARAMS.clear(); _POS_PROJ.clear()
    bump_weight_epoch()


def _refs(cache, key, params):
    """Weak references to `params` whose death evicts cache[key]."""
    def drop(_ref, cache=cache, key=key):
        cache.pop(key, None)
    return tuple(weakref.ref(p, drop) for p in params)


def _same(refs, params):
    return len(refs) == len(params) and all(r() is p for r, p in zip(refs, params))



_FLAT16 = {}  # id(param) -> (weight epoch, param version, data_ptr, bf16 view written by the fused optimizer kernel)


def register_flat_shadow(p, view16):
    _FLAT16[id(p)] = (WEIGHT_EPOCH, p._version, p.data_ptr(), view16, _refs(_FLAT16, id(p), (p,)))


def _flat_shadow(p):
    hit = _FLAT16.get(id(p))
    if (hit is not None and hit[0] == WEIGHT_EPOCH and hit[1] == p._version and hit[2] == p.data_ptr()
            and _same(hit[4], (p,))):
        return hit[3]
    return None


def bf16_shadow(*params):
    """bf16 copy of a parameter (or of several concatenated along dim 0), re-made only when a source version changes.
    Frozen layers therefore pay the fp32->bf16 cast once, not once per step as autocast does."""
    key = tuple(id(p) for p in params)
    ver = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params) + \
        ((WEIGHT_EPOCH,) if any(p.requires_grad for p in params) else ())
    hit = _SHADOW.get(key)
    if hit is not None and hit[0] == ver and _same(hit[2], params):
        if len(hit) > 3:
            _join_cast(hit[1], hit[3])
        return hit[1]
    made = None
    with torch.no_grad():
        flat16 = [_flat_shadow(p) for p in params]
        if all(v is not None for v in flat16):  # emitted by ia_adamw_step: no cast kernels at all
            # (cl.FlatParams lays linear_q | linear_k | linear_v out back to back: their concatenation is a VIEW, not a cat)
            if len(params) == 1:
                w = flat16[0]
            else:
                w = _adjacent_rows_view(flat16)
                if w is None:
                    w = torch.cat(flat16, 0)
        else:
            w = torch.cat([p.detach().reshape(p.shape[0], -1) for p in params], 0) if len(params) > 1 \
                else params[0].detach().reshape(params[0].shape[0], -1)
            w = w.to(torch.bfloat16).contiguous()
            made = _cast_marker(w)
    _SHADOW[key] = (ver, w, _refs(_SHADOW, key, params)) + ((made,) if made is not None else ())
    return w


def _cast_marker(w):
    """The image was made by a cast kernel on the CURRENT stream, and the step uses several (prediction network, CTC branch, the
    deferred decode of the in-step WER each run on their own): a consumer on another stream that finds it in the cache has to wait
    for that kernel -- a decode that asked for joint.pred.weight's image first, on its side stream, otherwise raced the joint's own
    projection GEMM on the compute stream (first step after a weight load: 7e-4 off in the transducer loss at 32 x 15 s)."""
    if not w.is_cuda:
        return None
    st = torch.cuda.current_stream(w.device)
    ev = torch.cuda.Event()
    ev.record(st)
    return [st.cuda_stream, ev, set()]


def _join_cast(w, made):
    st = torch.cuda.current_stream(w.device)
    sid = st.cuda_stream
    if sid != made[0] and sid not in made[2]:
        st.wait_event(made[1])
        w.record_stream(st)          # (allocated from the casting stream's pool)
        made[2].add(sid)


def glu_regrouped(weight, bias):
    """pointwise_conv1 weight [2d, d(,1)] / bias [2d] with the output channels regrouped for the GLU epilogue of the HIP GEMM
    (ia_gemm_bf16_ex act 4): every 128 consecutive rows are 64 value channels followed by their 64 gate channels.  bf16 weight,
    f32 bias; cached per parameter version like bf16_shadow.  None when d % 64 != 0."""
    d2 = weight.shape[0]
    d = d2 // 2
    if d % 64 != 0:
        return None
    key = ("glu", id(weight), id(bias))
    ver = (weight._version, bias._version, weight.data_ptr(), bias.data_ptr()) + \
        ((WEIGHT_EPOCH,) if (weight.requires_grad or bias.requires_grad) else ())
    hit = _SHADOW.get(key)
    if hit is not None and hit[0] == ver and _same(hit[2], (weight, bias)):
        return hit[1]
    with torch.no_grad():
        t = torch.arange(d // 64, device=weight.device).view(-1, 1, 1) * 64
        j = torch.arange(64, device=weight.device).view(1, 1, -1)
        idx = torch.cat([t + j, d + t + j], dim=1).reshape(-1)            # [2d]: per tile 64 values then 64 gates
        w = weight.detach().reshape(d2, -1).float().index_select(0, idx).to(torch.bfloat16).contiguous()
        b = bias.detach().float().index_select(0, idx).contiguous()
    _SHADOW[key] = (ver, (w, b), _refs(_SHADOW, key, (weight, bias)))
    return w, b


def _adjacent_rows_view(ts):
    """[rows_i, cols] tensors lying back to back in one storage -> their row-wise concatenation as a view, else None."""
    t0 = ts[0]
    if t0.dim() not in (1, 2) or not all(t.is_contiguous() and t.dtype == t0.dtype and t.dim() == t0.dim() for t in ts):
        return None
    cols = t0.shape[1] if t0.dim() == 2 else 1
    off = t0.storage_offset()
    st = t0.untyped_storage().data_ptr()
    rows = 0
    for t in ts:
        tc = t.shape[1] if t.dim() == 2 else 1
        if t.untyped_storage().data_ptr() != st or t.storage_offset() != off + rows * cols or tc != cols:
            return None
        rows += t.shape[0]
    out = t0.new_empty(0).set_(t0.untyped_storage(), off, (rows, cols) if t0.dim() == 2 else (rows,), (cols, 1) if t0.dim() == 2 else (1,))
    return out


def f32_cat(*params):
    if len(params) > 1 and all(p.dtype == torch.float32 for p in params):
        v = _adjacent_rows_view([p.detach() for p in params])     # biases back to back in the flat weight buffer: a view
        if v is not None:
            return v
    key = ("f32",) + tuple(id(p) for p in params)
    ver = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params) + \
        ((WEIGHT_EPOCH,) if any(p.requires_grad for p in params) else ())
    hit = _SHADOW.get(key)
    if hit is not None and hit[0] == ver and _same(hit[2], params):
        return hit[1]
    with torch.no_grad():
        w = torch.cat([p.detach().float().reshape(-1) for p in params], 0).contiguous()
    _SHADOW[key] = (ver, w, _refs(_SHADOW, key, params))
    return w


_SCRATCH = {}


def scratch(dev, n):
    """Stream-ordered fp32 scratch for the two-stage reductions (workgroup partial rows): one buffer per (device,
    stream), reused by consecutive launches on that stream."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = _SCRATCH[key] = torch.empty(max(int(n), 1 << 20), dtype=torch.float32, device=dev)
    return buf


def gemm_supported(K, N):
    """ia_gemm_bf16 shapes: rows of 16-byte vectors (K, N multiples of 8); K need not be a multiple of the 64-wide k-tile
    (d_model = 144: the tail of the last tile reads as zero)."""
    return K % 8 == 0 and N % 8 == 0


def gemm(a_bf16, w_bf16, bias=None, act=0, dropout_p=0.0, seed=0, alpha=1.0, residual=None, out_f32=None,
         want_bf16=True, out_f16=False):
    """out = alpha*dropout(act(a @ w^T + bias)) + residual.  a [M,K] bf16, w [N,K] bf16.  Returns (out_f32, out_bf16);
    `out_f32` may be the residual tensor itself (in-place residual update).  out_f16: the 16-bit output is IEEE half (the
    joint's operands) instead of bf16."""
    M, K = a_bf16.shape
    N = w_bf16.shape[0]
    outH = torch.empty(M, N, dtype=torch.float16 if out_f16 else torch.bfloat16, device=a_bf16.device) if want_bf16 else None
    if out_f16:
        st = _lib.lib().ia_gemm_bf16_ex2(_lib.ptr(a_bf16), a_bf16.stride(0), _lib.ptr(w_bf16), w_bf16.stride(0), M, N, K,
                                         _lib.ptr(bias), int(act), float(dropout_p), int(seed) & 0xFFFFFFFF, float(alpha),
                                         _lib.ptr(residual), residual.stride(0) if residual is not None else 0,
                                         _lib.ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
                                         _lib.ptr(outH), N, None, 0, None, 0, 1, _lib.stream_ptr())
        _lib.check(st, "ia_gemm_bf16_ex2")
        return out_f32, outH
    st = _lib.lib().ia_gemm_bf16(_lib.ptr(a_bf16), a_bf16.stride(0), _lib.ptr(w_bf16), w_bf16.stride(0), M, N, K,
                                 _lib.ptr(bias), int(act), float(dropout_p), int(seed) & 0xFFFFFFFF, float(alpha),
                                 _lib.ptr(residual), residual.stride(0) if residual is not None else 0,
                                 _lib.ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
                                 _lib.ptr(outH), N, _lib.stream_ptr())
    _lib.check(st, "ia_gemm_bf16")
    return out_f32, outH


class _TrJob(ctypes.Structure):
    _fields_ = [("inp", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


def transpose16_multi(mats):
    """[M_i [rows, cols] 16-bit contiguous] (<= 8) -> [M_i^T contiguous] in ONE launch (csrc/tail_ops.hip)."""
    n = len(mats)
    arr = (_TrJob * n)()
    outs = []
    for i, m in enumerate(mats):
        assert m.is_contiguous() and m.element_size() == 2 and m.dim() == 2
        o = torch.empty(m.shape[1], m.shape[0], dtype=m.dtype, device=m.device)
        outs.append(o)
        arr[i].inp, arr[i].out, arr[i].rows, arr[i].cols = m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1]
    _lib.check(_lib.lib().ia_transpose16_multi(ctypes.addressof(arr), n, _lib.stream_ptr()), "ia_transpose16_multi")
    return outs


def swap01_cast(x, out_dtype):
    """x [n0, n1, H] contiguous (f32 or bf16) -> [n1, n0, H] contiguous in `out_dtype` (f32 or bf16): one launch."""
    n0, n1, H = x.shape
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16) and out_dtype in (torch.float32, torch.bfloat16)
    out = torch.empty(n1, n0, H, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().ia_swap01_cast(_lib.ptr(x), int(x.dtype == torch.bfloat16), n0, n1, H, _lib.ptr(out),
                                         int(out_dtype == torch.bfloat16), _lib.stream_ptr()), "ia_swap01_cast")
    return out


def quantize_fp8_rows(x):
    """x [M,K] bf16 or f32 (K % 8 == 0) -> (q [M, K16] e4m3 bytes, scale [M] f32): x ~ scale[m] * q  (csrc/gemm_fp8.hip)."""
    M, K = x.shape
    ldq = (K + 15) // 16 * 16
    q = torch.empty(M, ldq, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    st = _lib.lib().ia_quantize_fp8_rows(_lib.ptr(x), int(x.dtype == torch.float32), x.stride(0), M, K, _lib.ptr(q), ldq,
                                         _lib.ptr(sc), _lib.stream_ptr())
    _lib.check(st, "ia_quantize_fp8_rows")
    return q, sc


def fp8_shadow(*params):
    """(e4m3 rows, per-row scales) of the concatenated weights, cached per parameter version like bf16_shadow."""
    def make():
        w = torch.cat([p.detach().reshape(p.shape[0], -1).float() for p in params], 0).contiguous()
        return quantize_fp8_rows(w)
    return _cached(("fp8",) + tuple(id(p) for p in params), params, make)


def gemm_fp8(a, wq_ws, bias=None, act=0, dropout_p=0.0, seed=0, alpha=1.0, residual=None, out_f32=None, want_bf16=True):
    """gemm() with e4m3 operands: `a` [M,K] bf16 / f32 is quantised per row on the fly, wq_ws = fp8_shadow(weights)."""
    wq, ws = wq_ws
    M, K = a.shape
    N = wq.shape[0]
    aq, asc = quantize_fp8_rows(a)
    outH = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    st = _lib.lib().ia_gemm_fp8(_lib.ptr(aq), aq.stride(0), _lib.ptr(asc), _lib.ptr(wq), wq.stride(0), _lib.ptr(ws), M, N,
                                aq.shape[1], _lib.ptr(bias), int(act), float(dropout_p), int(seed) & 0xFFFFFFFF, float(alpha),
                                _lib.ptr(residual), residual.stride(0) if residual is not None else 0,
                                _lib.ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0, _lib.ptr(outH), N,
                                _lib.stream_ptr())
    _lib.check(st, "ia_gemm_fp8")
    return out_f32, outH


def quantize_mxfp8(x):
    """x [M,K] bf16 or f32 (K % 32 == 0) -> (q [M,K] e4m3 bytes, scales [M, K/32 rounded to 4] e8m0 bytes): MX blocks of 32
    consecutive k (csrc/gemm_mxfp8.hip)."""
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    lds = (K // 32 + 3) // 4 * 4
    sc = torch.empty(M, lds, dtype=torch.uint8, device=x.device)
    st = _lib.lib().ia_quantize_mxfp8(_lib.ptr(x), int(x.dtype == torch.float32), x.stride(0), M, K, _lib.ptr(q), K, _lib.ptr(sc), lds,
                                      _lib.stream_ptr())
    _lib.check(st, "ia_quantize_mxfp8")
    return q, sc


def mxfp8_supported(K, N):
    return K % 128 == 0 and N % 8 == 0


def mxfp8_shadow(*params):
    """MX-quantised image of the concatenated weights, cached per parameter version."""
    def make():
        w = torch.cat([p.detach().reshape(p.shape[0], -1).float() for p in params], 0).contiguous()
        return quantize_mxfp8(w)
    return _cached(("mxfp8",) + tuple(id(p) for p in params), params, make)


def gemm_mxfp8(a, wq_ws, bias=None, act=0, dropout_p=0.0, seed=0, alpha=1.0, residual=None, out_f32=None, want_bf16=True):
    """gemm() on block-scaled fp8 operands (the 2x-rate MFMA): `a` [M,K] bf16 / f32 is MX-quantised on the fly."""
    wq, ws = wq_ws
    M, K = a.shape
    N = wq.shape[0]
    aq, asc = quantize_mxfp8(a)
    outH = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    st = _lib.lib().ia_gemm_mxfp8(_lib.ptr(aq), K, _lib.ptr(asc), asc.stride(0), _lib.ptr(wq), K, _lib.ptr(ws), ws.stride(0), M, N, K,
                                  _lib.ptr(bias), int(act), float(dropout_p), int(seed) & 0xFFFFFFFF, float(alpha),
                                  _lib.ptr(residual), residual.stride(0) if residual is not None else 0,
                                  _lib.ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0, _lib.ptr(outH), N,
                                  _lib.stream_ptr())
    _lib.check(st, "ia_gemm_mxfp8")
    return out_f32, outH


def fp8_weights(*params):
    """("mx" | "row", q, scales): the fp8 image of the concatenated weights for gemm_fp8_any -- MX blocks where the kernel of
    the block-scaled MFMA applies (K % 128 == 0), per-row scales otherwise."""
    K = params[0].reshape(params[0].shape[0], -1).shape[1]
    if K % 128 == 0:
        return ("mx",) + tuple(mxfp8_shadow(*params))
    return ("row",) + tuple(fp8_shadow(*params))


def gemm_fp8_any(a, w3, *args, **kw):
    kind, q, sc = w3
    return (gemm_mxfp8 if kind == "mx" else gemm_fp8)(a, (q, sc), *args, **kw)


def layernorm(x_f32, g1, b1, eps=1e-5, out_f32=None, g2=None, b2=None, want_bf16=True):
    N, d = x_f32.shape
    outH = torch.empty(N, d, dtype=torch.bfloat16, device=x_f32.device) if want_bf16 else None
    st = _lib.lib().ia_layernorm(_lib.ptr(x_f32), x_f32.stride(0), N, d, _lib.ptr(g1), _lib.ptr(b1), float(eps),
                                 _lib.ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0, _lib.ptr(g2),
                                 _lib.ptr(b2), _lib.ptr(outH), d, _lib.stream_ptr())
    _lib.check(st, "ia_layernorm")
    return outH


USE_FUSED_FFN = True   # tests flip this to compare the row-resident module with the LayerNorm + two-GEMM sequence


def ffn_fused_supported(d, d_ff):
    return USE_FUSED_FFN and bool(_lib.lib().ia_ffn_fused_supported(int(d), int(d_ff)))


def ffn_fused(x_f32, ln, lin1, lin2, alpha, p_ff=0.0, seed_ff=0, p_res=0.0, seed_res=0, ln2=None, y_out=None, ln2_to_y_only=False):
    """x <- [ln2](x + alpha * dropout(lin2(dropout(SiLU(lin1(ln(x))))))) in place on the fp32 residual stream [N, d]:
    one launch of csrc/ffn_fused.hip (the [N, 4d] intermediate stays in LDS)."""
    N, d = x_f32.shape
    w1, w2 = bf16_shadow(lin1.weight), bf16_shadow(lin2.weight)
    st = _lib.lib().ia_ffn_fused(_lib.ptr(x_f32), N, d, w1.shape[0], _lib.ptr(ln.weight), _lib.ptr(ln.bias), float(ln.eps),
                                 _lib.ptr(w1), _lib.ptr(lin1.bias), _lib.ptr(w2), _lib.ptr(lin2.bias), float(alpha), float(p_ff),
                                 int(seed_ff) & 0xFFFFFFFF, float(p_res), int(seed_res) & 0xFFFFFFFF,
                                 _lib.ptr(ln2.weight) if ln2 is not None else None,
                                 _lib.ptr(ln2.bias) if ln2 is not None else None, _lib.ptr(y_out), int(bool(ln2_to_y_only)),
                                 _lib.stream_ptr())
    _lib.check(st, "ia_ffn_fused")
    return x_f32


def bn_sync_group(bn):
    """The process group a SyncBatchNorm module exchanges statistics over, or None when no exchange is due: a plain
    BatchNorm1d, an uninitialised / single-rank torch.distributed, or eval mode (running statistics).
    (torch.nn.SyncBatchNorm.convert_sync_batchnorm, R/cl_baseline.py:133; conformer_modules.py:322.)"""
    import torch.distributed as dist
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not bn.training:
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


def bn_module_ok(bn):
    """BatchNorm modules the HIP paths handle: nn.BatchNorm1d and what convert_sync_batchnorm makes of it."""
    return type(bn) in (torch.nn.BatchNorm1d, torch.nn.SyncBatchNorm)


SYNC_BN_COLLECTIVES = 0   # bench.py: SyncBatchNorm exchanges issued (forward sums; the blocks' backward exchanges count themselves)


def bn_sync_sums(sums_cnt, n_local, d, bn, group):
    """sums_cnt [2d+1] f32 = [sum | sumsq | -]: all-reduce over the ranks with the row count, running statistics from the
    global batch, sums rescaled so that sums / n_local are the GLOBAL moments (csrc/encoder_ops.hip bn_sync_finish)."""
    import torch.distributed as dist
    global SYNC_BN_COLLECTIVES
    SYNC_BN_COLLECTIVES += 1
    sums_cnt[2 * d:].fill_(float(n_local))
    dist.all_reduce(sums_cnt, group=group)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = bn.momentum if bn.momentum is not None else 0.1
    st = _lib.lib().ia_bn_sync_finish(_lib.ptr(sums_cnt), d, n_local, _lib.ptr(bn.running_mean) if track else None,
                                      _lib.ptr(bn.running_var) if track else None,
                                      _lib.ptr(bn.num_batches_tracked) if track else None, float(mom), _lib.stream_ptr())
    _lib.check(st, "ia_bn_sync_finish")


def glu_dwconv_bn_silu_fast(x2_bf16, lens, B, T, d, dw_weight, dw_bias, bn, training, keep=False):
    """x2 [B*T, 2d] bf16 -> [B*T, d] bf16 : GLU, pad mask, depthwise conv, BatchNorm (batch stats in training), SiLU.
    SyncBatchNorm with more than one rank: the per-channel sums are all-reduced between the two launches.
    keep=True also returns (z, sums) for a backward."""
    L = _lib.lib()
    dev = x2_bf16.device
    z = torch.empty(B * T, d, dtype=torch.float32, device=dev)
    sums = torch.empty(2 * d + 1, dtype=torch.float32, device=dev)
    ksz = dw_weight.shape[-1]
    st = L.ia_glu_dwconv(_lib.ptr(x2_bf16), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(dw_weight), _lib.ptr(dw_bias),
                         _lib.ptr(z), _lib.ptr(sums[:d]), _lib.ptr(sums[d:2 * d]),
                         _lib.ptr(scratch(dev, L.ia_dwconv_scratch_elems(B, T, d, ksz))), _lib.stream_ptr())
    _lib.check(st, "ia_glu_dwconv")
    out = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    use_batch = bool(training or not bn.track_running_stats)
    mom = bn.momentum if bn.momentum is not None else 0.1
    group = bn_sync_group(bn) if use_batch else None
    if group is not None:
        bn_sync_sums(sums, B * T, d, bn, group)     # (running statistics updated there, from the global batch)
        rm = rv = nbt = None
    else:
        rm, rv, nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
    st = L.ia_bn_silu(_lib.ptr(z), B * T, d, _lib.ptr(sums[:d]), _lib.ptr(sums[d:2 * d]), _lib.ptr(bn.weight), _lib.ptr(bn.bias),
                      _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(nbt), float(mom), float(bn.eps), int(use_batch), _lib.ptr(out),
                      _lib.stream_ptr())
    _lib.check(st, "ia_bn_silu")
    if keep:
        return out, z, sums
    return out


def attention_supported(T, dk):
    """The all-keys-in-registers kernel pair (forward + backward row pass) of the trainable blocks."""
    return dk == 64 and T <= 384


USE_FLASH_ATTN = True   # tests flip this to compare the two forward kernels


def attention_flash_supported(T, dk):
    return USE_FLASH_ATTN and bool(_lib.lib().ia_relpos_attention_flash_supported(int(T), int(dk)))


def relpos_attention_flash(qkv_bf16, pos_proj_bf16, bias_u, bias_v, lens, B, T, H, dk, dropout_p=0.0, seed=0, want_lse=False):
    """qkv [B*T, 3*H*dk] bf16, pos_proj [>= 2T-1, H*dk] bf16 -> ctx [B*T, H*dk] bf16: key-tile loop with online softmax
    (csrc/attention_flash.hip), any T, head dim <= 64.  want_lse: also the per-query log-sum-exp [B*H, T] f32 that
    relpos_attention_flash_bwd needs -> (ctx, lse)."""
    ctx = torch.empty(B * T, H * dk, dtype=torch.bfloat16, device=qkv_bf16.device)
    lse = torch.empty(B * H, T, dtype=torch.float32, device=qkv_bf16.device) if want_lse else None
    st = _lib.lib().ia_relpos_attention_flash_lse(_lib.ptr(qkv_bf16), _lib.ptr(pos_proj_bf16), _lib.ptr(bias_u), _lib.ptr(bias_v),
                                                  _lib.ptr(lens), B, T, H, dk, float(dropout_p), int(seed) & 0xFFFFFFFF,
                                                  _lib.ptr(ctx), _lib.ptr(lse), _lib.stream_ptr())
    _lib.check(st, "ia_relpos_attention_flash_lse")
    return (ctx, lse) if want_lse else ctx


def attention_flash_bwd_dims(T):
    import ctypes
    rs, p0 = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().ia_relpos_attention_flash_bwd_dims(int(T), ctypes.byref(rs), ctypes.byref(p0)), "dims")
    return rs.value, p0.value


def relpos_attention_flash_bwd(qkv, pl, bias_u, bias_v, lens, ctx, dctx, lse, B, T, H, dk, dropout_p=0.0, seed=0, dub_out=None):
    """Backward of relpos_attention_flash (csrc/attention_flash_bwd.hip): qkv [B*T,3d], pl [>=2T-1,d], ctx / dctx [B*T,d] bf16,
    lse [B*H,T] f32 -> (dqkv [B*T,3d] bf16, dpl [pl rows, d] bf16, dbias_u [H,dk] f32, dbias_v [H,dk] f32).  Two key-tiled
    kernels (query-owner: dq, bias gradients, band-skewed dS; key-owner: dK, dV) + one TN GEMM per head for dpl, all issued
    by one C call."""
    L = _lib.lib()
    dev = qkv.device
    d = H * dk
    bf = torch.bfloat16
    Rs, pad0 = attention_flash_bwd_dims(T)
    dqkv = torch.empty(B * T, 3 * d, dtype=bf, device=dev)
    dpl = torch.empty_like(pl)
    dBand = torch.empty(H * B * T * Rs, dtype=bf, device=dev)
    QvHM = torch.empty(H * B * T * 64, dtype=bf, device=dev)
    ws = torch.empty(L.ia_relpos_attention_flash_bwd_ws_elems(B, T, H, dk), dtype=torch.float32, device=dev)
    dub = dub_out if dub_out is not None else torch.empty(2, H, dk, dtype=torch.float32, device=dev)
    dctx = dctx.contiguous()
    st = L.ia_relpos_attention_flash_bwd(_lib.ptr(qkv), _lib.ptr(pl), _lib.ptr(bias_u), _lib.ptr(bias_v), _lib.ptr(lens), _lib.ptr(ctx),
                                         _lib.ptr(dctx), _lib.ptr(lse), B, T, H, dk, float(dropout_p), int(seed) & 0xFFFFFFFF,
                                         _lib.ptr(dqkv), _lib.ptr(dpl), pl.shape[0], _lib.ptr(dub[0]), _lib.ptr(dub[1]),
                                         _lib.ptr(dBand), _lib.ptr(QvHM), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(st, "ia_relpos_attention_flash_bwd")
    return dqkv, dpl, dub[0], dub[1]


_VT = {}


def relpos_attention(qkv_bf16, pos_proj_bf16, bias_u, bias_v, lens, B, T, H, dk, dropout_p=0.0, seed=0):
    """qkv [B*T, 3*H*dk] bf16, pos_proj [2T-1, H*dk] bf16 -> ctx [B*T, H*dk] bf16 (csrc/attention.hip)."""
    L = _lib.lib()
    dev = qkv_bf16.device
    n = L.ia_attn_vt_elems(B, T, H)
    key = (dev.index, n)
    vt = _VT.get(key)
    if vt is None:
        vt = _VT[key] = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ctx = torch.empty(B * T, H * dk, dtype=torch.bfloat16, device=dev)
    st = L.ia_relpos_attention(_lib.ptr(qkv_bf16), _lib.ptr(pos_proj_bf16), _lib.ptr(bias_u), _lib.ptr(bias_v),
                               _lib.ptr(lens), B, T, H, dk, float(dropout_p), int(seed) & 0xFFFFFFFF, _lib.ptr(vt),
                               _lib.ptr(ctx), _lib.stream_ptr())
    _lib.check(st, "ia_relpos_attention")
    return ctx


def attention_bwd_dims(T):
    import ctypes
    ts, rs, p0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().ia_relpos_attention_bwd_dims(int(T), ctypes.byref(ts), ctypes.byref(rs), ctypes.byref(p0)), "dims")
    return ts.value, rs.value, p0.value


def relpos_attention_bwd(qkv, pl, bias_u, bias_v, lens, ctx, dctx, B, T, H, dk, dropout_p=0.0, seed=0, dub_out=None):
    """Backward of relpos_attention.  qkv [B*T,3d], pl [>=2T-1, d], ctx/dctx [B*T, d] (bf16) ->
    (dqkv [B*T,3d] bf16, dpl [pl rows, d] bf16, dbias_u [H,dk] f32, dbias_v [H,dk] f32).
    csrc/attention.hip writes dropout(P), dS and the band-skewed dS; the five contractions are batched GEMMs on them."""
    L = _lib.lib()
    dev = qkv.device
    d = H * dk
    Ts, Rs, pad0 = attention_bwd_dims(T)
    bf = torch.bfloat16
    Pd = torch.empty(B, H, T, Ts, dtype=bf, device=dev)
    dS = torch.empty(B, H, T, Ts, dtype=bf, device=dev)
    dBand = torch.empty(H, B * T, Rs, dtype=bf, device=dev)
    Qu, K, dO = (torch.empty(B, H, T, dk, dtype=bf, device=dev) for _ in range(3))
    Qv = torch.empty(H, B * T, dk, dtype=bf, device=dev)
    dctx = dctx.contiguous()
    st = L.ia_relpos_attention_bwd(_lib.ptr(qkv), _lib.ptr(pl), _lib.ptr(bias_u), _lib.ptr(bias_v), _lib.ptr(lens), _lib.ptr(ctx),
                                   _lib.ptr(dctx), B, T, H, dk, float(dropout_p), int(seed) & 0xFFFFFFFF, _lib.ptr(Pd),
                                   _lib.ptr(dS), _lib.ptr(dBand), _lib.ptr(Qu), _lib.ptr(Qv), _lib.ptr(K), _lib.ptr(dO),
                                   _lib.stream_ptr())
    _lib.check(st, "ia_relpos_attention_bwd")
    Pv, dSv = (Pd, dS) if Ts == T else (Pd[..., :T], dS[..., :T])
    dV = torch.matmul(Pv.transpose(-1, -2), dO)                                                # [B,H,T,dk]
    dK = torch.matmul(dSv.transpose(-1, -2), Qu)
    dQu = torch.matmul(dSv, K)
    R = 2 * T - 1
    posB = torch.zeros(H, Rs, dk, dtype=bf, device=dev)
    posB[:, pad0:pad0 + R] = pl[:R].view(R, H, dk).permute(1, 0, 2)
    dQv = torch.bmm(dBand, posB)                                                               # [H,B*T,dk]
    dposB = torch.bmm(dBand.transpose(1, 2), Qv)                                               # [H,Rs,dk]
    dqkv = torch.empty(B * T, 3 * d, dtype=bf, device=dev)
    dub = dub_out if dub_out is not None else torch.empty(2, H, dk, dtype=torch.float32, device=dev)   # (dbias_u, dbias_v) destinations
    st = L.ia_attn_bwd_unpack(_lib.ptr(dQu), _lib.ptr(dQv), _lib.ptr(dK), _lib.ptr(dV), _lib.ptr(dqkv), _lib.ptr(dub[0]),
                              _lib.ptr(dub[1]), B, T, H, dk, _lib.ptr(scratch(dev, L.ia_attn_bwd_unpack_scratch_elems(B, T, H))),
                              _lib.stream_ptr())
    _lib.check(st, "ia_attn_bwd_unpack")
    dpl = torch.zeros_like(pl)
    dpl[:R] = dposB[:, pad0:pad0 + R].permute(1, 0, 2).reshape(R, d)
    return dqkv, dpl, dub[0], dub[1]


def colsum(x_bf16):
    M, N = x_bf16.shape
    out = torch.zeros(N, dtype=torch.float32, device=x_bf16.device)
    st = _lib.lib().ia_colsum_bf16(_lib.ptr(x_bf16), M, N, x_bf16.stride(0), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(st, "ia_colsum_bf16")
    return out


def gemm_tn(dy_bf16, x_bf16):
    """(dW [N,K] f32, db [N] f32) = (dY^T X, column sums of dY) on csrc/gemm_tn.hip; dY [M,N], X [M,K] bf16 with 16-byte
    aligned rows (row strides may exceed the widths: slices of a wider matrix are fine)."""
    L = _lib.lib()
    M, N = dy_bf16.shape
    K = x_bf16.shape[1]
    buf = torch.empty(N * K + N, dtype=torch.float32, device=dy_bf16.device)
    dW, db = buf[:N * K].view(N, K), buf[N * K:]
    st = L.ia_gemm_tn_bf16(_lib.ptr(dy_bf16), dy_bf16.stride(0), _lib.ptr(x_bf16), x_bf16.stride(0), M, N, K, _lib.ptr(dW),
                           _lib.ptr(db), _lib.ptr(scratch(dy_bf16.device, L.ia_gemm_tn_scratch_elems(M, N, K))),
                           _lib.stream_ptr())
    _lib.check(st, "ia_gemm_tn_bf16")
    return dW, db


def gemm_tn_grouped(pairs):
    """[(dY [M,N] bf16, X [M,K] bf16), ...] (<= 8) -> [(dW [N,K] f32, db [N] f32), ...] in ONE GEMM launch + one finishing
    launch (csrc/gemm_tn.hip, grouped form)."""
    import ctypes
    L = _lib.lib()
    n_ = len(pairs)
    arr = (_lib.TnProblem * n_)()
    outs = []
    dev = pairs[0][0].device
    for i, (dy, x) in enumerate(pairs):
        M, N = dy.shape
        K = x.shape[1]
        buf = torch.empty(N * K + N, dtype=torch.float32, device=dev)
        outs.append((buf[:N * K].view(N, K), buf[N * K:]))
        arr[i].dY, arr[i].X, arr[i].dW, arr[i].db = dy.data_ptr(), x.data_ptr(), buf.data_ptr(), buf.data_ptr() + 4 * N * K
        arr[i].ldy, arr[i].ldx, arr[i].M, arr[i].n, arr[i].k = dy.stride(0), x.stride(0), M, N, K
    scr = scratch(dev, L.ia_gemm_tn_grouped_scratch_elems(ctypes.addressof(arr), n_))
    st = L.ia_gemm_tn_bf16_grouped(ctypes.addressof(arr), n_, _lib.ptr(scr), _lib.stream_ptr())
    _lib.check(st, "ia_gemm_tn_bf16_grouped")
    return outs


def weight_t_shadow(weight):
    """W^T [K,N] bf16, cached per parameter version: the 'weight' of the data-gradient GEMM dX = dY W = dY (W^T)^T."""
    return _cached(("wT", id(weight)), (weight,), lambda: bf16_shadow(weight).t().contiguous())


def gemm_data_grad(dy_bf16, weight):
    """dX [M,K] bf16 = dY W for y = x W^T, on the HIP NT GEMM against the cached transposed shadow of `weight` [N,K]."""
    return gemm(dy_bf16, weight_t_shadow(weight))[1]


class _LinearHip(torch.autograd.Function):
    """y = x W^T + b for trainable projections outside the fused blocks (joint enc / pred): forward on the HIP GEMM (bias
    in the epilogue, bf16 out), data gradient as a library GEMM, weight + bias gradient by csrc/gemm_tn.hip (hipBLASLt's
    single TN GEMM launches 16-40 workgroups for these [256..1024 x 256..1024 x 12032] shapes: 70-80 us vs 17 us)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        if x.dtype == torch.bfloat16 and x.is_contiguous():
            xb = x.detach().view(-1, shp[-1])
        elif x.is_contiguous():
            from . import tail
            xb = tail.bf16_of(x)                   # shared with the other consumers of the same activation (CTC head)
        else:   # strided view (the prediction network's time-major output seen batch-major): ONE cast-and-gather launch
            xb = torch.empty(shp, dtype=torch.bfloat16, device=x.device)
            xb.copy_(x.detach())
            xb = xb.view(-1, shp[-1])
        wb = bf16_shadow(weight)
        _, y = gemm(xb, wb, bias.detach().float() if bias is not None else None)
        ctx.save_for_backward(xb, wb)
        ctx.weight = weight
        ctx.meta = (shp, x.dtype, weight.dtype, weight.shape, bias.dtype if bias is not None else None)
        return y.view(*shp[:-1], wb.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        shp, xdt, wdt, wshape, bdt = ctx.meta
        dyb = dy.reshape(-1, wb.shape[0]).to(torch.bfloat16).contiguous()
        M, N = dyb.shape
        K = xb.shape[1]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            if gemm_supported(N, K):   # dX = dY W on the HIP GEMM against the cached W^T
                dx = gemm_data_grad(dyb, ctx.weight).view(shp).to(xdt)
            else:
                dx = torch.mm(dyb, wb).view(shp).to(xdt)
        if ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2]):
            if N % 8 == 0 and K % 8 == 0:
                dWf, dbf = gemm_tn(dyb, xb)
            else:
                dWf = torch.mm(dyb.t(), xb, out_dtype=torch.float32)
                dbf = dyb.float().sum(0)
            if ctx.needs_input_grad[1]:
                dW = dWf.view(wshape).to(wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                db = dbf.to(bdt)
        return dx, dW, db


def linear(x, weight, bias=None):
    """Drop-in for F.linear inside the bf16 (autocast) region of the trainable blocks."""
    w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)
    if (x.is_cuda and torch.is_autocast_enabled() and gemm_supported(w2.shape[1], w2.shape[0])
            and (x.requires_grad or weight.requires_grad)):
        return _LinearHip.apply(x, weight, bias)
    return torch.nn.functional.linear(x, w2, bias)


def _cached(key, params, make):
    ver = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params) + \
        ((WEIGHT_EPOCH,) if any(p.requires_grad for p in params) else ())
    hit = _SHADOW.get(key)
    if hit is not None and hit[0] == ver and _same(hit[2], params):
        return hit[1]
    with torch.no_grad():
        val = make()
    _SHADOW[key] = (ver, val, _refs(_SHADOW, key, params))
    return val


def subsample_supported(C, d, feat_in):
    return (C % 8 == 0 and C <= 2048 and d % 8 == 0
            and (C * (((feat_in - 1) // 2 + 1 - 1) // 2 + 1)) % 8 == 0)


def conv_subsampling(feats_bft, conv1, conv2, lin, alpha=1.0, dropout_p=0.0, seed=0):
    """feats [B,feat_in,Tm] f32 -> [B*T2, d] f32 (ConvSubsampling.forward, subsampling.py:385-437) on the HIP path:
    direct conv1+ReLU (channels-last), implicit-GEMM conv2+ReLU on the matrix cores, Linear on the bf16 GEMM.
    alpha / dropout_p: RelPositionalEncoding's `dropout(x * xscale)` (multi_head_attention.py:968-979) in the Linear's epilogue."""
    L = _lib.lib()
    x = feats_bft.float().contiguous()
    B, Fm, Tm = x.shape
    C = conv1.weight.shape[0]
    T1, F1 = (Tm - 1) // 2 + 1, (Fm - 1) // 2 + 1
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    dev = x.device
    w1 = _cached(("sub_w1", id(conv1.weight)), [conv1.weight], lambda: conv1.weight.detach().float().reshape(C, 9).contiguous())
    o1 = torch.empty(B * T1 * F1, C, dtype=torch.bfloat16, device=dev)
    _lib.check(L.ia_subsample_conv1(_lib.ptr(x), B, Fm, Tm, C, _lib.ptr(w1), _lib.ptr(conv1.bias), _lib.ptr(o1),
                                    _lib.stream_ptr()), "ia_subsample_conv1")
    N = conv2.weight.shape[0]
    w2r = _cached(("sub_w2", id(conv2.weight)), [conv2.weight],
                  lambda: conv2.weight.detach().permute(0, 2, 3, 1).reshape(N, 9 * C).to(torch.bfloat16).contiguous())
    o2 = torch.empty(B * T2 * F2, N, dtype=torch.bfloat16, device=dev)
    _lib.check(L.ia_subsample_conv2(_lib.ptr(o1), B, T1, F1, C, _lib.ptr(w2r), _lib.ptr(conv2.bias), N, _lib.ptr(o2),
                                    _lib.stream_ptr()), "ia_subsample_conv2")
    d = lin.weight.shape[0]
    wl = _cached(("sub_wl", id(lin.weight)), [lin.weight],
                 lambda: lin.weight.detach().view(d, N, F2).permute(0, 2, 1).reshape(d, F2 * N).to(torch.bfloat16).contiguous())
    y = torch.empty(B * T2, d, dtype=torch.float32, device=dev)
    gemm(o2.view(B * T2, F2 * N), wl, lin.bias, out_f32=y, want_bf16=False, alpha=float(alpha), dropout_p=float(dropout_p), seed=seed)
    return y.view(B, T2, d)


# ---------------------------------------------------------------------------------------------------------------------
# Native executor of the no-autograd Conformer prefix (csrc/block_exec.hip): one C call for all frozen blocks.
_BLOCK_PARAMS = {}


def _block_params(layer):
    """ia_block_params of one ConformerLayer, cached until one of its tensors changes (version / storage / weight epoch)."""
    ff1, ff2, att, cv, bn = layer.feed_forward1, layer.feed_forward2, layer.self_attn, layer.conv, layer.conv.batch_norm
    watch = (ff1.linear1.weight, ff1.linear2.weight, att.linear_q.weight, att.linear_k.weight, att.linear_v.weight,
             att.linear_pos.weight, att.linear_out.weight, cv.pointwise_conv1.weight, cv.pointwise_conv2.weight,
             ff2.linear1.weight, ff2.linear2.weight, att.linear_q.bias, att.linear_k.bias, att.linear_v.bias,
             cv.depthwise_conv.weight, cv.pointwise_conv1.bias)
    ver = tuple(p._version for p in watch) + tuple(p.data_ptr() for p in watch) + \
        ((WEIGHT_EPOCH,) if any(p.requires_grad for p in watch) else ()) + \
        (layer.dropout.p, ff1.dropout.p, ff2.dropout.p, att.dropout_rate)
    hit = _BLOCK_PARAMS.get(id(layer))
    if hit is not None and hit[0] == ver and _same(hit[3], watch):
        return hit[1]
    d = ff1.linear1.weight.shape[1]
    keep = dict(
        w_ff1a=bf16_shadow(ff1.linear1.weight), w_ff1b=bf16_shadow(ff1.linear2.weight),
        w_qkv=bf16_shadow(att.linear_q.weight, att.linear_k.weight, att.linear_v.weight), w_pos=bf16_shadow(att.linear_pos.weight),
        w_out=bf16_shadow(att.linear_out.weight), w_pw1=bf16_shadow(cv.pointwise_conv1.weight),
        w_pw2=bf16_shadow(cv.pointwise_conv2.weight), w_ff2a=bf16_shadow(ff2.linear1.weight), w_ff2b=bf16_shadow(ff2.linear2.weight),
        b_ff1a=ff1.linear1.bias, b_ff1b=ff1.linear2.bias, b_qkv=f32_cat(att.linear_q.bias, att.linear_k.bias, att.linear_v.bias),
        b_out=att.linear_out.bias, b_pw1=cv.pointwise_conv1.bias, b_pw2=cv.pointwise_conv2.bias, b_ff2a=ff2.linear1.bias,
        b_ff2b=ff2.linear2.bias,
        ln_ff1_g=layer.norm_feed_forward1.weight, ln_ff1_b=layer.norm_feed_forward1.bias, ln_att_g=layer.norm_self_att.weight,
        ln_att_b=layer.norm_self_att.bias, ln_conv_g=layer.norm_conv.weight, ln_conv_b=layer.norm_conv.bias,
        ln_ff2_g=layer.norm_feed_forward2.weight, ln_ff2_b=layer.norm_feed_forward2.bias, ln_out_g=layer.norm_out.weight,
        ln_out_b=layer.norm_out.bias, pos_u=att.pos_bias_u, pos_v=att.pos_bias_v, dw_w=cv.depthwise_conv.weight,
        dw_b=cv.depthwise_conv.bias, bn_g=bn.weight, bn_b=bn.bias, bn_rm=bn.running_mean, bn_rv=bn.running_var,
        bn_nbt=bn.num_batches_tracked)
    # (frozen weights only: the regrouped image costs half a dozen small launches to rebuild, and the trainable blocks' own
    #  executor keeps the pre-GLU tensor for its backward anyway)
    frozen_pw1 = not (cv.pointwise_conv1.weight.requires_grad or cv.pointwise_conv1.bias.requires_grad)
    glu = glu_regrouped(cv.pointwise_conv1.weight, cv.pointwise_conv1.bias) if frozen_pw1 else None
    if glu is not None:
        keep["w_pw1_glu"], keep["b_pw1_glu"] = glu
    bp = _lib.BlockParams()
    for k, t in keep.items():
        setattr(bp, k, t.data_ptr())
    bp.ln_eps, bp.bn_eps = float(layer.norm_out.eps), float(bn.eps)
    bp.bn_momentum = float(bn.momentum if bn.momentum is not None else 0.1)
    bp.p_drop, bp.p_ff, bp.p_att = float(layer.dropout.p), float(ff1.dropout.p), float(att.dropout_rate)
    bp.fc_factor = float(layer.fc_factor)
    bp.d, bp.d_ff, bp.n_heads, bp.ksz = d, ff1.linear1.weight.shape[0], att.h, cv.depthwise_conv.weight.shape[-1]
    _BLOCK_PARAMS[id(layer)] = (ver, bp, keep, _refs(_BLOCK_PARAMS, id(layer), watch))
    return bp


_PREFIX_WS = {}
_POS_PROJ = {}   # id(layer) -> (weight version key, {pos rows: (pos_emb data_ptr, pl)}, weakrefs)


def pos_proj_cached(layer, pos_emb_bf16, max_entries=8):
    """linear_pos(pos_emb) of a layer whose position projection is frozen: pos_emb is a pure function of T (a centred
    slice of the sin/cos table), so the projection is computed once per (layer, T) and reused by every later step.
    None for trainable weights (recomputed per call by the executor)."""
    w = layer.self_attn.linear_pos.weight
    if w.requires_grad:
        return None
    ver = (w._version, w.data_ptr())
    hit = _POS_PROJ.get(id(layer))
    if hit is None or hit[0] != ver or not _same(hit[2], (w,)):
        hit = _POS_PROJ[id(layer)] = (ver, {}, _refs(_POS_PROJ, id(layer), (w,)))
    rows = pos_emb_bf16.shape[0]
    ent = hit[1].get(rows)
    if ent is None:
        if len(hit[1]) >= max_entries:
            hit[1].pop(next(iter(hit[1])))
        with torch.no_grad():
            _, pl = gemm(pos_emb_bf16, bf16_shadow(w))
        ent = hit[1][rows] = pl
    return ent


def conformer_prefix(layers, xr, pos_emb_bf16, lens, B, T, seed_base, seed_stride, training):
    """Run `layers` (ConformerLayer list) over the fp32 residual stream xr [B*T, d] in place with ONE native call."""
    L = _lib.lib()
    arr = (_lib.BlockParams * len(layers))(*[_block_params(l) for l in layers])
    for i, l in enumerate(layers):   # frozen position projections depend on T only: one GEMM per (layer, T), not per step
        pl = pos_proj_cached(l, pos_emb_bf16)
        arr[i].pl_cached = pl.data_ptr() if pl is not None else None
    p0 = arr[0]
    n = L.ia_conformer_prefix_ws_bytes(B, T, p0.d, p0.d_ff, p0.n_heads, p0.ksz, pos_emb_bf16.shape[0])
    key = (xr.device.index, torch.cuda.current_stream(xr.device).cuda_stream)
    ws = _PREFIX_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _PREFIX_WS[key] = torch.empty(n, dtype=torch.uint8, device=xr.device)
    import ctypes
    groups = [bn_sync_group(l.conv.batch_norm) for l in layers]
    if all(g is None for g in groups):
        st = L.ia_conformer_prefix_fwd(ctypes.addressof(arr), len(layers), _lib.ptr(xr), _lib.ptr(pos_emb_bf16), pos_emb_bf16.shape[0],
                                       _lib.ptr(lens), B, T, int(seed_base) & 0xFFFFFFFF, int(seed_stride), int(bool(training)),
                                       _lib.ptr(ws), n, _lib.stream_ptr())
        _lib.check(st, "ia_conformer_prefix_fwd")
        return xr
    # SyncBatchNorm over several ranks: n+1 native segments with the all-reduce of each block's [sum | sumsq | count] in between
    d = p0.d
    off = L.ia_conformer_prefix_ws_sums_offset(B, T, p0.d, p0.d_ff, p0.n_heads, p0.ksz, pos_emb_bf16.shape[0])
    sums = ws[off:off + (2 * d + 1) * 4].view(torch.float32)
    nl = len(layers)
    for k in range(nl + 1):
        a, b = max(0, 2 * k - 1), min(2 * nl, 2 * k + 1)
        st = L.ia_conformer_prefix_fwd_seg(ctypes.addressof(arr), nl, _lib.ptr(xr), _lib.ptr(pos_emb_bf16), pos_emb_bf16.shape[0],
                                           _lib.ptr(lens), B, T, int(seed_base) & 0xFFFFFFFF, int(seed_stride), int(bool(training)),
                                           a, b, 1, _lib.ptr(ws), n, _lib.stream_ptr())
        _lib.check(st, "ia_conformer_prefix_fwd_seg")
        if k < nl:
            g = groups[k]
            if g is None:
                raise RuntimeError("conformer_prefix: mixed BatchNorm / SyncBatchNorm layers in one prefix")
            bn_sync_sums(sums, B * T, d, layers[k].conv.batch_norm, g)
    return xr
