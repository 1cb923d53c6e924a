"""csrc/ffn_fused.hip (LayerNorm -> W1 -> SiLU -> dropout -> W2 -> dropout -> residual [-> LayerNorm] in one launch)
against (a) the fp64 restatement of the same bf16-quantised computation and (b) the unfused HIP sequence it replaces
(ia_layernorm + two ia_gemm_bf16 launches), which shares its dropout masks bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _modules(d=256, dff=1024, seed=0):
    torch.manual_seed(seed)
    ln = torch.nn.LayerNorm(d).cuda()
    l1, l2 = torch.nn.Linear(d, dff).cuda(), torch.nn.Linear(dff, d).cuda()
    ln2 = torch.nn.LayerNorm(d).cuda()
    with torch.no_grad():
        for m in (ln, ln2):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    return ln, l1, l2, ln2


def _reference(x, ln, l1, l2, alpha, ln2):
    """fp64 math on the operands the kernel sees: bf16 LN output, bf16 weights, bf16 intermediate."""
    xd = x.double()
    y = torch.nn.functional.layer_norm(xd, (x.shape[1],), ln.weight.double(), ln.bias.double(), ln.eps).bfloat16().double()
    h = y @ l1.weight.detach().bfloat16().double().t() + l1.bias.double()
    h = (h * torch.sigmoid(h)).bfloat16().double()
    o = h @ l2.weight.detach().bfloat16().double().t() + l2.bias.double()
    r = xd + alpha * o
    if ln2 is not None:
        r = torch.nn.functional.layer_norm(r, (x.shape[1],), ln2.weight.double(), ln2.bias.double(), ln2.eps)
    return r


@pytest.mark.parametrize("N,with_ln2", [(64, False), (12032, True), (1000, True), (77, False)])
def test_ffn_fused_matches_fp64_restatement(N, with_ln2):
    from indic_cl_asr_amd.ops import fast
    ln, l1, l2, ln2 = _modules()
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(N, 256, generator=g) * 1.5).cuda()
    ref = _reference(x, ln, l1, l2, 0.5, ln2 if with_ln2 else None)
    y16 = torch.empty(N, 256, dtype=torch.bfloat16, device="cuda")
    out = fast.ffn_fused(x.clone(), ln, l1, l2, 0.5, ln2=ln2 if with_ln2 else None, y_out=y16)
    err = (out.double() - ref).abs().max().item()
    assert err <= 3e-3 * ref.abs().max().item() + 1e-4, err   # fp32 accumulation of bf16 products, bf16 rounding ties
    assert torch.equal(y16, out.bfloat16())


@pytest.mark.parametrize("p_res", [0.0, 0.1, 0.25])
def test_ffn_fused_matches_unfused_sequence_with_identical_output_dropout_mask(p_res):
    """The module-output dropout uses ia_gemm_bf16's mask for the same (seed, row, column): with it on (and the inner
    dropout off) the fused launch and the LayerNorm + two-GEMM sequence agree element by element."""
    from indic_cl_asr_amd.ops import fast
    ln, l1, l2, ln2 = _modules(seed=1)
    N = 4000
    x = (torch.randn(N, 256, generator=torch.Generator().manual_seed(5)) * 1.2).cuda()
    xa = x.clone()
    y = fast.layernorm(xa, ln.weight, ln.bias, ln.eps)
    _, h = fast.gemm(y, fast.bf16_shadow(l1.weight), l1.bias, act=1)
    fast.gemm(h, fast.bf16_shadow(l2.weight), l2.bias, dropout_p=p_res, seed=12, alpha=0.5, residual=xa, out_f32=xa, want_bf16=False)
    fast.layernorm(xa, ln2.weight, ln2.bias, ln2.eps, out_f32=xa, want_bf16=False)
    xb = fast.ffn_fused(x.clone(), ln, l1, l2, 0.5, 0.0, 11, p_res, 12, ln2=ln2)
    err = (xa - xb).abs().max().item()
    assert err <= 4e-3 * xa.abs().max().item(), err


def test_ffn_fused_inner_dropout_is_unbiased_and_deterministic():
    """The dropout behind the activation draws the kernel's own cheap mask (one word per frame and 4 hidden units):
    deterministic in the seed, different across seeds, and -- inverted dropout -- unbiased: the mean over many seeds
    returns to the undropped result."""
    from indic_cl_asr_amd.ops import fast
    ln, l1, l2, _ = _modules(seed=2)
    N = 1024
    x = (torch.randn(N, 256, generator=torch.Generator().manual_seed(6)) * 1.2).cuda()
    base = fast.ffn_fused(x.clone(), ln, l1, l2, 1.0) - x
    d1 = fast.ffn_fused(x.clone(), ln, l1, l2, 1.0, 0.25, 21) - x
    d2 = fast.ffn_fused(x.clone(), ln, l1, l2, 1.0, 0.25, 21) - x
    d3 = fast.ffn_fused(x.clone(), ln, l1, l2, 1.0, 0.25, 22) - x
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    single = ((d1 - base).norm() / base.norm()).item()
    assert single > 0.05
    acc = torch.zeros_like(base)
    n = 32
    for s in range(n):
        acc += fast.ffn_fused(x.clone(), ln, l1, l2, 1.0, 0.25, 100 + s) - x
    rel = ((acc / n - base).norm() / base.norm()).item()
    assert rel < 0.3 * single, (rel, single)       # ~ 1/sqrt(32) of one draw's deviation
    # keep rate: with W2 = I-like probe the fraction of zeroed hidden units is p; cheap proxy: E|d1| / E|base| ~ 1 within noise
    assert 0.8 < (d1.abs().mean() / base.abs().mean()).item() < 1.3


def test_ffn_fused_can_emit_the_next_modules_layernorm_without_touching_the_residual():
    """ln2_to_y_only: x receives the un-normalised residual, y_out = LN2(x) in bf16 (norm_self_att behind the first
    feed-forward module of a block: saves that LayerNorm launch)."""
    from indic_cl_asr_amd.ops import fast
    ln, l1, l2, ln2 = _modules(seed=3)
    N = 777
    x = (torch.randn(N, 256, generator=torch.Generator().manual_seed(8)) * 1.1).cuda()
    ref_x = fast.ffn_fused(x.clone(), ln, l1, l2, 0.5)
    ref_y = fast.layernorm(ref_x, ln2.weight, ln2.bias, ln2.eps)
    y = torch.empty(N, 256, dtype=torch.bfloat16, device="cuda")
    out_x = fast.ffn_fused(x.clone(), ln, l1, l2, 0.5, ln2=ln2, y_out=y, ln2_to_y_only=True)
    assert torch.equal(out_x, ref_x)
    assert (y.float() - ref_y.float()).abs().max().item() <= 2e-2 * ref_y.float().abs().max().item()


@pytest.mark.parametrize("N,p_ff,p_res", [(12032, 0.0, 0.0), (1000, 0.1, 0.1), (77, 0.0, 0.25), (64, 0.0, 0.0)])
def test_ffn_fused_tail_projection_equals_the_projection_as_a_launch_of_its_own(N, p_ff, p_res):
    """ia_ffn_fused_tail: the q|k|v projection of the LN2 rows inside the feed-forward launch (rows from the epilogue into LDS, W_qkv
    through the ring) against ia_ffn_fused + ia_gemm_bf16 on the rows it wrote out: same bf16 operands, fp32 accumulation in a
    different k order -- equal up to one bf16 rounding of the output; the residual stream itself is bit-identical."""
    from indic_cl_asr_amd import _lib
    from indic_cl_asr_amd.ops import fast
    L = _lib.lib()
    ln, l1, l2, ln2 = _modules(seed=3)
    torch.manual_seed(9)
    qkv = torch.nn.Linear(256, 768).cuda()
    x = (torch.randn(N, 256, generator=torch.Generator().manual_seed(N)) * 1.3).cuda()
    w1, w2, wq = fast.bf16_shadow(l1.weight), fast.bf16_shadow(l2.weight), fast.bf16_shadow(qkv.weight)
    thr = lambda p_: int(p_ * 256 + 0.5) / 256.0

    def run(tail):
        xa = x.clone()
        y = torch.full((N, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
        out = torch.full((N, 768), float("nan"), dtype=torch.bfloat16, device="cuda")
        common = (_lib.ptr(xa), N, 256, 1024, _lib.ptr(ln.weight), _lib.ptr(ln.bias), ln.eps, _lib.ptr(w1), _lib.ptr(l1.bias), _lib.ptr(w2),
                  _lib.ptr(l2.bias), 0.5, thr(p_ff), 31, thr(p_res), 32, _lib.ptr(ln2.weight), _lib.ptr(ln2.bias))
        if tail:
            _lib.check(L.ia_ffn_fused_tail(*common, None, 1, _lib.ptr(wq), _lib.ptr(qkv.bias), _lib.ptr(out), 768, _lib.stream_ptr()), "ia_ffn_fused_tail")
        else:
            _lib.check(L.ia_ffn_fused(*common, _lib.ptr(y), 1, _lib.stream_ptr()), "ia_ffn_fused")
            _lib.check(L.ia_gemm_bf16(_lib.ptr(y), 256, _lib.ptr(wq), 256, N, 768, 256, _lib.ptr(qkv.bias), 0, 0.0, 0, 1.0, None, 0, None, 0,
                                      _lib.ptr(out), 768, _lib.stream_ptr()), "ia_gemm_bf16")
        torch.cuda.synchronize()
        return xa, out
    assert int(L.ia_ffn_fused_tail_supported(256, 1024, 768)) == 1 and int(L.ia_ffn_fused_tail_supported(256, 1024, 100)) == 0
    xa, oa = run(False)
    xb, ob = run(True)
    assert torch.equal(xa, xb)                                  # the residual stream does not know about the tail
    assert torch.isfinite(ob.float()).all()
    d = (oa.float() - ob.float()).abs()
    tol = 2.0 ** -7 * oa.float().abs().clamp_min(1e-2)          # one bf16 step (8 bits of mantissa) of the value
    assert (d <= tol).all(), (d.max().item(), (d > tol).sum().item())
    assert (d == 0).float().mean().item() > 0.97                # ... and a rounding tie is rare
