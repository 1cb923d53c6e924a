"""Step-level parity AT THE BENCHMARKED DIMENSIONS AND DTYPE (BASELINE.json configs) against the fp32 CPU oracle
(oracle/step_ref.py) on identical weights and inputs.

configs[1]  Conformer-medium (d=256, 16 L, H=640, 22 x 257 heads), EWC recipe: layers <= 12 frozen, bf16 projections,
            f16 fused joint -- batch 4 x 15 s here (the oracle finishes in seconds), every trainable tensor's gradient.
configs[2]  the MAS importance pass (stashed logits, R/cl_baseline_mas.py:258-265) at the same dimensions.
configs[3]  the LwF teacher + student step (R/cl_baseline_lwf.py:212-264) at the same dimensions.
configs[0]  Conformer-small (d=144, 4 heads of 36), batch 2 x 5 s.
configs[4]  Conformer-large (d=512, 18 L, 8 heads), 30 s utterances (T' = 751), batch 2.

Tolerances.  Losses: the north_star's 1e-3 relative holds at the benchmarked dtype (observed 1e-5 .. 5e-5: the loss is
a sum over ~10^5 lattice cells, rounding noise averages out).  Gradients: the benchmarked dtype rounds every projection
operand to bf16 (2^-9 relative per operand) and the joint's logits to f16; through 16 layers individual gradient tensors
differ from the fp32 oracle by up to 2 % relative L2 (median 0.4 %).  The asserted bounds are the tightest that hold with
margin on the seeds below; the measured values are printed (pytest -s) and recorded in DESIGN.md section 5.
"""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import step_ref as S


def _synth(B, seconds, seed=7, vocab=256, sr=16000):
    """SURVEY 8(d) synthetic batch (same recipe as bench.py): 0.1*N(0,1) through a 2-tap low-pass, ragged lengths with
    one full-length item, 7 tokens / s."""
    from indic_cl_asr_amd.encoder import subsampled_length
    from indic_cl_asr_amd.features import mel_frame_count
    g = torch.Generator().manual_seed(seed)
    L = int(seconds * sr)
    lens = torch.round(L * (0.6 + 0.4 * torch.rand(B, generator=g))).long()
    lens[0] = L
    x = 0.1 * torch.randn(B, L, generator=g)
    x[:, 1:] = 0.5 * (x[:, 1:] + x[:, :-1])
    for b in range(B):
        x[b, lens[b]:] = 0
    tl = []
    for b in range(B):
        tp = subsampled_length(mel_frame_count(int(lens[b])))
        tl.append(max(1, min(int(round(7 * float(lens[b]) / sr)), tp - 1)))
    tl = torch.tensor(tl)
    tok = torch.randint(0, vocab, (B, int(tl.max())), generator=g)
    return x, lens, tok, tl


def _pair(preset, seed=0, freeze=None, **kw):
    from indic_cl_asr_amd.config import PRESETS, model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(seed)
    p = dict(PRESETS[preset]); p.update(kw)
    o = S.OracleHybridModel(**p)
    with torch.no_grad():  # make the zero-initialised pieces non-trivial
        for l in o.encoder.layers:
            l.self_attn.pos_bias_u.normal_(0, 0.1); l.self_attn.pos_bias_v.normal_(0, 0.1)
            l.conv.batch_norm.weight.uniform_(0.7, 1.3); l.conv.batch_norm.bias.normal_(0, 0.1)
    cfg = model_config(preset, compute_dtype='bf16', dither=0.0, **kw)
    m = EncDecHybridRNNTCTCModel(cfg)
    m.load_state_dict(o.state_dict())
    m = m.disable_dropout().cuda()
    m.spec_augment_enabled = False
    if freeze is not None:
        S.freeze_layer(o, freeze)
        freeze_layer(m, freeze); m.encoder.encoder_frozen_till = freeze
    return o, m


def _rel_l2(a, b):
    a, b = a.detach().float().cpu().flatten().double(), b.detach().float().cpu().flatten().double()
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


def _structural_zero(name):  # see tests/test_step_gpu.py
    return name.endswith("depthwise_conv.bias") or name.endswith("self_attn.linear_k.bias")


def _grad_table(m, o, min_checked):
    og = dict(o.named_parameters())
    rows = []
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        g_ref = og[n].grad
        if g_ref is None:  # e.g. the other languages' heads: no gradient on either side
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        if _structural_zero(n):
            continue
        rows.append((_rel_l2(p.grad, g_ref), n))
    rows.sort(reverse=True)
    assert len(rows) >= min_checked, len(rows)
    return rows


def _check_losses(mp, mo, tol):
    errs = {k: abs(mp[k] - mo[k]) / abs(mo[k]) for k in ('train_rnnt_loss', 'train_ctc_loss', 'train_loss')}
    print("loss rel err:", {k: f"{v:.2e}" for k, v in errs.items()}, "oracle", {k: round(mo[k], 4) for k in errs})
    for k, v in errs.items():
        assert v <= tol, (k, mp[k], mo[k], v)
    return errs


# ------------------------------------------------------------------------------------------------ configs[1]
def test_config2_medium_bf16_step_matches_oracle():
    o, m = _pair('medium', freeze=12)
    batch = _synth(4, 15.0)
    o.train(); m.train()
    lo, mo = o.training_step(batch, ['hi'] * 4)
    lo.backward()
    cb = tuple(t.cuda() for t in batch)
    lp, mp = m.training_step(cb, ['hi'] * 4)
    lp.backward()
    torch.cuda.synchronize()
    _check_losses(mp, mo, 1e-3)     # the north_star's bound, at the benchmarked dtype (observed 1e-5)
    rows = _grad_table(m, o, min_checked=3 * 30 + 8)
    print("worst gradient tensors (relative L2 vs fp32 oracle):")
    for e, n in rows[:8]:
        print(f"  {e:.3e}  {n}")
    print("median", f"{rows[len(rows) // 2][0]:.3e}")
    assert rows[0][0] <= 0.03, rows[0]               # observed 2.0e-2 (linear_pos of the first trainable layer) + 50 %
    assert rows[len(rows) // 2][0] <= 0.006          # observed 3.9e-3 + 50 %
    # encoder frame counts bit-exact, BatchNorm running stats of a frozen layer updated alike (reference quirk)
    bo, bp = o.encoder.layers[3].conv.batch_norm, m.encoder.layers[3].conv.batch_norm
    assert torch.allclose(bp.running_var.cpu(), bo.running_var, rtol=3e-2, atol=1e-4)
    assert int(bp.num_batches_tracked) == int(bo.num_batches_tracked) == 1


def test_config2_full_benchmarked_batch_32x15s_matches_oracle():
    """BASELINE configs[1] AT ITS FULL SIZE: the batch bench.py times (32 utterances x 15 s from bench.synth_batch's recipe,
    Conformer-medium, layers <= 12 frozen, bf16 projections, f16 fused joint, 8 sub-batches of 4) against the fp32 oracle on the
    same weights -- losses and every trainable tensor's gradient (the oracle needs about a minute of the box's host cores)."""
    o, m = _pair('medium', freeze=12)
    batch = _synth(32, 15.0, seed=1234)
    o.train(); m.train()
    lo, mo = o.training_step(batch, ['hi'] * 32)
    lo.backward()
    cb = tuple(t.cuda() for t in batch)
    lp, mp = m.training_step(cb, ['hi'] * 32)
    lp.backward()
    torch.cuda.synchronize()
    errs = _check_losses(mp, mo, 1e-3)            # north_star bound at the benchmarked size and dtype
    assert max(errs.values()) <= 3e-4, errs       # (the 4-utterance test observes 1e-5 .. 5e-5)
    rows = _grad_table(m, o, min_checked=3 * 30 + 8)
    print("worst gradient tensors (relative L2 vs fp32 oracle), 32 x 15 s:")
    for e, n in rows[:6]:
        print(f"  {e:.3e}  {n}")
    print("median", f"{rows[len(rows) // 2][0]:.3e}")
    assert rows[0][0] <= 0.023, rows[0]              # observed 1.5e-2 (more utterances average the rounding noise) + 50 %
    assert rows[len(rows) // 2][0] <= 0.0055         # observed 3.6e-3 + 50 %


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config3_mas_importance_pass_medium_dims():
    from indic_cl_asr_amd import cl
    o, m = _pair('medium', freeze=12)
    flat = cl.FlatParams(m)
    batch = _synth(4, 15.0, seed=11)
    for mod in (o, m):
        mod.train()
        mod.joint.store_sub_logits = True; mod.ctc_decoder.return_logits_ = True
    lo, mo = o.training_step(batch, ['hi'] * 4)
    imp_o = S.mas_importance_loss(o.joint.store_list, o.ctc_decoder.decoder_logits, 0.3)
    imp_o.backward()
    lp, mp = m.training_step(tuple(t.cuda() for t in batch), ['hi'] * 4)
    flat.zero_grad()
    imp_p = cl.mas_importance_loss(m, 0.3)
    imp_p.backward()
    torch.cuda.synchronize()
    _check_losses(mp, mo, 1e-3)
    rel = abs(imp_p.item() - imp_o.item()) / abs(imp_o.item())
    print("importance loss rel err", f"{rel:.2e}")
    assert rel <= 2e-3                                # observed 3.4e-4
    om = cl.get_zero_params(m)
    cl.importance_accumulate(flat, om)
    og = dict(o.named_parameters())
    worst = 0.0
    n_checked = 0
    for n in flat.names:
        if og[n].grad is None or _structural_zero(n):
            continue
        e = _rel_l2(om[n], og[n].grad.abs())
        worst = max(worst, e); n_checked += 1
        assert e <= 0.04, (n, e)                      # observed worst 1.4e-2
    print("omega worst rel L2", f"{worst:.3e}", "tensors", n_checked)
    assert n_checked >= 90


# ------------------------------------------------------------------------------------------------ configs[3]
def test_config4_lwf_teacher_student_medium_dims():
    from indic_cl_asr_amd import cl
    o, m = _pair('medium', freeze=12)
    flat = cl.FlatParams(m)
    batch = _synth(4, 15.0, seed=13)
    cb = tuple(t.cuda() for t in batch)
    m.train(); o.train()
    teacher = cl.get_params_clone(m)
    o_teacher = copy.deepcopy(o)
    with torch.no_grad():   # the student has moved away from the teacher
        flat.theta.add_(0.01 * torch.randn_like(flat.theta) * flat.theta.abs().mean())
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        o_teacher.joint.store_sub_enc = True; o_teacher.joint.detach_sub_enc = True
        _, _, prob_ = o_teacher.training_step(batch, ['hi'] * 4, return_probs=True)
        store = o_teacher.joint.store_list
    o.joint.store_sub_enc = True; o.joint.detach_sub_enc = False
    lo, mo, prob = o.training_step(batch, ['hi'] * 4, return_probs=True)
    tot_o, rn_o, ct_o = S.lwf_kd_loss(lo, prob, prob_, o.joint.store_list, store, 0.1, 0.3)
    tot_o.backward()
    p_prob_, p_store = cl.lwf_teacher_forward(m, flat, teacher, cb, ['hi'] * 4)
    m.joint.store_sub_enc = True; m.joint.detach_sub_enc = False
    lp, mp, p_prob = m.training_step(cb, ['hi'] * 4, return_probs=True)
    tot_p, rn_p, ct_p = cl.lwf_kd_loss(lp, p_prob, p_prob_, m.joint.store_list, p_store, 0.1, 0.3)
    flat.zero_grad()
    tot_p.backward()
    torch.cuda.synchronize()
    _check_losses(mp, mo, 5e-3)
    print("kd: rnnt", rn_p.item(), rn_o.item(), "ctc", ct_p.item(), ct_o.item(), "total", tot_p.item(), tot_o.item())
    # the KD terms are differences of two nearly equal distributions: compare them on the scale of the loss they enter
    scale = abs(tot_o.item())
    assert abs(rn_p.item() - rn_o.item()) <= 2e-3 * scale + 0.05 * abs(rn_o.item())
    assert abs(ct_p.item() - ct_o.item()) <= 2e-3 * scale + 0.05 * abs(ct_o.item())
    assert math.isclose(tot_p.item(), tot_o.item(), rel_tol=5e-3)
    rows = _grad_table(m, o, min_checked=90)
    print("worst", [(f"{e:.3e}", n) for e, n in rows[:5]], "median", f"{rows[len(rows) // 2][0]:.3e}")
    # observed worst 3.0e-2 (pos_bias_v / linear_pos of the first trainable layer: the KD terms add a second, f16-scaled lattice
    # gradient); worst bound = observed + 50 %
    assert rows[0][0] <= 0.045, rows[0]
    assert rows[len(rows) // 2][0] <= 0.02


# ------------------------------------------------------------------------------------------------ configs[0]
def test_config1_small_bf16_step_matches_oracle():
    o, m = _pair('small', freeze=None)
    batch = _synth(2, 5.0, seed=3)
    o.train(); m.train()
    S.freeze_layer(o, 12)
    from indic_cl_asr_amd.model import freeze_layer
    freeze_layer(m, 12); m.encoder.encoder_frozen_till = 12
    lo, mo = o.training_step(batch, ['hi'] * 2)
    lo.backward()
    lp, mp = m.training_step(tuple(t.cuda() for t in batch), ['hi'] * 2)
    lp.backward()
    torch.cuda.synchronize()
    _check_losses(mp, mo, 5e-3)
    rows = _grad_table(m, o, min_checked=90)
    print("worst", [(f"{e:.3e}", n) for e, n in rows[:5]], "median", f"{rows[len(rows) // 2][0]:.3e}")
    # d = 144: k-tiles straddle the 36-wide heads and the bf16 rounding is relatively larger on 5 s utterances (few frames to
    # average over): observed worst 3.5e-2 (norm_conv / depthwise of layer 14), median 6.3e-3; bounds = observed + 50 %
    assert rows[0][0] <= 0.053, rows[0]
    assert rows[len(rows) // 2][0] <= 0.0095


# ------------------------------------------------------------------------------------------------ configs[4]
def test_config5_large_30s_bf16_step_matches_oracle():
    o, m = _pair('large', freeze=None)
    batch = _synth(2, 30.0, seed=5)
    o.train(); m.train()
    S.freeze_layer(o, 14)
    from indic_cl_asr_amd.model import freeze_layer
    freeze_layer(m, 14); m.encoder.encoder_frozen_till = 14
    lo, mo = o.training_step(batch, ['hi'] * 2)
    lo.backward()
    lp, mp = m.training_step(tuple(t.cuda() for t in batch), ['hi'] * 2)
    lp.backward()
    torch.cuda.synchronize()
    _check_losses(mp, mo, 5e-3)
    rows = _grad_table(m, o, min_checked=90)
    print("worst", [(f"{e:.3e}", n) for e, n in rows[:5]], "median", f"{rows[len(rows) // 2][0]:.3e}")
    assert rows[0][0] <= 0.029, rows[0]              # observed 1.9e-2 (linear_pos of layer 16) + 50 %
    assert rows[len(rows) // 2][0] <= 0.007          # observed 4.7e-3 + 50 %
