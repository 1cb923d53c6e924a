"""Checkpoint interchange: trainable-only .pth (R/utils.py:265-271), .nemo archive round trip
(model_config.yaml + model_weights.ckpt), and the persisted continual-learning state."""
import pytest
import torch

from indic_cl_asr_amd import checkpoint as ck
from indic_cl_asr_amd import cl
from indic_cl_asr_amd.config import model_config
from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer


def _model(seed):
    torch.manual_seed(seed)
    return EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32'))


def test_trainable_only_state_dict_round_trip(tmp_path):
    a, b = _model(0), _model(1)
    freeze_layer(a, 0); freeze_layer(b, 0)
    p = tmp_path / "task0.pth"
    ck.save_trainable(a, p)
    saved = torch.load(p)
    names = {n for n, q in a.named_parameters() if q.requires_grad}
    assert set(saved) == names and not any(n.startswith("encoder.layers.0.") for n in saved)
    rep = ck.load_weights(b, p)
    assert not rep.unexpected_keys and all(not k.startswith("decoder.") for k in rep.missing_keys)
    for n, q in b.named_parameters():
        if n in names:
            assert torch.equal(q, dict(a.named_parameters())[n])


def test_nemo_archive_round_trip(tmp_path):
    a = _model(2)
    p = tmp_path / "model.nemo"
    ck.write_nemo(a, p)
    cfg, state = ck.read_nemo(p)
    assert cfg["encoder"]["d_model"] == a.cfg.d_model and cfg["decoder"]["prednet"]["pred_hidden"] == a.cfg.pred_hidden
    b, rep = ck.model_from_nemo(p, strict=True, languages=a.cfg.languages, vocab_per_lang=a.cfg.vocab_per_lang,
                                compute_dtype='fp32')
    assert b.cfg.n_layers == a.cfg.n_layers and b.cfg.n_window_size == a.cfg.n_window_size
    sa, sb = a.state_dict(), b.state_dict()
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_cl_state_persists_and_checks_layout(tmp_path):
    m = _model(3)
    freeze_layer(m, 0)
    flat = cl.FlatParams(m)
    fisher, star = flat.zeros(), flat.clone_theta()
    fisher.flat.uniform_(0, 1)
    p = tmp_path / "cl_state.pt"
    ck.save_cl_state(p, fisher=fisher, checkpoint=star, importance=None)
    got = ck.load_cl_state(p, flat)
    assert set(got) == {"fisher", "checkpoint"}
    assert torch.equal(got["fisher"].flat, fisher.flat) and torch.equal(got["checkpoint"]["joint.enc.weight"], star["joint.enc.weight"])
    other = _model(3)
    freeze_layer(other, 1)                         # a different set of trainable tensors
    try:
        ck.load_cl_state(p, cl.FlatParams(other))
        assert False, "layout mismatch must be refused"
    except ValueError:
        pass


@pytest.mark.gpu
def test_nemo_archive_into_hip_model_step_matches_oracle(tmp_path):
    """SURVEY 8(f).4 end to end on the device: an oracle's weights -> `.nemo` archive (model_config.yaml +
    model_weights.ckpt, as save_restore_connector.py lays it out) -> checkpoint.model_from_nemo -> bf16 HIP model -> one
    training step; losses and gradients against the oracle stepping on the weights the archive was written from."""
    import math

    from oracle import step_ref as S
    from test_parity_configs_gpu import _grad_table, _synth
    torch.manual_seed(21)
    dims = dict(d_model=64, n_layers=3, n_heads=4, pred_hidden=64, joint_hidden=64, languages=['hi', 'ta'], vocab_per_lang=32,
                fused_batch_size=2)
    o = S.OracleHybridModel(**dims)
    with torch.no_grad():
        for l in o.encoder.layers:
            l.self_attn.pos_bias_u.normal_(0, 0.1); l.self_attn.pos_bias_v.normal_(0, 0.1)
    src = EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32', dither=0.0, **{k: v for k, v in dims.items()}))
    src.load_state_dict(o.state_dict())
    p = tmp_path / "oracle.nemo"
    ck.write_nemo(src, p)
    m, rep = ck.model_from_nemo(p, strict=True, languages=dims['languages'], vocab_per_lang=dims['vocab_per_lang'],
                                compute_dtype='bf16', dither=0.0)
    assert not rep.missing_keys and not rep.unexpected_keys
    assert (m.cfg.d_model, m.cfg.n_layers, m.cfg.pred_hidden, m.cfg.fused_batch_size) == (64, 3, 64, 2)
    m = m.disable_dropout().cuda().train()
    m.spec_augment_enabled = False
    S.freeze_layer(o, 1); freeze_layer(m, 1); m.encoder.encoder_frozen_till = 1
    o.train()
    batch = _synth(3, 4.0, seed=9, vocab=32)
    lo, mo = o.training_step(batch, ['ta'] * 3)
    lo.backward()
    lp, mp = m.training_step(tuple(t.cuda() for t in batch), ['ta'] * 3)
    lp.backward()
    torch.cuda.synchronize()
    for k in ('train_rnnt_loss', 'train_ctc_loss', 'train_loss'):
        assert math.isclose(mp[k], mo[k], rel_tol=1e-3), (k, mp[k], mo[k])
    rows = _grad_table(m, o, min_checked=40)
    assert rows[0][0] <= 0.05, rows[:3]
    # ... and the trainable-only .pth of R/utils.py:265-271 written from the device model loads back into the oracle
    q = tmp_path / "task.pth"
    ck.save_trainable(m, q)
    st = torch.load(q)
    assert set(st) == {n for n, prm in o.named_parameters() if prm.requires_grad}
    assert not o.load_state_dict(st, strict=False).unexpected_keys
