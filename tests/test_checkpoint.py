"""Checkpoint interchange: trainable-only .pth (R/utils.py:265-271), .nemo archive round trip
(model_config.yaml + model_weights.ckpt), and the persisted continual-learning state."""
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
