"""csrc/greedy_decode.hip (device-resident greedy transducer decoding: one launch, no host read per micro-step) against the
host-driven loop of indic_cl_asr_amd.decoding (the restatement of rnnt_greedy_decoding.py:711-909 that tests/test_decoding.py
pins against a per-utterance oracle): identical token sequences on the same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(**kw):
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel
    torch.manual_seed(0)
    m = EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32', **kw)).cuda().eval()
    return m


@pytest.mark.parametrize("kw,T,blank_bias,max_symbols", [
    (dict(d_model=64, n_layers=1, n_heads=1, pred_hidden=64, joint_hidden=64, vocab_per_lang=32), 40, 2.0, 10),
    (dict(d_model=64, n_layers=1, n_heads=1, pred_hidden=640, joint_hidden=640, vocab_per_lang=256), 60, 3.0, 10),
    (dict(d_model=64, n_layers=1, n_heads=1, pred_hidden=128, joint_hidden=320, vocab_per_lang=256), 50, 0.0, 3),
    (dict(d_model=64, n_layers=1, n_heads=1, pred_hidden=64, joint_hidden=64, vocab_per_lang=32), 30, 50.0, 10),   # all blank
])
def test_device_decode_equals_the_host_driven_loop(kw, T, blank_bias, max_symbols):
    from indic_cl_asr_amd import decoding as D
    m = _model(**kw)
    lang = m.cfg.languages[0]
    with torch.no_grad():   # a head that emits blanks and labels in comparable numbers
        head = m.joint.joint_net[-1][lang]
        head.weight.mul_(3.0)
        head.bias[-1] += blank_bias
    g = torch.Generator().manual_seed(T)
    B = 5
    enc = torch.randn(B, kw["d_model"], T, generator=g).cuda()
    lens = torch.tensor([T, T - 7, 1, T // 2, T - 1]).cuda()
    host = D.greedy_rnnt_decode_host(m, enc, lens, [lang] * B, max_symbols)
    dev = D.greedy_rnnt_decode_device(m, enc, lens, [lang] * B, max_symbols)
    assert dev == host
    n = sum(len(h) for h in host)
    print("tokens", n, "of", int(lens.sum()) * max_symbols)
    if blank_bias < 10:
        assert n > 0
    else:
        assert n == 0
    assert D.greedy_rnnt_decode(m, enc, lens, [lang] * B, max_symbols) == host      # the dispatching entry


def test_device_decode_uses_the_label_state_after_the_first_blank():
    """An utterance whose first micro-step is blank continues from embedding[blank_idx] with a zero state: make that row non-zero."""
    from indic_cl_asr_amd import decoding as D
    m = _model(d_model=64, n_layers=1, n_heads=1, pred_hidden=64, joint_hidden=64, vocab_per_lang=32)
    lang = m.cfg.languages[0]
    with torch.no_grad():
        m.decoder.prediction["embed"].weight[m.decoder.blank_idx] = torch.randn(64).cuda()
        head = m.joint.joint_net[-1][lang]
        head.weight.mul_(3.0); head.bias[-1] += 1.5
    enc = torch.randn(4, 64, 35, generator=torch.Generator().manual_seed(1)).cuda()
    lens = torch.tensor([35, 20, 35, 9]).cuda()
    assert D.greedy_rnnt_decode_device(m, enc, lens, [lang] * 4, 10) == D.greedy_rnnt_decode_host(m, enc, lens, [lang] * 4, 10)


def test_device_decode_of_an_empty_utterance_emits_nothing():
    from indic_cl_asr_amd import decoding as D
    m = _model(d_model=64, n_layers=1, n_heads=1, pred_hidden=64, joint_hidden=64, vocab_per_lang=32)
    lang = m.cfg.languages[0]
    enc = torch.randn(3, 64, 20).cuda()
    lens = torch.tensor([20, 0, 7]).cuda()
    out = D.greedy_rnnt_decode_device(m, enc, lens, [lang] * 3, 5)
    assert out[1] == [] and out == D.greedy_rnnt_decode_host(m, enc, lens, [lang] * 3, 5)
