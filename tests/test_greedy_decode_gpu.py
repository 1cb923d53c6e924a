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


# ---------------------------------------------------------------------------------------------------------------------------
# the bf16 decode with the head on the matrix cores (16 frames per evaluation) and clusters of workgroups per utterance
def _bf16_case(B, T, Hp, Hj, V, seed, blank_bias, lens=None):
    g = torch.Generator().manual_seed(seed)
    f_all = torch.randn(B, T, Hj, generator=g)
    EW = torch.randn(V + 1, 4 * Hp, generator=g) * 0.5
    Whh = (torch.randn(4 * Hp, Hp, generator=g) / Hp ** 0.5).bfloat16()
    Wp = (torch.randn(Hj, Hp, generator=g) / Hp ** 0.5).bfloat16()
    bp = torch.randn(Hj, generator=g) * 0.1
    Wh = (torch.randn(V, Hj, generator=g) / Hj ** 0.5).bfloat16()
    bh = torch.randn(V, generator=g) * 0.1
    bh[-1] += blank_bias
    if lens is None:
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
    return dict(f_all=f_all, EW=EW, Whh=Whh, Wp=Wp, bp=bp, Wh=Wh, bh=bh, lens=lens.long())


def _bf16_decode_hip(case, cluster, max_symbols):
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    c = {k: v.cuda().contiguous() for k, v in case.items()}
    B, T, Hj = c["f_all"].shape
    Hp = c["Whh"].shape[1]; V = c["Wh"].shape[0]
    cap = T * max_symbols
    tokens = torch.full((B, cap), -1, dtype=torch.int32, device="cuda")
    counts = torch.zeros(B, dtype=torch.int32, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    scratch = torch.empty(int(L.ia_greedy_decode_scratch_bytes(B, Hp, Hj)), dtype=torch.uint8, device="cuda")
    st = L.ia_greedy_rnnt_decode_bf16w_ex(_lib.ptr(c["f_all"]), _lib.ptr(c["lens"]), _lib.ptr(c["EW"]), _lib.ptr(c["Whh"]), _lib.ptr(c["Wp"]),
                                          _lib.ptr(c["bp"]), _lib.ptr(c["Wh"]), _lib.ptr(c["bh"]), B, T, Hp, Hj, V, V - 1, V - 1, V, max_symbols,
                                          _lib.ptr(tokens), cap, _lib.ptr(counts), _lib.ptr(ovf), cluster, _lib.ptr(scratch), scratch.numel(),
                                          _lib.stream_ptr())
    _lib.check(st, "ia_greedy_rnnt_decode_bf16w_ex")
    torch.cuda.synchronize()
    n = counts.cpu().tolist()
    tk = tokens.cpu()
    return [tk[b, :n[b]].tolist() for b in range(B)], int(ovf.item())


def _bf16_decode_torch(case, max_symbols, act_bf16):
    """The per-utterance reference loop (rnnt_greedy_decoding.py:711-909 as csrc/greedy_decode.hip restates it) in fp32 torch on
    the bf16-rounded weights; act_bf16: the head's input rounded to bf16 as the MFMA kernel (and the training joint) does."""
    f_all, EW, lens = case["f_all"], case["EW"], case["lens"]
    Whh, Wp, Wh = case["Whh"].float(), case["Wp"].float(), case["Wh"].float()
    bp, bh = case["bp"], case["bh"]
    Hp = Whh.shape[1]; V = Wh.shape[0]; blank = V - 1
    out = []
    for b in range(f_all.shape[0]):
        h = torch.zeros(Hp); c = torch.zeros(Hp)

        def pend(row):
            g = Whh @ h + EW[row]
            i, f, gg, o = torch.sigmoid(g[:Hp]), torch.sigmoid(g[Hp:2 * Hp]), torch.tanh(g[2 * Hp:3 * Hp]), torch.sigmoid(g[3 * Hp:])
            cn = f * c + i * gg
            hn = o * torch.tanh(cn)
            return hn, cn, Wp @ hn + bp
        hn, cn, gp = pend(V)
        first, emitted, hyp = True, False, []
        for t in range(int(lens[b])):
            for s in range(max_symbols):
                act = torch.relu(f_all[b, t] + gp)
                if act_bf16:
                    act = act.bfloat16().float()
                k = int(torch.argmax(Wh @ act + bh))
                if k == blank:
                    if first and not emitted:
                        hn, cn, gp = pend(blank)
                    first = False
                    break
                first = False; emitted = True
                hyp.append(k)
                h, c = hn, cn
                hn, cn, gp = pend(k)
        out.append(hyp)
    return out


@pytest.mark.parametrize("B,T,Hp,Hj,V,blank_bias,ms", [
    (11, 45, 640, 640, 257, 3.0, 10),     # the medium model's dimensions, ragged lengths, B not a multiple of 8
    (5, 37, 64, 96, 33, 0.8, 3),          # small slices (Hp / 4 = 16 units per workgroup), V not a multiple of 16, cap of 3 per frame
    (3, 20, 128, 64, 17, 50.0, 10),       # all blank
    (4, 18, 64, 64, 40, -50.0, 2),        # never blank: every frame emits max_symbols labels
])
def test_bf16_decode_mfma_head_and_clusters(B, T, Hp, Hj, V, blank_bias, ms):
    from indic_cl_asr_amd import _lib
    case = _bf16_case(B, T, Hp, Hj, V, seed=B * 100 + T, blank_bias=blank_bias)
    one, ovf = _bf16_decode_hip(case, 1, ms)
    assert ovf == 0
    for nw in (2, 4):
        if Hp % (8 * nw) == 0 and Hj % (4 * nw) == 0:
            got, ovf = _bf16_decode_hip(case, nw, ms)
            assert ovf == 0 and got == one, nw        # the same dot products, whoever computes them
    ref = _bf16_decode_torch(case, ms, act_bf16=True)
    n_ref = sum(len(h) for h in ref)
    same = sum(int(a == b) for a, b in zip(one, ref))
    print("symbols", n_ref, "utterances identical", same, "of", B)
    if blank_bias > 10:
        assert n_ref == 0 and one == ref
    elif blank_bias < -10:
        assert one == ref or same >= B - 1
        assert all(len(h) == ms * int(l) for h, l in zip(one, case["lens"]))
    else:
        assert n_ref > 0
        assert same >= B - max(1, B // 8), (one, ref)   # fp32 accumulation order differs: a near-tie may flip one path
    # the GEMV loop (fp32 activations into the head) decodes the same weights; it differs only by the bf16 rounding of the head's input
    gemv, _ = _bf16_decode_hip(case, 0, ms)
    ref32 = _bf16_decode_torch(case, ms, act_bf16=False)
    assert sum(int(a == b) for a, b in zip(gemv, ref32)) >= B - max(1, B // 8)
    assert int(_lib.lib().ia_greedy_decode_cluster(32, 640, 640)) == 4 and int(_lib.lib().ia_greedy_decode_cluster(32, 640, 100)) == 0


def test_bf16_decode_lost_handoff_is_flagged_and_the_launch_drains(monkeypatch):
    case = _bf16_case(6, 30, 64, 64, 33, seed=9, blank_bias=0.5)
    monkeypatch.setenv("IA_DECODE_SPIN_LIMIT", "0")          # every wait gives up at its first poll
    _, ovf = _bf16_decode_hip(case, 4, 5)
    monkeypatch.delenv("IA_DECODE_SPIN_LIMIT")
    assert ovf & 2                                           # (and the launch returned: no workgroup waits for ever)
    good, ovf = _bf16_decode_hip(case, 4, 5)
    assert ovf == 0 and good == _bf16_decode_hip(case, 1, 5)[0]
