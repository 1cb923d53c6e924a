"""Greedy decoding + WER: the batched device loop against the per-utterance oracle restatement, the
CTC collapse rule, and the edit-distance metric on known answers (wer.py:58-60 uses editdistance.eval)."""
import pytest
import torch

from oracle import step_ref as S


def _models(seed=0):
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel
    torch.manual_seed(seed)
    o = S.OracleHybridModel(d_model=32, n_layers=2, n_heads=4, pred_hidden=24, joint_hidden=24, languages=['hi', 'ta'],
                            vocab_per_lang=16, fused_batch_size=2)
    with torch.no_grad():   # make non-blank symbols likely enough to exercise the inner loop
        for l in ('hi', 'ta'):
            o.joint.joint_net[-1][l].bias[-1] -= 1.0
    m = EncDecHybridRNNTCTCModel(model_config('tiny', n_layers=2, compute_dtype='fp32', dither=0.0))
    m.load_state_dict(o.state_dict())
    return o.eval(), m.eval()


def test_edit_distance_and_wer_known_answers():
    from indic_cl_asr_amd.decoding import WER, _edit_distance, word_error_rate
    assert _edit_distance("kitten", "sitting") == 3
    assert _edit_distance([1, 2, 3], [1, 2, 3]) == 0 and _edit_distance([], [1, 2]) == 2 and _edit_distance([5], []) == 1
    wer, s, w = word_error_rate([[1, 2, 3], [4]], [[1, 3], [4, 5, 6]])
    assert (s, w) == (1 + 2, 5) and abs(wer - 0.6) < 1e-12
    words = {1: "the", 2: "cat", 3: "sat", 4: "mat"}
    det = lambda ids: " ".join(words[i] for i in ids)
    m = WER(detokenize=det)
    m.update([[1, 2, 3]], [[1, 2, 4]]); m.update([[1]], [[1, 2]])
    assert m.compute() == (2 / 5, 2, 5)
    m.reset()
    assert m.compute()[0] == float("inf")


def test_native_edit_distance_batch_equals_the_dynamic_programme():
    """csrc/host_metrics.hip (the `editdistance` extension of wer.py:58-60, batched) against the plain two-row programme on
    random token, word and character sequences, empty sides included."""
    import random
    from indic_cl_asr_amd.decoding import _edit_distance_py, _edit_distances
    rnd = random.Random(3)
    pairs = [([rnd.randint(0, 12) for _ in range(rnd.randint(0, 90))], [rnd.randint(0, 12) for _ in range(rnd.randint(0, 60))])
             for _ in range(40)]
    pairs += [([], [1, 2, 3]), ([4], []), ([], []), ([7] * 50, [7] * 50)]
    words = "the cat sat on a mat with one red hat and two big dogs too".split()
    pairs += [([rnd.choice(words) for _ in range(rnd.randint(5, 40))], [rnd.choice(words) for _ in range(rnd.randint(5, 40))]) for _ in range(10)]
    pairs += [(list("kitten sitting on the mitten"), list("sitting kitten in the kitchen"))]
    assert _edit_distances(pairs) == [_edit_distance_py(a, b) for a, b in pairs]


def test_greedy_rnnt_matches_per_utterance_oracle_cpu():
    from indic_cl_asr_amd.decoding import greedy_rnnt_decode
    o, m = _models()
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(4, 32, 23, generator=g)
    lens = torch.tensor([23, 17, 9, 1])
    for lang in ('hi', 'ta'):
        hyp = greedy_rnnt_decode(m, enc, lens, [lang] * 4, max_symbols=3)
        ref = S.greedy_rnnt_decode_ref(o, enc, lens, lang, max_symbols=3)
        assert hyp == ref
        assert any(len(h) > 0 for h in hyp)                 # the test is not vacuous
        assert max(len(h) for h in hyp) <= 3 * 23


def test_greedy_rnnt_of_a_mixed_language_batch_decodes_each_utterance_through_its_own_head():
    from indic_cl_asr_amd.decoding import greedy_rnnt_decode
    o, m = _models()
    g = torch.Generator().manual_seed(4)
    enc = torch.randn(4, 32, 19, generator=g)
    lens = torch.tensor([19, 12, 7, 19])
    langs = ['hi', 'ta', 'ta', 'hi']
    hyp = greedy_rnnt_decode(m, enc, lens, langs, max_symbols=3)
    for i, lang in enumerate(langs):
        assert hyp[i] == S.greedy_rnnt_decode_ref(o, enc[i:i + 1], lens[i:i + 1], lang, max_symbols=3)[0]
    assert any(len(h) > 0 for h in hyp)


def test_greedy_ctc_collapse_rule():
    from indic_cl_asr_amd.decoding import greedy_ctc_decode
    g = torch.Generator().manual_seed(5)
    lp = torch.randn(3, 19, 6, generator=g).log_softmax(-1)
    lens = torch.tensor([19, 11, 0])
    assert greedy_ctc_decode(lp, lens) == S.greedy_ctc_decode_ref(lp, lens, blank=5)
    hand = torch.full((1, 6, 3), -10.0)
    for t, k in enumerate([0, 0, 2, 1, 1, 0]):              # blank = 2: "0 0 _ 1 1 0" -> 0 1 0
        hand[0, t, k] = 0.0
    assert greedy_ctc_decode(hand, torch.tensor([6])) == [[0, 1, 0]]


@pytest.mark.gpu
def test_greedy_rnnt_on_device_matches_oracle():
    from indic_cl_asr_amd.decoding import greedy_rnnt_decode
    o, m = _models(seed=1)
    m = m.cuda()
    g = torch.Generator().manual_seed(7)
    enc = torch.randn(5, 32, 31, generator=g)
    lens = torch.tensor([31, 30, 12, 5, 1])
    hyp = greedy_rnnt_decode(m, enc.cuda(), lens.cuda(), ['hi'] * 5, max_symbols=4)
    ref = S.greedy_rnnt_decode_ref(o, enc, lens, 'hi', max_symbols=4)
    same = sum(h == r for h, r in zip(hyp, ref))
    assert same >= 4, (hyp, ref)    # fp32 GPU vs CPU argmax ties may flip one path; sequences must agree otherwise


@pytest.mark.gpu
def test_training_step_monitor_carries_batch_wer_when_asked():
    o, m = _models()
    m = m.cuda()
    m.train(); m.spec_augment_enabled = False
    g = torch.Generator().manual_seed(11)
    sig = torch.randn(3, 8000, generator=g) * 0.1
    sl = torch.tensor([8000, 6000, 4000]); tr = torch.randint(0, 16, (3, 6), generator=g); tl = torch.tensor([6, 3, 1])
    batch = tuple(t.cuda() for t in (sig, sl, tr, tl))
    loss, mon = m.training_step(batch, ['hi'] * 3, compute_wer=True)
    # the reference's types: a 0-dim tensor from WER.compute for the transducer rate, a python float for the CTC rate
    assert torch.is_tensor(mon['training_batch_wer']) and mon['training_batch_wer'].dim() == 0
    assert isinstance(mon['training_batch_wer_ctc'], float)
    for key in ('training_batch_wer', 'training_batch_wer_ctc'):
        assert float(mon[key]) >= 0.0 and float(mon[key]) == float(mon[key])
    loss2, mon2 = m.training_step(batch, ['hi'] * 3, compute_wer=False)
    assert mon2['training_batch_wer_ctc'] != mon2['training_batch_wer_ctc']     # NaN when switched off
    m.compute_wer_in_step = False                                                # ... by the attribute, for callers that cannot pass it
    _, mon2b = m.training_step(batch, ['hi'] * 3)
    assert mon2b['training_batch_wer_ctc'] != mon2b['training_batch_wer_ctc']
    m.compute_wer_in_step = None
    _, mon2c = m.training_step(batch, ['hi'] * 3)                                # the reference's call: both rates, as it computes them
    assert float(mon2c['training_batch_wer']) >= 0.0 and mon2c['training_batch_wer_ctc'] == mon2c['training_batch_wer_ctc']
    # the deferred form (decode enqueued on a side stream behind the encoder output, scored on the monitor's first read) and the
    # synchronous one give the same numbers
    m.disable_dropout()
    res = {}
    for deferred in (True, False):
        m.defer_wer = deferred
        _, mon3 = m.training_step(batch, ['hi'] * 3, compute_wer=True)
        res[deferred] = (float(mon3['training_batch_wer']), float(mon3['training_batch_wer_ctc']), float(mon3['train_loss']))
    m.defer_wer = True
    assert res[True] == res[False]


def test_wer_metric_objects_follow_the_reference_surface():
    """model.wer / model.ctc_wer: update / compute / reset (A/metrics/wer.py:293-360) -- word units through the
    detokenizer, update REPLACES the state, compute returns (rate, edits, words) tensors, grouped() is the fused joint's
    mean over sub-batch rates (A/modules/rnnt.py:1548-1553)."""
    o, m = _models()
    words = {i: w for i, w in enumerate("the cat sat on a mat with one red hat and two big dogs too".split())}
    m.detokenize = lambda ids: " ".join(words[int(i)] for i in ids)
    V = 17
    def lp_for(seqs, T):          # log-probs whose greedy CTC path spells `seqs` (blank between the symbols)
        lp = torch.full((len(seqs), T, V), -20.0)
        for b, seq in enumerate(seqs):
            path = []
            for k in seq:
                path += [k, V - 1]
            path += [V - 1] * (T - len(path))
            for t, k in enumerate(path):
                lp[b, t, k] = 0.0
        return lp
    hyp = [[0, 1, 2], [3, 4]]
    tgt = torch.tensor([[0, 1, 5, 0], [3, 4, 0, 0]]); tl = torch.tensor([3, 2])          # "the cat mat" / "on a"
    m.ctc_wer.log_prediction = False
    m.ctc_wer.update(predictions=lp_for(hyp, 9), predictions_lengths=torch.tensor([9, 9]), targets=tgt, targets_lengths=tl,
                     lang_ids=['hi', 'hi'])
    wer, s, w = m.ctc_wer.compute()
    assert (float(s), float(w)) == (1.0, 5.0) and abs(float(wer) - 0.2) < 1e-7 and wer.dim() == 0
    m.ctc_wer.update(predictions=lp_for([[0]], 4), predictions_lengths=torch.tensor([4]), targets=torch.tensor([[0, 1]]),
                     targets_lengths=torch.tensor([2]), lang_ids=['hi'])
    assert [float(v) for v in m.ctc_wer.compute()[1:]] == [1.0, 2.0]       # replaced, not accumulated (wer.py:359-360)
    m.ctc_wer.reset()
    assert int(m.ctc_wer.scores) == 0 and int(m.ctc_wer.words) == 0
    # character error rate switch
    m.ctc_wer.use_cer = True
    m.ctc_wer.update(predictions=lp_for([[1]], 4), predictions_lengths=torch.tensor([4]), targets=torch.tensor([[5]]),
                     targets_lengths=torch.tensor([1]), lang_ids=['hi'])
    assert [float(v) for v in m.ctc_wer.compute()[1:]] == [1.0, 3.0]       # "cat" vs "mat"
    m.ctc_wer.use_cer = False
    # grouped: sub-batches of 2 -> mean of (1/5, 2/2)
    g, gs, gw = m.wer.grouped([[0, 1, 2], [3, 4], [6], []], [[0, 1, 5], [3, 4], [7, 8], []][:3] + [[9]], ['hi'] * 4, 2)
    assert (float(gs), float(gw)) == (1.0 + 3.0, 5.0 + 3.0)
    assert abs(float(g) - 0.5 * (1 / 5 + 3 / 3)) < 1e-7
    # the transducer metric decodes through the model: same ids as decoding.greedy_rnnt_decode, scored as strings
    from indic_cl_asr_amd.decoding import greedy_rnnt_decode
    gen = torch.Generator().manual_seed(3)
    enc = torch.randn(2, 32, 12, generator=gen); lens = torch.tensor([12, 7])
    ids = greedy_rnnt_decode(m, enc, lens, ['hi'] * 2)
    m.wer.log_prediction = False
    m.wer.update(predictions=enc, predictions_lengths=lens, targets=torch.tensor([[1, 2, 3], [4, 0, 0]]),
                 targets_lengths=torch.tensor([3, 1]), lang_ids=['hi'] * 2)
    _, s2, w2 = m.wer.compute()
    from indic_cl_asr_amd.decoding import word_error_rate
    _, es, ew = word_error_rate(ids, [[1, 2, 3], [4]], m.detokenize)
    assert (float(s2), float(w2)) == (float(es), float(ew))
