"""Input pipeline: manifest and dataset.pkl formats, WAV decoding + resampling, per-language
SentencePiece ids, NeMo-style padding, duration buckets and rank shards."""
import os
import pickle

import numpy as np
import pytest
import torch

from indic_cl_asr_amd import data as D


from conftest import _tone  # noqa: E402  (the `corpus` fixture lives in conftest.py: shared with the boundary tests)


def test_manifest_and_pkl_formats(corpus, tmp_path):
    root, files, texts, durs = corpus
    m = tmp_path / "manifest.json"
    D.write_manifest(m, files, texts, durs, "hi")
    rows = D.read_manifest(m)
    assert rows[2] == {"audio_filepath": files[2], "duration": durs[2], "text": texts[2], "lang": "hi"}
    pkl = {"train": {"hindi": {"audio": [os.path.basename(f) for f in files],
                               "transcript": {os.path.basename(f): t for f, t in zip(files, texts)}, "duration": durs}},
           "noisy_train": {"hindi": {"audio": [os.path.basename(files[0])],
                                     "transcript": {os.path.basename(files[0]): texts[0]}, "duration": durs[:1]}}}
    p = tmp_path / "dataset.pkl"
    pickle.dump(pkl, open(p, "wb"))
    ds = D.load_dataset_pkl(p, str(root), ["hindi"])
    assert ds["train"]["hindi"]["audio"] == files
    assert ds["noisy_train"]["hindi"]["audio"][0] == files[0]           # noisy_<split> reads from <split>'s folder
    pkl["train"]["hindi"]["audio"][0] = "missing.wav"
    pickle.dump(pkl, open(p, "wb"))
    with pytest.raises(FileNotFoundError):
        D.load_dataset_pkl(p, str(root), ["hindi"])


def test_wav_decode_and_resample(tmp_path):
    x = _tone(8000, 440, sr=8000)
    f = tmp_path / "t8k.wav"
    D.save_wav(str(f), x, sample_rate=8000)
    y8 = D.load_audio(str(f), sample_rate=8000)
    assert y8.dtype == np.float32 and y8.shape == (8000,) and np.abs(y8 - x).max() < 1.0 / 32768 + 1e-6
    y16 = D.load_audio(str(f), sample_rate=16000)
    assert y16.shape == (16000,)
    ref = _tone(16000, 440, sr=16000)
    assert np.abs(y16[200:-200] - ref[200:-200]).max() < 2e-2           # same tone after polyphase resampling


def test_tokenizer_dataset_collate(corpus):
    root, files, texts, durs = corpus
    tok = D.MultilingualTokenizer({"hi": str(root / "hi.model")}, vocab_per_lang=256)
    ids = tok.text_to_ids(texts[1], "hi")
    assert all(0 <= i < 256 for i in ids) and tok.ids_to_text(ids, "hi") == texts[1]
    assert tok.detokenizer("hi")(ids) == texts[1]
    ds = D.SpeechDataset(files, texts, durs, tok, "hi", max_duration=1.2)
    assert len(ds) == 4                                                  # the 1.5 s utterance is filtered out
    sig, sl, t, tl = D.speech_collate([ds[0], ds[1], ds[3]])
    assert sig.shape == (3, 16000) and sl.tolist() == [16000, 8000, 4000]
    assert float(sig[1, 8000:].abs().max()) == 0.0 and float(sig[1, :8000].abs().max()) > 0.1
    assert t.shape[1] == int(tl.max()) and (t[2, int(tl[2]):] == 0).all()


def test_buckets_and_rank_shards():
    durs = [float(d) for d in np.random.RandomState(0).uniform(1, 15, size=103)]
    b = D.batch_indices(durs, 8, shuffle=True, seed=1, bucket=True)
    assert sorted(i for bb in b for i in bb) == list(range(103))
    spread = np.mean([max(durs[i] for i in bb) - min(durs[i] for i in bb) for bb in b if len(bb) == 8])
    plain = D.batch_indices(durs, 8, shuffle=True, seed=1, bucket=False)
    spread_plain = np.mean([max(durs[i] for i in bb) - min(durs[i] for i in bb) for bb in plain if len(bb) == 8])
    assert spread < 0.5 * spread_plain                                    # neighbouring lengths per batch
    r0 = D.batch_indices(durs, 8, shuffle=True, seed=1, rank=0, world=2)
    r1 = D.batch_indices(durs, 8, shuffle=True, seed=1, rank=1, world=2)
    assert len(r0) == len(r1) and not (set(i for bb in r0 for i in bb) & set(i for bb in r1 for i in bb))


def test_batch_loader_host_iteration(corpus):
    root, files, texts, durs = corpus
    tok = D.MultilingualTokenizer({"hi": str(root / "hi.model")})
    ds = D.SpeechDataset(files, texts, durs, tok, "hi")
    got = list(D.BatchLoader(ds, batch_size=2))
    assert len(got) == 3
    (sig, sl, t, tl), (h_sig, h_tok) = got[0]
    assert h_sig == sl.tolist() and h_tok == tl.tolist() and sig.shape[0] == 2


@pytest.mark.gpu
def test_batch_loader_feeds_training_step(corpus):
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel
    root, files, texts, durs = corpus
    tok = D.MultilingualTokenizer({"hi": str(root / "hi.model")}, vocab_per_lang=64)
    ds = D.SpeechDataset(files, texts, durs, tok, "hi")
    torch.manual_seed(0)
    m = EncDecHybridRNNTCTCModel(model_config('tiny', vocab_per_lang=64, compute_dtype='fp32')).cuda().train()
    m.detokenize = tok.detokenizer("hi")
    n = 0
    for batch, host_lens in D.BatchLoader(ds, batch_size=2, device="cuda", bucket=True):
        assert all(t.is_cuda for t in batch)
        loss, mon = m.training_step(batch, ['hi'] * batch[0].shape[0], host_lengths=host_lens, compute_wer=(n == 0))
        assert torch.isfinite(loss)
        if n == 0:
            assert mon['training_batch_wer'] >= 0.0                     # word-level through the detokenizer
        n += 1
    assert n == 3
