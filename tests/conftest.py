import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _tone(n, f, sr=16000):
    import numpy as np
    return (0.3 * np.sin(2 * np.pi * f * np.arange(n) / sr)).astype(np.float32)


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    """Five short WAV files + a SentencePiece model trained on their transcripts: (root, files, texts, durations)."""
    import sentencepiece as spm
    from indic_cl_asr_amd import data as D
    root = tmp_path_factory.mktemp("data")
    texts = ["namaste duniya", "yah ek pariksha hai", "duniya gol hai", "ek do teen char", "pariksha safal"] * 8
    (root / "corpus.txt").write_text("\n".join(texts))
    spm.SentencePieceTrainer.Train(input=str(root / "corpus.txt"), model_prefix=str(root / "hi"), vocab_size=40,
                                   model_type="unigram", hard_vocab_limit=False, minloglevel=2)
    files, durs = [], []
    os.makedirs(root / "train" / "hindi")
    for i, n in enumerate((16000, 8000, 24000, 12000, 4000)):
        f = root / "train" / "hindi" / f"u{i}.wav"
        D.save_wav(str(f), _tone(n, 200 + 50 * i))
        files.append(str(f)); durs.append(n / 16000)
    return root, files, texts[:5], durs
