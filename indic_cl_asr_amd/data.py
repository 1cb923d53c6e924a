"""Input pipeline and data formats of the CL scripts, host side.

* manifests -- JSON lines `{audio_filepath, duration, text, lang}` as `_transcribe_input_processing` writes them
  (A/models/hybrid_rnnt_ctc_models.py:420-431): `write_manifest` / `read_manifest`.
* `dataset.pkl` -- `{split: {lang: {audio: [file…], transcript: {file: text}, duration: […]}}}` with the audio paths
  expanded to `<root>/<split without noisy_>/<lang>/<file>` (R/cl_baseline.py:80-90): `load_dataset_pkl`.
* audio -- RIFF/WAVE PCM16 / PCM32 / float32 through the standard library (`soundfile` / `librosa`, which the reference's
  AudioSegment.from_file uses, are not part of this image), mono mix-down and polyphase resampling to the model rate.
* tokenisation -- one SentencePiece model per language, ids local to the language's 256-entry block as the
  language-restricted joint / CTC heads expect (C/tokenizers/multilingual_tokenizer.py): `MultilingualTokenizer`.
* batching -- (signal, signal_len, tokens, tokens_len) zero-padded like NeMo's `_speech_collate_fn`, optional
  duration bucketing (neighbouring lengths per batch: fewer padded frames through the T'-quadratic attention), rank
  sharding for data parallel runs ("ddp" sampler of the reference), pinned host buffers + asynchronous H2D copies with
  the host-side length lists `training_step(..., host_lengths=…)` wants.
"""
import json
import os
import pickle
import random
import struct
import wave
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------- manifests / pkl
def write_manifest(path, audio_files, transcripts, durations, language_id):
    with open(path, "w", encoding="utf-8") as fp:
        for a, t, d in zip(audio_files, transcripts, durations):
            fp.write(json.dumps({"audio_filepath": a, "duration": d, "text": t, "lang": language_id}, ensure_ascii=False) + "\n")


def read_manifest(path) -> List[dict]:
    with open(path, encoding="utf-8") as fp:
        return [json.loads(line) for line in fp if line.strip()]


def load_dataset_pkl(annotation_path, dataset_root, languages: Sequence[str], check_files=True) -> dict:
    """R/cl_baseline.py:80-90: expand file names to paths and check that the first file of every split/language exists and has
    a transcript."""
    with open(annotation_path, "rb") as fh:
        dataset = pickle.load(fh)
    for split, per_lang in dataset.items():
        for lang in languages:
            rec = per_lang[lang]
            rec["audio"] = [os.path.join(dataset_root, split.replace("noisy_", ""), lang, f) for f in rec["audio"]]
            if check_files and rec["audio"]:
                if not os.path.exists(rec["audio"][0]):
                    raise FileNotFoundError(rec["audio"][0])
                if os.path.basename(rec["audio"][0]) not in rec["transcript"]:
                    raise KeyError(f"Transcript not found for {rec['audio'][0]}")
    return dataset


# ---------------------------------------------------------------------------------------------------- audio
def load_audio(path, sample_rate=16000) -> np.ndarray:
    """float32 mono samples in [-1, 1) at `sample_rate`."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        # PCM32 or IEEE float: the wave module only accepts format tag 1, so this is PCM32
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    if rate != sample_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sample_rate))
        x = resample_poly(x, sample_rate // g, rate // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def save_wav(path, samples: np.ndarray, sample_rate=16000):
    pcm = np.clip(np.round(np.asarray(samples, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())


# ---------------------------------------------------------------------------------------------------- tokenisation
class MultilingualTokenizer:
    """lang -> SentencePiece processor; ids are local to the language (0 … vocab_per_lang-1)."""

    def __init__(self, model_files: Dict[str, str], vocab_per_lang=256):
        import sentencepiece as spm
        self.vocab_per_lang = vocab_per_lang
        self.sp = {}
        for lang, f in model_files.items():
            p = spm.SentencePieceProcessor()
            p.Load(f)
            if p.GetPieceSize() > vocab_per_lang:
                raise ValueError(f"{lang}: {p.GetPieceSize()} pieces > vocab_per_lang={vocab_per_lang}")
            self.sp[lang] = p

    def text_to_ids(self, text, lang) -> List[int]:
        return list(self.sp[lang].EncodeAsIds(text))

    def ids_to_text(self, ids, lang) -> str:
        sp = self.sp[lang]
        n = sp.GetPieceSize()   # a head wider than the language's piece table can emit ids without a piece: dropped
        return sp.DecodeIds([int(i) for i in ids if 0 <= int(i) < n])

    def detokenizer(self, lang) -> Callable:
        """`model.detokenize` / decoding.word_error_rate(detokenize=…) hook for word-level WER."""
        return lambda ids: self.ids_to_text(ids, lang)


# ---------------------------------------------------------------------------------------------------- dataset / batches
class SpeechDataset:
    def __init__(self, audio_files, transcripts, durations, tokenizer: MultilingualTokenizer, language_id, sample_rate=16000,
                 max_duration: Optional[float] = None):
        keep = [i for i, d in enumerate(durations) if max_duration is None or d <= max_duration]
        self.audio = [audio_files[i] for i in keep]
        self.text = [transcripts[i] for i in keep]
        self.dur = [float(durations[i]) for i in keep]
        self.tok, self.lang, self.sr = tokenizer, language_id, sample_rate

    def __len__(self):
        return len(self.audio)

    def __getitem__(self, i):
        x = torch.from_numpy(load_audio(self.audio[i], self.sr))
        t = torch.tensor(self.tok.text_to_ids(self.text[i], self.lang), dtype=torch.long)
        return x, torch.tensor(x.shape[0], dtype=torch.long), t, torch.tensor(t.shape[0], dtype=torch.long)


def speech_collate(samples, pad_id=0):
    """NeMo `_speech_collate_fn`: right-pad signals with zeros and token rows with pad_id."""
    sig_len = torch.stack([s[1] for s in samples])
    tok_len = torch.stack([s[3] for s in samples])
    L, U = int(sig_len.max()), max(1, int(tok_len.max()))
    sig = torch.zeros(len(samples), L, dtype=torch.float32)
    tok = torch.full((len(samples), U), pad_id, dtype=torch.long)
    for i, (x, n, t, m) in enumerate(samples):
        sig[i, :int(n)] = x
        tok[i, :int(m)] = t
    return sig, sig_len, tok, tok_len


def batch_indices(durations, batch_size, shuffle=False, seed=0, bucket=False, rank=0, world=1, drop_last=False):
    """Index lists per batch.  `bucket`: sort by duration inside windows of 50 batches so that a batch holds neighbouring
    lengths (the order of batches is still shuffled); `world` > 1: every rank takes every world-th batch ("ddp" sampler
    of the reference: disjoint shards, same number of batches on every rank)."""
    idx = list(range(len(durations)))
    rng = random.Random(seed)
    if shuffle:
        rng.shuffle(idx)
    if bucket:
        win = 50 * batch_size
        idx = [j for s in range(0, len(idx), win) for j in sorted(idx[s:s + win], key=lambda k: durations[k])]
    batches = [idx[s:s + batch_size] for s in range(0, len(idx), batch_size)]
    if drop_last and batches and len(batches[-1]) < batch_size:
        batches.pop()
    if bucket and shuffle:
        rng.shuffle(batches)
    if world > 1:
        n = len(batches) // world * world
        batches = batches[rank:n:world]
    return batches


class BatchLoader:
    """Iterates device batches: collate on the host into pinned buffers, copy on a side stream one batch ahead, hand the
    training loop (batch, host_lengths) so that training_step needs no device->host read."""

    def __init__(self, dataset: SpeechDataset, batch_size, device=None, shuffle=False, seed=0, bucket=False, rank=0, world=1,
                 drop_last=False):
        self.ds, self.device = dataset, device
        self.batches = batch_indices(dataset.dur, batch_size, shuffle, seed, bucket, rank, world, drop_last)

    def __len__(self):
        return len(self.batches)

    def _host_batch(self, ids):
        b = speech_collate([self.ds[i] for i in ids])
        lens = (b[1].tolist(), b[3].tolist())
        if self.device is not None and torch.device(self.device).type == "cuda":
            b = tuple(t.pin_memory() for t in b)
        return b, lens

    def __iter__(self) -> Iterator[Tuple[tuple, tuple]]:
        dev = torch.device(self.device) if self.device is not None else None
        if dev is None or dev.type != "cuda":
            for ids in self.batches:
                yield self._host_batch(ids)
            return
        copy = torch.cuda.Stream(device=dev)
        nxt = None
        for ids in self.batches + [None]:
            cur = nxt
            if ids is not None:
                host, lens = self._host_batch(ids)
                with torch.cuda.stream(copy):
                    devb = tuple(t.to(dev, non_blocking=True) for t in host)
                ev = torch.cuda.Event(); ev.record(copy)
                nxt = (devb, lens, ev, host)       # `host` keeps the pinned buffers alive until the copy has run
            if cur is not None:
                devb, lens, ev, _ = cur
                torch.cuda.current_stream(dev).wait_event(ev)
                for t in devb:
                    t.record_stream(torch.cuda.current_stream(dev))
                yield devb, lens


def move_to_device(batch, device):
    """The reference's helper of the same name (R/cl_baseline.py): tuple of tensors -> device."""
    return tuple(t.to(device, non_blocking=True) for t in batch)
