"""The transcription / data-loader surface the CL scripts call on the model, with the reference's
signatures:

  model._transcribe_input_processing(audio, trcfg, transcripts=None, durations=None, language_id='ta', shuffle=False,
                                     sampler=None) -> DataLoader        A/models/hybrid_rnnt_ctc_models.py:498-547
      callers: R/cl_baseline.py:172-175 (training loader), the model's own transcribe()
  model.transcribe(audio, batch_size=4, return_hypotheses=False, num_workers=0, verbose=True, logprobs=False,
                   language_id=None, channel_selector=None, augmentor=None, override_config=None, **config_kwargs)
      -> (hypotheses, all_hypotheses)                                    :262-340;  caller R/utils.py:120-145 takes [0]

What differs in HOW: no temporary manifest + NeMo dataset classes -- the loader is data.BatchLoader (pinned buffers,
asynchronous H2D one batch ahead); it yields tuples that are ALREADY on the model's device, so the scripts'
`move_to_device(batch, device)` (R/cl_baseline.py:49-58: `tensor.to(device)`) is a no-op, and it registers the batch's
host-side lengths under the device tensors' addresses: `training_step(batch, lang_ids)` -- the reference's exact call --
finds them there and needs no device->host read (a D2H read at that point would wait for everything the host has
queued: the previous step).  Tokenisation needs the per-language SentencePiece models, which the reference gets from
its `.nemo` archive: attach a data.MultilingualTokenizer with `model.set_tokenizer(tok)`.
"""
import os
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import data as D

# device address of a batch's length tensor -> host list (entries die with the tensor)
_HOST_LENGTHS = {}


def register_host_lengths(tensor: torch.Tensor, values: List[int]):
    key = (tensor.device.index, tensor.data_ptr())
    _HOST_LENGTHS[key] = (weakref.ref(tensor), list(values))
    weakref.finalize(tensor, _HOST_LENGTHS.pop, key, None)


def lookup_host_lengths(tensor: torch.Tensor) -> Optional[List[int]]:
    if not isinstance(tensor, torch.Tensor) or not tensor.is_cuda:
        return None
    hit = _HOST_LENGTHS.get((tensor.device.index, tensor.data_ptr()))
    if hit is None or hit[0]() is not tensor:
        return None
    return hit[1]


class _TensorDataset:
    """Audio given as tensors / arrays (hybrid_rnnt_ctc_models.py:527-537): no transcripts."""

    def __init__(self, tensors):
        self.items = [torch.as_tensor(t, dtype=torch.float32).reshape(-1) for t in tensors]
        self.dur = [float(t.numel()) for t in self.items]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        x = self.items[i]
        return x, torch.tensor(x.numel(), dtype=torch.long), torch.zeros(0, dtype=torch.long), torch.tensor(0, dtype=torch.long)


class _FileDataset(D.SpeechDataset):
    """Audio files with optional transcripts; without a tokenizer (plain transcription) the token side is empty."""

    def __init__(self, audio_files, transcripts, durations, tokenizer, language_id, sample_rate):
        self.audio, self.text = list(audio_files), list(transcripts)
        self.dur = [float(d) for d in durations]
        self.tok, self.lang, self.sr = tokenizer, language_id, sample_rate

    def __getitem__(self, i):
        x = torch.from_numpy(D.load_audio(self.audio[i], self.sr))
        if self.tok is not None and self.text[i]:
            t = torch.tensor(self.tok.text_to_ids(self.text[i], self.lang), dtype=torch.long)
        else:
            t = torch.zeros(0, dtype=torch.long)
        return x, torch.tensor(x.shape[0], dtype=torch.long), t, torch.tensor(t.shape[0], dtype=torch.long)


class TranscribeLoader:
    """What `_transcribe_input_processing` returns: an iterable with len() over 4-tuples
    (signal [B,L] f32, signal_len [B] i64, tokens [B,U] i64, tokens_len [B] i64) -- the layout of NeMo's
    `_speech_collate_fn` (A/data/audio_to_text.py:57-116) -- resident on `device` when that is the MI355X."""

    def __init__(self, dataset, batch_size, device, shuffle=False, sampler=None, seed=0):
        rank, world = 0, 1
        if sampler == "ddp" and torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        self.loader = D.BatchLoader(dataset, batch_size, device=device, shuffle=shuffle, seed=seed, rank=rank, world=world)
        self.epoch = 0

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch, (h_sig, h_tok) in self.loader:
            batch = tuple(batch)
            if batch[1].is_cuda:
                register_host_lengths(batch[1], h_sig)
                register_host_lengths(batch[3], h_tok)
            yield batch


class TranscriptionMixin:
    """Mixed into EncDecHybridRNNTCTCModel."""

    tokenizer = None

    def set_tokenizer(self, tokenizer):
        """data.MultilingualTokenizer (one SentencePiece model per language; the reference reads them from the .nemo file)."""
        self.tokenizer = tokenizer
        return self

    # -------------------------------------------------------------------------------------------------- loader
    def _transcribe_input_processing(self, audio, trcfg, transcripts=None, durations=None, language_id='ta', shuffle=False,
                                     sampler=None):
        if isinstance(audio, (list, tuple)):
            if len(audio) == 0:
                raise ValueError("Input `audio` is empty")
        else:
            audio = [audio]
        device = getattr(getattr(trcfg, "_internal", None), "device", None) or next(self.parameters()).device
        bs = int(getattr(trcfg, "batch_size", 4) or 4)
        if isinstance(audio[0], str):
            files = list(audio)
            if transcripts is not None and self.tokenizer is None:
                raise RuntimeError("transcripts were given but the model has no tokenizer: call model.set_tokenizer("
                                   "data.MultilingualTokenizer({lang: spm_model_file, ...})) first")
            if transcripts is None:
                transcripts = [''] * len(files)
            if durations is None:
                durations = [0] * len(files)
            tmp = getattr(getattr(trcfg, "_internal", None), "temp_dir", None)
            if tmp:   # the reference leaves a manifest there (hybrid_rnnt_ctc_models.py:420-431): same file, same fields
                D.write_manifest(os.path.join(tmp, "manifest.json"), files, transcripts, durations, language_id)
            ds = _FileDataset(files, transcripts, durations, self.tokenizer, language_id, self.cfg.sample_rate)
            return TranscribeLoader(ds, min(bs, len(files)), device, shuffle=shuffle, sampler=sampler, seed=self.seed)
        if isinstance(audio[0], (np.ndarray, torch.Tensor)):
            return TranscribeLoader(_TensorDataset(audio), min(bs, len(audio)), device, shuffle=False, sampler=None)
        raise ValueError(f"Input `audio` is of type {type(audio[0])}. Only `str` (path to audio file), `np.ndarray`, and "
                         "`torch.Tensor` are supported as input.")

    # -------------------------------------------------------------------------------------------------- transcribe
    def _ids_to_text(self, ids, language_id):
        det = getattr(self, "detokenize", None)   # explicit ids -> str hook (data.MultilingualTokenizer.detokenizer(lang))
        if det is not None:
            return det(list(ids))
        if self.tokenizer is not None and language_id in getattr(self.tokenizer, "sp", {}):
            return self.tokenizer.ids_to_text(ids, language_id)
        return " ".join(str(int(i)) for i in ids)     # no SentencePiece model at hand: the token ids themselves

    @torch.no_grad()
    def transcribe(self, audio, batch_size: int = 4, return_hypotheses: bool = False, num_workers: int = 0,
                   verbose: bool = True, logprobs: bool = False, language_id: str = None, channel_selector=None,
                   augmentor=None, override_config=None, **config_kwargs):
        from .model import InternalTranscribeConfig, TranscribeConfig
        if audio is None or (isinstance(audio, (list, tuple)) and len(audio) == 0):
            return {}
        if language_id is None:
            raise ValueError("language_id is required by the multilingual heads (rnnt.py:1624-1640)")
        cfg = override_config or TranscribeConfig(batch_size=batch_size, return_hypotheses=return_hypotheses,
                                                  num_workers=num_workers, channel_selector=channel_selector,
                                                  augmentor=augmentor, verbose=verbose, logprobs=logprobs,
                                                  language_id=language_id)
        if cfg._internal is None:
            cfg._internal = InternalTranscribeConfig()
        # _transcribe_on_begin (:456-496): eval mode, dither and pad_to off; restored by _transcribe_on_end
        f = self.preprocessor.featurizer
        cfg._internal.training_mode, cfg._internal.dither_value, cfg._internal.pad_to_value = self.training, f.dither, f.pad_to
        cfg._internal.device = cfg._internal.device or next(self.parameters()).device
        f.dither, f.pad_to = 0.0, 0
        self.eval()
        hyps, logits_list = [], []
        try:
            loader = self._transcribe_input_processing(audio, cfg, transcripts=None, language_id=language_id)
            for batch in loader:
                sig, sig_len = batch[0], batch[1]
                if not sig.is_cuda and cfg._internal.device.type == "cuda":
                    sig, sig_len = sig.to(cfg._internal.device), sig_len.to(cfg._internal.device)
                langs = [language_id] * sig.shape[0]
                encoded, encoded_len = self.forward(input_signal=sig, input_signal_length=sig_len)
                if self.cur_decoder == "rnnt":
                    ids = self.decode(encoded, encoded_len, langs)
                else:
                    from .decoding import greedy_ctc_decode
                    lp = self.ctc_decoder(encoder_output=encoded, language_ids=langs)
                    if logprobs:   # the reference's deprecated branch (:653-657): per-utterance log-prob matrices
                        lens = encoded_len.tolist()
                        logits_list += [lp[i, :n].cpu() for i, n in enumerate(lens)]
                        continue
                    ids = greedy_ctc_decode(lp, encoded_len)
                hyps += [self._ids_to_text(h, language_id) for h in ids]
        finally:   # _transcribe_on_end (:677-691)
            self.train(mode=cfg._internal.training_mode)
            f.dither, f.pad_to = cfg._internal.dither_value, cfg._internal.pad_to_value
        if self.cur_decoder != "rnnt" and logprobs:
            return logits_list
        return (hyps, list(hyps))
