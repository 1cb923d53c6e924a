"""WER metric objects of the hybrid model (`model.wer`, `model.ctc_wer`) with the reference's surface
(A/metrics/wer.py:225-360): `update(predictions, predictions_lengths, targets, targets_lengths, ..., lang_ids)`,
`compute() -> (wer, scores, words)` as 0-dim float tensors, `reset()`, `log_prediction`, `use_cer`.

What the reference does and this keeps:
* references and hypotheses are STRINGS (token ids through the language's SentencePiece model,
  `decode_tokens_to_str(target, lang)`), split on whitespace into words (characters with `use_cer`), scored with the
  Levenshtein distance (`editdistance.eval`, wer.py:346-357);
* `update` REPLACES the state with this call's sums (wer.py:359-360 assigns, it does not accumulate);
* `compute` sums (scores, words) over the ranks of the default process group first (torchmetrics' sync-on-compute: the
  fused joint calls it once per sub-batch, A/modules/rnnt.py:1535-1537 "Sync and all_reduce on all processes").  Here the
  exchange is ONE all-reduce of an int64 pair (or of an [n, 2] table when the step scores several sub-batches at once).

Without a tokenizer on the model (`model.set_tokenizer`) a token id stands for a word (ids joined by spaces), so the
rates are token-level; the SentencePiece models are not part of either repository.
"""
import logging
import weakref
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn as nn

from .decoding import _edit_distance, _edit_distances


def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _sums_device(default):
    """Where the (edits, units) sums of a host-scored update live.  They are born on the host; they only have to be on the
    device for an RCCL exchange.  Everywhere else (one process, gloo) they stay host tensors: `torch.tensor(x, device=cuda)` is a
    synchronous pageable copy on the compute stream, i.e. a wait for everything queued on it -- the whole training step, every
    time the step's monitor is read."""
    dist = _dist_world()
    if dist is not None and dist.get_backend() == "nccl":
        return default
    return torch.device("cpu")


class WER(nn.Module):
    """`kind` = "rnnt": predictions are encoder outputs [B, D, T] decoded greedily through the prediction network and the
    joint (rnnt_decoder_predictions_tensor); "ctc": predictions are the CTC head's log-probs [B, T, V]
    (ctc_decoder_predictions_tensor, repeats folded)."""

    def __init__(self, model, kind: str, use_cer: bool = False, log_prediction: bool = True, dist_sync_on_step: bool = True):
        super().__init__()
        if kind not in ("rnnt", "ctc"):
            raise TypeError(f"WER metric does not support decoding of type {kind}")
        self.bind(model)
        self.kind = kind
        self.use_cer = use_cer
        self.log_prediction = log_prediction
        self.dist_sync_on_step = dist_sync_on_step
        self.register_buffer("scores", torch.tensor(0, dtype=torch.int64), persistent=False)
        self.register_buffer("words", torch.tensor(0, dtype=torch.int64), persistent=False)

    def bind(self, model):
        """Point the metric at the model that owns it (a weak reference, not a sub-module: the model owns the metric, not
        vice versa).  The model re-binds before every use, so deep copies of a model score with their own weights."""
        object.__setattr__(self, "_model", weakref.ref(model))
        return self

    def _device(self):
        m = self._model()
        p = next(m.parameters(), None) if m is not None else None
        return p.device if p is not None else self.scores.device

    # ------------------------------------------------------------------ strings
    def decode_tokens_to_str(self, tokens: Sequence[int], lang: Optional[str] = None) -> str:
        m = self._model()
        return m._ids_to_text(list(tokens), lang)

    def _units(self, s: str) -> List[str]:
        return list(s) if self.use_cer else s.split()

    def score(self, hypotheses: List[str], references: List[str]):
        """(sum of edit distances, number of reference units) of string pairs."""
        pairs = [(self._units(h), self._units(r)) for h, r in zip(hypotheses, references)]
        return sum(_edit_distances(pairs)), sum(len(r) for _, r in pairs)

    def hypotheses(self, predictions, predictions_lengths, lang_ids) -> List[List[int]]:
        m = self._model()
        if self.kind == "rnnt":
            return m.decode(predictions.detach(), predictions_lengths, lang_ids)
        from .decoding import greedy_ctc_decode
        return greedy_ctc_decode(predictions.detach(), predictions_lengths)

    # ------------------------------------------------------------------ metric surface
    @torch.no_grad()
    def update(self, predictions, predictions_lengths, targets, targets_lengths, predictions_mask=None, input_ids=None,
               lang_ids: Optional[List[str]] = None):
        lens = targets_lengths.long().tolist()
        rows = targets.long().tolist()
        B = len(rows)
        langs = lang_ids if lang_ids is not None else [None] * B
        references = [self.decode_tokens_to_str(rows[i][:lens[i]], langs[i]) for i in range(B)]
        ids = self.hypotheses(predictions, predictions_lengths, lang_ids)
        hyps = [self.decode_tokens_to_str(h, langs[i]) for i, h in enumerate(ids)]
        if self.log_prediction:
            logging.info("\n")
            logging.info(f"reference:{references[0]}")
            logging.info(f"predicted:{hyps[0]}")
        s, w = self.score(hyps, references)
        dev = _sums_device(self._device())
        self.scores = torch.tensor(s, device=dev, dtype=self.scores.dtype)
        self.words = torch.tensor(w, device=dev, dtype=self.words.dtype)

    @torch.no_grad()
    def update_from_ids(self, hyp_ids: List[List[int]], ref_ids: List[List[int]], lang_ids: Optional[List[str]] = None):
        """`update` with the hypotheses' and references' token ids already on the host (the deferred in-step path)."""
        B = len(ref_ids)
        langs = lang_ids if lang_ids is not None else [None] * B
        references = [self.decode_tokens_to_str(r, langs[i]) for i, r in enumerate(ref_ids)]
        hyps = [self.decode_tokens_to_str(h, langs[i]) for i, h in enumerate(hyp_ids)]
        if self.log_prediction and B:
            logging.info("\n")
            logging.info(f"reference:{references[0]}")
            logging.info(f"predicted:{hyps[0]}")
        s_, w_ = self.score(hyps, references)
        dev = _sums_device(self._device())
        self.scores = torch.tensor(s_, device=dev, dtype=self.scores.dtype)
        self.words = torch.tensor(w_, device=dev, dtype=self.words.dtype)

    def compute(self):
        pair = torch.stack([self.scores.detach(), self.words.detach()])
        dist = _dist_world() if self.dist_sync_on_step else None
        if dist is not None:
            dist.all_reduce(pair)
        scores, words = pair[0].float(), pair[1].float()
        return scores / words, scores, words

    def reset(self):
        self.scores.zero_()
        self.words.zero_()

    # ------------------------------------------------------------------ the step's form: several groups, one exchange
    @torch.no_grad()
    def grouped(self, hyp_ids: List[List[int]], ref_ids: List[List[int]], lang_ids, group_size: int):
        """The fused joint's per-sub-batch update / compute / reset loop (A/modules/rnnt.py:1511-1542) from ONE decode of the
        whole batch: (mean over the groups of the groups' rates, sum of scores, sum of words) as float tensors; the groups'
        (scores, words) pairs are summed over the ranks in ONE all-reduce."""
        B = len(ref_ids)
        langs = lang_ids if lang_ids is not None else [None] * B
        refs = [self.decode_tokens_to_str(r, langs[i]) for i, r in enumerate(ref_ids)]
        hyps = [self.decode_tokens_to_str(h, langs[i]) for i, h in enumerate(hyp_ids)]
        if self.log_prediction and B:
            logging.info("\n")
            logging.info(f"reference:{refs[0]}")
            logging.info(f"predicted:{hyps[0]}")
        table = [list(self.score(hyps[b0:b0 + group_size], refs[b0:b0 + group_size])) for b0 in range(0, B, group_size)]
        t = torch.tensor(table, dtype=torch.int64, device=_sums_device(self._device()))
        dist = _dist_world() if self.dist_sync_on_step else None
        if dist is not None:
            dist.all_reduce(t)
        tf = t.float()
        return (tf[:, 0] / tf[:, 1]).mean(), tf[:, 0].sum(), tf[:, 1].sum()
