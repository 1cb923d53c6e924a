"""Greedy decoding and WER for the hybrid model.

* greedy_rnnt_decode  -- GreedyBatchedRNNTInfer._greedy_decode_blank_as_pad_loop_frames
  (A/parts/submodules/rnnt_greedy_decoding.py:711-909): frame-synchronous over the batch, at most `max_symbols`
  prediction-net + joint micro-steps per frame, sticky per-frame blank mask, hidden state rolled back for utterances that
  emitted blank.  All tensors stay on the device and the loop makes ONE host read per micro-step (`blank_mask.all()`, as the
  reference does); the encoder and prediction projections of the joint are hoisted out of the loop.  This is the
  host-driven form; a device-resident persistent kernel (no host read per micro-step) is the planned MI355X form.
* greedy_ctc_decode   -- GreedyCTCInfer (ctc_greedy_decoding.py:145-229): argmax, collapse repeats, drop blanks.
* word_error_rate / WER -- A/metrics/wer.py:293-360: sum of edit distances over sum of reference lengths.  Hypotheses
  and references are sequences of tokens here; pass `detokenize` (ids -> str) to score words as the reference does
  (its SentencePiece models are not part of the repository).
"""
from typing import Callable, List, Optional, Sequence

import torch


def _edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (editdistance.eval in the reference, wer.py:58-60)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def word_error_rate(hypotheses: List[Sequence], references: List[Sequence], detokenize: Optional[Callable] = None):
    """(wer, total_edits, total_reference_units).  With `detokenize` both sides are turned into strings and split on
    whitespace (words); without it the units are the tokens themselves."""
    scores = words = 0
    for h, r in zip(hypotheses, references):
        if detokenize is not None:
            h, r = detokenize(list(h)).split(), detokenize(list(r)).split()
        words += len(r)
        scores += _edit_distance(list(h), list(r))
    wer = scores / words if words > 0 else float("inf")
    return wer, scores, words


class WER:
    """update()/compute()/reset() surface of A/metrics/wer.py for the step's monitor."""

    def __init__(self, detokenize: Optional[Callable] = None):
        self.detokenize = detokenize
        self.reset()

    def reset(self):
        self.scores, self.words = 0, 0

    def update(self, hypotheses, references):
        _, s, w = word_error_rate(hypotheses, references, self.detokenize)
        self.scores += s
        self.words += w

    def compute(self):
        wer = self.scores / self.words if self.words > 0 else float("inf")
        return wer, self.scores, self.words


@torch.no_grad()
def greedy_rnnt_decode(model, encoded, encoded_len, language_ids, max_symbols: Optional[int] = 10) -> List[List[int]]:
    """encoded [B,d,T'] (encoder output), encoded_len [B] -> per-utterance language-local token ids."""
    dec, joint = model.decoder, model.joint
    dev = encoded.device
    B = encoded.shape[0]
    V = model.cfg.vocab_per_lang + 1
    blank = V - 1
    if len(set(language_ids)) != 1:
        raise NotImplementedError("greedy_rnnt_decode: one language per batch (as the CL scripts evaluate)")
    head = joint.joint_net[-1][language_ids[0]]
    f_all = joint.project_encoder(encoded.transpose(1, 2).float())          # [B,T,H] hoisted out of the loop
    out_len = encoded_len.to(dev)
    T = int(out_len.max().item())
    last_label = torch.full((B, 1), blank, dtype=torch.long, device=dev)
    hidden = None
    tokens = torch.full((B, 0), -1, dtype=torch.long, device=dev)
    cols = []
    for t in range(T):
        f = f_all[:, t]                                                      # [B,H]
        blank_mask = t >= out_len
        symbols = 0
        while max_symbols is None or symbols < max_symbols:
            if hidden is None and t == 0 and symbols == 0:
                g, hidden_prime = dec.predict(None, None, add_sos=False, batch_size=B)      # SOS = zero embedding
            else:
                y = torch.where(last_label == blank, torch.full_like(last_label, dec.blank_idx), last_label)
                g, hidden_prime = dec.predict(y, hidden, add_sos=False, batch_size=B)
            logits = head(torch.relu(f + joint.project_prednet(g[:, 0].float())))            # [B,V]
            k = logits.float().argmax(-1)
            blank_mask = blank_mask | (k == blank)
            if bool(blank_mask.all()):
                break
            if hidden is not None:   # utterances that emitted blank keep their previous state
                keep = blank_mask.view(1, B, 1)
                hidden_prime = tuple(torch.where(keep, h_old, h_new) for h_old, h_new in zip(hidden, hidden_prime))
            else:
                keep = blank_mask.view(1, B, 1)
                hidden_prime = tuple(torch.where(keep, torch.zeros_like(h_new), h_new) for h_new in hidden_prime)
            k = torch.where(blank_mask, last_label[:, 0], k)
            cols.append(torch.where(blank_mask, torch.full_like(k, -1), k))
            last_label = k.view(B, 1)
            hidden = hidden_prime
            symbols += 1
    if cols:
        tokens = torch.stack(cols, 1)
    host = tokens.tolist()
    return [[v for v in row if v >= 0] for row in host]


@torch.no_grad()
def greedy_ctc_decode(log_probs, lengths, blank: Optional[int] = None) -> List[List[int]]:
    """log_probs [B,T,V] (language-restricted), lengths [B] -> collapsed token ids (repeats merged, blanks removed)."""
    B, T, V = log_probs.shape
    blank = V - 1 if blank is None else blank
    k = log_probs.argmax(-1)                                                  # [B,T]
    valid = torch.arange(T, device=k.device)[None, :] < lengths.to(k.device)[:, None]
    prev = torch.cat([torch.full((B, 1), -1, dtype=k.dtype, device=k.device), k[:, :-1]], 1)
    keep = valid & (k != blank) & (k != prev)
    host_k, host_keep = k.tolist(), keep.tolist()
    return [[v for v, m in zip(r, mk) if m] for r, mk in zip(host_k, host_keep)]
