"""Greedy decoding and WER for the hybrid model.

* greedy_rnnt_decode  -- GreedyBatchedRNNTInfer._greedy_decode_blank_as_pad_loop_frames
  (A/parts/submodules/rnnt_greedy_decoding.py:711-909): frame-synchronous over the batch, at most `max_symbols`
  prediction-net + joint micro-steps per frame, sticky per-frame blank mask, hidden state rolled back for utterances that
  emitted blank.  All tensors stay on the device and the loop makes ONE host read per micro-step (`blank_mask.all()`, as the
  reference does); the encoder and prediction projections of the joint are hoisted out of the loop.  On the MI355X the
  same decode runs DEVICE-RESIDENT (csrc/greedy_decode.hip: one launch, one persistent workgroup per utterance, no host
  read per micro-step) -- greedy_rnnt_decode dispatches to it; the host-driven loop stays as greedy_rnnt_decode_host
  (CPU tensors, and the cross-check of the kernel in tests).
* greedy_ctc_decode   -- GreedyCTCInfer (ctc_greedy_decoding.py:145-229): argmax, collapse repeats, drop blanks.
* word_error_rate / WER -- A/metrics/wer.py:293-360: sum of edit distances over sum of reference lengths.  Hypotheses
  and references are sequences of tokens here; pass `detokenize` (ids -> str) to score words as the reference does
  (its SentencePiece models are not part of the repository).
"""
from typing import Callable, List, Optional, Sequence

import torch


def _edit_distance_py(a: Sequence, b: Sequence) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def _edit_distances(pairs) -> List[int]:
    """Levenshtein distances of (hypothesis, reference) pairs of unit sequences (tokens, words or characters): ONE call of
    the library's host routine (csrc/host_metrics.hip, the reference's `editdistance` C extension: wer.py:58-60) -- the
    pure-Python programme cost more per training step than the decode.  Units are numbered through a dictionary first."""
    pairs = [(list(a), list(b)) for a, b in pairs]
    if not pairs:
        return []
    if sum(len(a) * len(b) for a, b in pairs) < 256:      # tiny: not worth the call
        return [_edit_distance_py(a, b) for a, b in pairs]
    import ctypes
    import numpy as np
    from . import _lib
    L = _lib.lib()
    ids = {}
    fa, fb, oa, ob = [], [], [0], [0]
    for a, b in pairs:
        fa.extend(ids.setdefault(x, len(ids)) for x in a)
        fb.extend(ids.setdefault(x, len(ids)) for x in b)
        oa.append(len(fa)); ob.append(len(fb))
    A, Bv = np.asarray(fa, dtype=np.int32), np.asarray(fb, dtype=np.int32)
    OA, OB = np.asarray(oa, dtype=np.int64), np.asarray(ob, dtype=np.int64)
    out = np.zeros(len(pairs), dtype=np.int64)
    vp = lambda arr: ctypes.c_void_p(arr.ctypes.data)
    _lib.check(L.ia_edit_distance_batch(vp(A), vp(OA), vp(Bv), vp(OB), len(pairs), vp(out)), "ia_edit_distance_batch")
    return out.tolist()


def _edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (editdistance.eval in the reference, wer.py:58-60)."""
    return _edit_distances([(a, b)])[0]


def word_error_rate(hypotheses: List[Sequence], references: List[Sequence], detokenize: Optional[Callable] = None):
    """(wer, total_edits, total_reference_units).  With `detokenize` both sides are turned into strings and split on
    whitespace (words); without it the units are the tokens themselves."""
    pairs = []
    for h, r in zip(hypotheses, references):
        if detokenize is not None:
            h, r = detokenize(list(h)).split(), detokenize(list(r)).split()
        pairs.append((list(h), list(r)))
    words = sum(len(r) for _, r in pairs)
    scores = sum(_edit_distances(pairs))
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


class PendingHyps:
    """Hypotheses whose device work and device->host copies have been ENQUEUED (on the stream that was current at the call);
    `result()` waits for the copies' event and builds the token lists.  training_step(compute_wer=True) holds two of them
    (transducer + CTC) in the monitor, so the decode runs beside the rest of the step instead of in front of it."""

    def __init__(self, finish):
        self._finish, self._res = finish, None

    def result(self) -> List[List[int]]:
        if self._finish is not None:
            self._res = self._finish()
            self._finish = None
        return self._res


def _pinned_async(t):
    """Asynchronous device->host copy of `t` into pinned memory on the current stream."""
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


@torch.no_grad()
def greedy_rnnt_decode(model, encoded, encoded_len, language_ids, max_symbols: Optional[int] = 10) -> List[List[int]]:
    """encoded [B,d,T'] (encoder output), encoded_len [B] -> per-utterance language-local token ids.  A batch that mixes
    languages is decoded language by language (every utterance goes through its own language's head, rnnt.py:1632-1640;
    utterances decode independently of their neighbours)."""
    langs = list(language_ids)
    if len(set(langs)) > 1:
        out = [None] * len(langs)
        for lang in dict.fromkeys(langs):
            idx = [i for i, l in enumerate(langs) if l == lang]
            sel = torch.tensor(idx, device=encoded.device)
            sub = greedy_rnnt_decode(model, encoded.index_select(0, sel), encoded_len.to(encoded.device).index_select(0, sel),
                                     [lang] * len(idx), max_symbols)
            for i, h in zip(idx, sub):
                out[i] = h
        return out
    if encoded.is_cuda and device_decode_supported(model):
        return greedy_rnnt_decode_device(model, encoded, encoded_len, language_ids, max_symbols)
    return greedy_rnnt_decode_host(model, encoded, encoded_len, language_ids, max_symbols)


def device_decode_supported(model) -> bool:
    from . import _lib
    c = model.cfg
    return (c.pred_hidden % 4 == 0 and c.joint_hidden % 4 == 0
            and _lib.lib().ia_greedy_decode_lds_bytes(c.pred_hidden, c.joint_hidden, c.vocab_per_lang + 1) <= 160 * 1024
            and model.decoder.prediction["dec_rnn"].lstm.num_layers == 1)


@torch.no_grad()
def greedy_rnnt_decode_device(model, encoded, encoded_len, language_ids, max_symbols: Optional[int] = 10, defer: bool = False):
    """The same decode in ONE launch (csrc/greedy_decode.hip).  Setup on the HIP fp32 GEMM: the joint's encoder projection of
    all frames and the table EW = W_ih embedding[row] + b_ih + b_hh over the 257 rows the loop can feed the prediction
    network (the language's labels by their ids as decoding.greedy_rnnt_decode_host feeds them, the blank / padding row, the
    zero SOS input); one device->host copy of the token matrix at the end."""
    from . import _lib
    from . import cl
    cl.flush_pending_updates()
    if len(set(language_ids)) != 1:
        raise NotImplementedError("greedy_rnnt_decode: one language per batch (as the CL scripts evaluate)")
    L = _lib.lib()
    dec, joint = model.decoder, model.joint
    dev = encoded.device
    B, d, T = encoded.shape
    V = model.cfg.vocab_per_lang + 1
    blank = V - 1
    Hp, Hj = model.cfg.pred_hidden, model.cfg.joint_hidden
    head = joint.joint_net[-1][language_ids[0]]
    lstm = dec.prediction["dec_rnn"].lstm
    emb = dec.prediction["embed"].weight

    def gemm32(a, w):   # a [M,K] @ w[N,K]^T on csrc/gemm_f32.hip
        a, w = a.float().contiguous(), w.float().contiguous()
        if a.shape[1] % 16 != 0:
            return a @ w.t()
        out = torch.empty(a.shape[0], w.shape[0], dtype=torch.float32, device=dev)
        _lib.check(L.ia_gemm_f32(_lib.ptr(a), a.shape[1], _lib.ptr(w), w.shape[1], a.shape[0], w.shape[0], a.shape[1], _lib.ptr(out),
                                 w.shape[0], _lib.stream_ptr()), "ia_gemm_f32")
        return out

    f_all = (gemm32(encoded.transpose(1, 2).reshape(B * T, d), joint.enc.weight) + joint.enc.bias.float()).view(B, T, Hj).contiguous()
    rows = torch.cat([emb[:blank].float(), emb[dec.blank_idx:dec.blank_idx + 1].float(),
                      torch.zeros(1, Hp, dtype=torch.float32, device=dev)], 0)                      # [V + 1, Hp]
    EW = (gemm32(rows, lstm.weight_ih_l0) + (lstm.bias_ih_l0 + lstm.bias_hh_l0).float()).contiguous()
    ms = int(max_symbols) if max_symbols is not None else 1 << 30
    cap = T * (int(max_symbols) if max_symbols is not None else 8)
    tokens = torch.empty(B, cap, dtype=torch.int32, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    out_len = encoded_len.to(dev).long().contiguous()
    # a model that trains in bf16 decodes on the bf16 images of W_hh / W_pred / the head (what its training step multiplies
    # with, kept current by the fused optimizer: ops/fast.bf16_shadow): half the bytes per emitted symbol of a loop bound by
    # the CU's L2 bandwidth; an fp32 model decodes in fp32
    w16 = model.cfg.compute_dtype == "bf16" and Hp % 8 == 0 and Hj % 8 == 0

    def snap(p, image16=False):   # contiguous image; with `defer` a private COPY: the optimizer may rewrite the weights meanwhile
        if image16:
            from .ops import fast
            t = fast.bf16_shadow(p)
            return t.clone() if defer else t
        t = p.detach().float().contiguous()
        return t.clone() if (defer and t.data_ptr() == p.data_ptr()) else t

    Whh = snap(lstm.weight_hh_l0, w16)
    Wp, bp = snap(joint.pred.weight, w16), snap(joint.pred.bias)
    Wh, bh = snap(head.weight, w16), snap(head.bias)
    args = (_lib.ptr(f_all), _lib.ptr(out_len), _lib.ptr(EW), _lib.ptr(Whh), _lib.ptr(Wp), _lib.ptr(bp),
            _lib.ptr(Wh), _lib.ptr(bh), B, T, Hp, Hj, V, blank, blank, V, ms, _lib.ptr(tokens), cap,
            _lib.ptr(counts), _lib.ptr(overflow))
    scratch = None
    if w16:
        # head over 16 frames at once on the matrix cores, the per-symbol GEMVs split over a cluster of workgroups per utterance
        nw = int(L.ia_greedy_decode_cluster(B, Hp, Hj))
        scratch = torch.empty(int(L.ia_greedy_decode_scratch_bytes(B, Hp, Hj)), dtype=torch.uint8, device=dev) if nw > 1 else None
        st = L.ia_greedy_rnnt_decode_bf16w_ex(*args, nw, _lib.ptr(scratch) if scratch is not None else None,
                                              scratch.numel() if scratch is not None else 0, _lib.stream_ptr())
    else:
        st = L.ia_greedy_rnnt_decode(*args, _lib.stream_ptr())
    _lib.check(st, "ia_greedy_rnnt_decode")

    def check_flags(ovf):
        if ovf & 2:
            raise RuntimeError("greedy_rnnt_decode: a hand-off between the workgroups of an utterance timed out (csrc/greedy_decode.hip); "
                               "the hypotheses of this batch are not valid")
        if ovf & 1:
            raise RuntimeError("greedy_rnnt_decode: an utterance emitted more symbols than the output buffer holds "
                               f"({cap} per utterance); pass a finite max_symbols")
    if defer:
        h_tok, h_n, h_ovf = _pinned_async(tokens), _pinned_async(counts), _pinned_async(overflow)
        ev = torch.cuda.Event()
        ev.record()
        keep = [tokens, counts, overflow, f_all, EW, Whh, Wp, bp, Wh, bh, out_len, scratch]   # alive until the kernel has run

        def finish():
            ev.synchronize()
            keep.clear()
            check_flags(int(h_ovf[0]))
            n = h_n.tolist()
            return [h_tok[b, :n[b]].tolist() for b in range(B)]
        return PendingHyps(finish)
    host_tok, host_n, ovf = tokens.cpu(), counts.cpu().tolist(), int(overflow.item())
    check_flags(ovf)
    return [host_tok[b, :host_n[b]].tolist() for b in range(B)]


@torch.no_grad()
def greedy_rnnt_decode_host(model, encoded, encoded_len, language_ids, max_symbols: Optional[int] = 10) -> List[List[int]]:
    """Host-driven form (one host read per micro-step, as the reference)."""
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
def greedy_ctc_decode(log_probs, lengths, blank: Optional[int] = None, defer: bool = False):
    """log_probs [B,T,V] (language-restricted), lengths [B] -> collapsed token ids (repeats merged, blanks removed).
    defer (CUDA): PendingHyps -- argmax / collapse mask enqueued, one asynchronous copy, lists built on result()."""
    B, T, V = log_probs.shape
    blank = V - 1 if blank is None else blank
    k = log_probs.argmax(-1)                                                  # [B,T]
    valid = torch.arange(T, device=k.device)[None, :] < lengths.to(k.device)[:, None]
    prev = torch.cat([torch.full((B, 1), -1, dtype=k.dtype, device=k.device), k[:, :-1]], 1)
    keep = valid & (k != blank) & (k != prev)
    if defer and k.is_cuda:
        h = _pinned_async(torch.where(keep, k, torch.full_like(k, -1)).to(torch.int32))
        ev = torch.cuda.Event()
        ev.record()

        def finish():
            ev.synchronize()
            arr = h.numpy()
            return [row[row >= 0].tolist() for row in arr]
        return PendingHyps(finish)
    host_k, host_keep = k.tolist(), keep.tolist()
    return [[v for v, m in zip(r, mk) if m] for r, mk in zip(host_k, host_keep)]
