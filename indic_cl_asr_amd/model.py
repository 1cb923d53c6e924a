"""EncDecHybridRNNTCTCModel for MI355X: the reference's Python surface for the hot path
(A/models/hybrid_rnnt_ctc_models.py:859-930 training_step, A/models/rnnt_models.py:606-655 forward,
A/models/hybrid_rnnt_ctc_bpe_models.py:43-170 construction) so R/cl_baseline{,_ewc,_mas,_lwf}.py keep working
on `model.module.training_step(batch, [lang]*B)` and the MAS/LwF stash flags.

What is deliberately NOT reproduced from the reference step: 6x gc.collect()+empty_cache(),
4x .item() (one batched D2H read instead), per-sub-batch .max() syncs, CUDA_LAUNCH_BLOCKING.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import os

import torch
import torch.nn as nn

from .config import ModelConfig, model_config
from .decoder import ConvASRDecoder, RNNTDecoder, RNNTJoint
from .encoder import ConformerEncoder, subsampled_length
from .features import AudioToMelSpectrogramPreprocessor, SpectrogramAugmentation, mel_frame_count
from .losses.ctc import CTCLoss
from .losses.rnnt import RNNTLoss
from .transcribe import TranscriptionMixin, lookup_host_lengths


@dataclass
class InternalTranscribeConfig:  # hybrid_rnnt_ctc_models.py:98-111 (importable name kept for the CL scripts)
    device: Optional[torch.device] = None
    dtype: Optional[torch.dtype] = None
    training_mode: bool = False
    logging_level: Optional[int] = None
    dither_value: float = 0.0
    pad_to_value: int = 0
    temp_dir: Optional[str] = None


@dataclass
class TranscribeConfig:  # :114-127
    batch_size: int = 4
    return_hypotheses: bool = False
    num_workers: Optional[int] = None
    channel_selector: Optional[int] = None
    augmentor: Optional[dict] = None
    verbose: bool = True
    partial_hypothesis: Optional[List] = None
    logprobs: bool = False                  # fields of the reference's fork (R/cl_baseline.py:162-170 passes both)
    language_id: Optional[str] = None
    _internal: Optional[InternalTranscribeConfig] = None


_SIDE_STREAMS = {}  # one side HIP stream per device (process-wide: models stay deep-copyable)


class StepMonitor(dict):
    """The step's monitor dict (hybrid_rnnt_ctc_models.py:899-913 fills it with `.item()` floats).  The loss values stay
    on the device until somebody reads them: the D2H copy is started asynchronously inside training_step and resolved on
    first access, so the step itself contains no device->host synchronisation (the CL loops read the monitor only after
    backward + optimizer step, R/cl_baseline_ewc.py:258-260)."""

    def __init__(self, static, keys=(), device_values=None):
        super().__init__(static)
        self._pending = None
        self._deferred = []   # callables that fill further keys on first access (the in-step WERs: decode enqueued, scored on read)
        if device_values is not None and device_values.is_cuda:
            host = torch.empty(device_values.shape, dtype=device_values.dtype, pin_memory=True)
            host.copy_(device_values, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (tuple(keys), host, ev)
            for k in keys:
                if not k.startswith('_'):
                    super().__setitem__(k, None)
        elif device_values is not None:
            for k, v in zip(keys, device_values.tolist()):
                super().__setitem__(k, v)

    def defer(self, fn):
        """fn(monitor_dict_setter) runs once, on the first read of any key."""
        self._deferred.append(fn)

    def _resolve(self):
        if self._deferred:
            fns, self._deferred = self._deferred, []
            for fn in fns:
                for k, v in fn().items():
                    super().__setitem__(k, v)
        if self._pending is not None:
            keys, host, ev = self._pending
            self._pending = None
            ev.synchronize()
            for k, v in zip(keys, host.tolist()):
                if k == '_lstm_timeout':
                    if v != 0:
                        from .ops import lstm as hip_lstm
                        hip_lstm.raise_if_timed_out()
                        raise RuntimeError("persistent LSTM: a workgroup hand-off timed out during this step")
                    continue
                if super().__getitem__(k) is None:
                    super().__setitem__(k, v)

    def __getitem__(self, k):
        self._resolve()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._resolve()
        return super().get(k, default)

    def items(self):
        self._resolve()
        return super().items()

    def values(self):
        self._resolve()
        return super().values()

    def __repr__(self):
        self._resolve()
        return super().__repr__()


class EncDecHybridRNNTCTCModel(TranscriptionMixin, nn.Module):
    def __init__(self, cfg: Optional[ModelConfig] = None, **kw):
        super().__init__()
        self.cfg = cfg = cfg or model_config(**kw)
        self.preprocessor = AudioToMelSpectrogramPreprocessor(cfg)
        self.spec_augmentation = SpectrogramAugmentation(cfg) if (cfg.freq_masks + cfg.time_masks) > 0 else None
        self.encoder = ConformerEncoder(cfg)
        self.decoder = RNNTDecoder(cfg)
        self.loss = RNNTLoss(num_classes=cfg.vocab_per_lang, reduction='mean_batch',
                             loss_kwargs=dict(fastemit_lambda=cfg.fastemit_lambda, clamp=cfg.clamp))
        self.joint = RNNTJoint(cfg, loss=self.loss)
        self.ctc_decoder = ConvASRDecoder(cfg)
        self.ctc_loss = CTCLoss(num_classes=cfg.vocab_per_lang, zero_infinity=True, reduction='mean_batch')
        self.ctc_loss_weight = cfg.ctc_loss_weight
        # metric objects with the reference's update / compute / reset surface (A/metrics/wer.py; hybrid_rnnt_ctc_bpe_models.py
        # :126-149 builds them with log_prediction from the config, the CL scripts switch it off: R/cl_baseline.py:127-128)
        from .metrics import WER
        self.wer = WER(self, "rnnt", log_prediction=True)
        self.ctc_wer = WER(self, "ctc", log_prediction=True)
        # the reference decodes EVERY training batch for its monitor (compute_wer = True, hybrid_rnnt_ctc_models.py:875), and so
        # does training_step by default; this switch is for callers that cannot pass the argument: None -> on, True / False as set
        # (an explicit training_step(compute_wer=...) wins)
        self.compute_wer_in_step = None
        self.cur_decoder = "rnnt"
        self.language_masks = self.ctc_decoder.language_masks
        self._step = 0
        self.seed = 1234
        self.spec_augment_enabled = True
        self.dither_enabled = True
        self.overlap_decoder = True      # prediction network on a side HIP stream (training_step)
        self.overlap_ctc = True          # CTC head + CTC loss on a second side stream, under the joint (training_step)
        self.defer_wer = True            # compute_wer: decode on a third side stream, scored when the monitor is read

    @staticmethod
    def _side_stream(device, which=0):
        key = (device.index if device.index is not None else torch.cuda.current_device(), which)
        st = _SIDE_STREAMS.get(key)
        if st is None:
            st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return st

    def disable_dropout(self):
        """Deterministic numerics runs (parity tests): p=0 everywhere, module tree / parameter names unchanged."""
        for mod in self.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, "dropout_rate"):
                mod.dropout_rate = 0.0
        return self

    # ------------------------------------------------------------------ forward (rnnt_models.py:606-655)
    def forward(self, input_signal=None, input_signal_length=None, processed_signal=None,
                processed_signal_length=None, language_ids=None):
        has_input = input_signal is not None and input_signal_length is not None
        has_proc = processed_signal is not None and processed_signal_length is not None
        if (has_input ^ has_proc) is False:
            raise ValueError(f"{self} Arguments ``input_signal`` and ``input_signal_length`` are mutually exclusive "
                             " with ``processed_signal`` and ``processed_signal_len`` arguments.")
        sub_len = None
        if not has_proc:
            spans = None
            flen_dev = None
            host_len, self._host_signal_len = getattr(self, "_host_signal_len", None), None   # (one-shot hint of training_step)
            if host_len is not None and len(host_len) != input_signal.shape[0]:
                host_len = None
            if host_len is not None and input_signal.is_cuda:
                # the lengths are known on the host (training_step): frame counts before / after subsampling by the integer
                # rule, one pinned asynchronous copy instead of ~12 tiny integer kernels
                fl = [mel_frame_count(int(n), self.cfg.n_fft, self.cfg.n_window_stride) for n in host_len]
                both = torch.tensor([fl, [subsampled_length(n) for n in fl]], dtype=torch.int64).pin_memory()
                both = both.to(input_signal.device, non_blocking=True)
                flen_dev, sub_len = both[0], both[1]
            if self.spec_augmentation is not None and self.training and self.spec_augment_enabled:
                if host_len is not None:
                    # training_step knows the lengths on the host: spans from a CPU generator, one asynchronous copy
                    g = torch.Generator()
                    g.manual_seed(self.seed * 1000003 + self._step)
                    flen_h = [mel_frame_count(int(n), self.cfg.n_fft, self.cfg.n_window_stride) for n in host_len]
                    spans = self.spec_augmentation.draw_host(flen_h, self.cfg.feat_in, input_signal.device, g)
                else:
                    flen = self.preprocessor.featurizer.get_seq_len(input_signal_length)
                    g = torch.Generator(device=input_signal.device)
                    g.manual_seed(self.seed * 1000003 + self._step)
                    spans = self.spec_augmentation.draw(flen, self.cfg.feat_in, g)
            processed_signal, processed_signal_length = self.preprocessor(
                input_signal=input_signal, length=input_signal_length, spec_aug=spans, dither=self.dither_enabled,
                seed=self.seed * 7919 + self._step, seq_len=flen_dev)
        elif self.spec_augmentation is not None and self.training and self.spec_augment_enabled:
            processed_signal = self.spec_augmentation(input_spec=processed_signal, length=processed_signal_length)
        self.encoder.fast_seed = (self.seed * 31 + self._step) & 0x7FFFFFFF
        encoded, encoded_len = self.encoder(audio_signal=processed_signal, length=processed_signal_length, subsampled_len=sub_len)
        return encoded, encoded_len

    # ------------------------------------------------------------------ greedy decoding / WER
    @torch.no_grad()
    def decode(self, encoded, encoded_len, language_ids, max_symbols=10):
        """Greedy transducer hypotheses (language-local token ids) with the prediction net and joint frozen in eval mode
        (rnnt_decoding.py: `with self.decoder.as_frozen(), self.joint.as_frozen()`)."""
        from .decoding import greedy_rnnt_decode
        modes = (self.decoder.training, self.joint.training)
        self.decoder.eval(); self.joint.eval()
        try:
            return greedy_rnnt_decode(self, encoded, encoded_len, language_ids, max_symbols)
        finally:
            self.decoder.train(modes[0]); self.joint.train(modes[1])

    @torch.no_grad()
    def batch_wer(self, encoded, encoded_len, log_probs, transcript, transcript_len, language_ids):
        """The step's two rates as the reference forms them: `training_batch_wer` = mean over the fused joint's sub-batches of
        each sub-batch's (cross-rank) WER (A/modules/rnnt.py:1511-1553), a 0-dim tensor; `training_batch_wer_ctc` = the CTC
        head's WER over the whole batch through ctc_wer.update / compute / reset (hybrid_rnnt_ctc_models.py:903-912), a float.
        ONE device-resident greedy decode serves all sub-batches (utterances decode independently of their neighbours)."""
        self.wer.bind(self); self.ctc_wer.bind(self)
        lens = transcript_len.tolist()
        refs = [row[:n] for row, n in zip(transcript.tolist(), lens)]
        hyps = self.decode(encoded.detach(), encoded_len, language_ids)
        wer, _, _ = self.wer.grouped(hyps, refs, language_ids, max(1, int(self.joint.fused_batch_size)))
        self.ctc_wer.update(predictions=log_probs, predictions_lengths=encoded_len, targets=transcript,
                            targets_lengths=transcript_len, lang_ids=language_ids)
        ctc_wer, _, _ = self.ctc_wer.compute()
        self.ctc_wer.reset()
        return wer, ctc_wer.item()

    def wer_decode_begin(self, encoded, encoded_len, transcript, language_ids, enc_ready, stream):
        """First half of the deferred in-step WER (training_step, compute_wer): right behind the encoder output `enc_ready` the
        greedy transducer decode (one persistent workgroup per utterance for ~8 ms at 32 x 15 s) and the copy of the references
        are ENQUEUED on `stream`.  The decode reads private copies of the weights it needs, so the optimizer may update them
        meanwhile; the returned event marks the point behind which those copies (and the setup GEMMs) have been issued."""
        from .decoding import _pinned_async, greedy_rnnt_decode_device
        stream.wait_event(enc_ready)
        with torch.cuda.stream(stream):
            refs_h = _pinned_async(transcript)
            hyps_p = greedy_rnnt_decode_device(self, encoded.detach(), encoded_len, language_ids, 10, defer=True)
            snap = torch.cuda.Event()
            snap.record(stream)
        for t in (encoded, encoded_len, transcript):
            t.record_stream(stream)
        return {"refs": refs_h, "hyps": hyps_p, "snap": snap, "stream": stream}

    def wer_decode_finish(self, pend, ctc_logits, encoded_len, language_ids, host_tgt_lens, ctc_ready):
        """Second half: the CTC head's greedy path (argmax, collapse mask, one copy) joins the same stream behind `ctc_ready`;
        returns fn() -> {'training_batch_wer', 'training_batch_wer_ctc'}, which waits for the copies and scores on the host --
        the monitor calls it on its first read (the CL loops read it after the optimizer step, R/cl_baseline_ewc.py:258-260)."""
        from .decoding import greedy_ctc_decode
        self.wer.bind(self); self.ctc_wer.bind(self)
        stream = pend["stream"]
        stream.wait_event(ctc_ready)
        with torch.cuda.stream(stream):
            ctc_p = greedy_ctc_decode(ctc_logits.detach(), encoded_len, defer=True)
            done = torch.cuda.Event()
            done.record(stream)
        ctc_logits.record_stream(stream)
        lens = [int(n) for n in host_tgt_lens]
        group = max(1, int(self.joint.fused_batch_size))
        refs_h, hyps_p = pend["refs"], pend["hyps"]

        def fn():
            done.synchronize()
            refs = [row[:n] for row, n in zip(refs_h.tolist(), lens)]
            # scored under the decode's stream: whatever device work the sums need (an RCCL exchange of the (edits, units) pairs
            # over several ranks -- nothing in one process) then orders behind the decode only, not behind the training step
            # queued on the compute stream
            with torch.cuda.stream(stream):
                wer, _, _ = self.wer.grouped(hyps_p.result(), refs, language_ids, group)
                self.ctc_wer.update_from_ids(ctc_p.result(), refs, language_ids)
                ctc_wer, _, _ = self.ctc_wer.compute()
                self.ctc_wer.reset()
                ctc_wer = ctc_wer.item()
                if torch.is_tensor(wer) and wer.is_cuda:
                    stream.synchronize()         # the caller reads it from another stream
            return {'training_batch_wer': wer, 'training_batch_wer_ctc': ctc_wer}
        return fn

    # ------------------------------------------------------------------ training_step (:859-930)
    def training_step(self, batch, lang_ids, return_probs=False, host_lengths=None, compute_wer=None):
        """batch = (signal [B,L] f32, signal_len [B] i64, transcript [B,U] i64, transcript_len [B] i64), all on the
        device.  `host_lengths` = (signal_len list, transcript_len list): optional host copies (the collate
        function has them) that remove the only device->host read the sub-batch loop needs.
        `compute_wer`: the reference decodes every training batch greedily for its monitor (compute_wer = True hard-wired,
        hybrid_rnnt_ctc_models.py:877-912: one host-driven micro-step loop per frame) -- so does this step unless told otherwise
        (compute_wer=False here, or `model.compute_wer_in_step = False` for callers that cannot pass it: the monitor then carries
        NaN).  Word-level through `model.set_tokenizer`, token-level without one (decoding.py)."""
        signal, signal_len, transcript, transcript_len = batch
        language_ids = lang_ids
        if host_lengths is None:
            # batches from model._transcribe_input_processing (the loader the CL scripts use) carry their host-side lengths
            # in a registry keyed by the device tensors: the reference's own call -- training_step(batch, lang_ids) -- then
            # runs without a device->host read (which would wait for everything already queued: the previous step)
            h_sig, h_tgt = lookup_host_lengths(signal_len), lookup_host_lengths(transcript_len)
            if h_sig is not None and h_tgt is not None and len(h_sig) == signal.shape[0]:
                host_lengths = (h_sig, h_tgt)
        if host_lengths is None:
            both = torch.stack([signal_len.long(), transcript_len.long()]).tolist()  # one D2H read
            host_lengths = (both[0], both[1])
        h_sig, h_tgt = host_lengths
        h_enc = [subsampled_length(mel_frame_count(int(n), self.cfg.n_fft, self.cfg.n_window_stride)) for n in h_sig]

        # The prediction network only meets the encoder in the joint: its persistent LSTM kernel occupies 40 of the 256
        # CUs, so it runs on a side stream under the encoder.  It is ISSUED after the encoder on purpose: autograd runs
        # ready nodes in reverse creation order, so its backward (replayed on the same side stream) is then enqueued
        # before the encoder blocks' backward and overlaps it even when the host runs barely ahead of the GPU (profilers,
        # multi-rank runs), instead of trailing the step on an idle GPU; the side stream only waits for the event recorded before the encoder was enqueued, not for the encoder itself.
        side = self._side_stream(signal.device) if self.overlap_decoder and signal.is_cuda else None
        if side is not None:
            main = torch.cuda.current_stream(signal.device)
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(main)
        self._host_signal_len = h_sig   # lets forward() draw the SpecAugment spans on the host
        encoded, encoded_len = self.forward(input_signal=signal, input_signal_length=signal_len)
        if side is not None:
            # ... and it starts BEHIND the front end and the subsampling, not beside them: those are chip-filling kernels
            # (0.5 GB write, a 284 GFLOP implicit GEMM) that lose a sixth of their speed to the 40 CUs of the persistent LSTM;
            # under the blocks' latency-bound launches the CUs are cheaper (3 x 150 steps, one box: 10.31 -> 10.26 ms;
            # IA_DELAY_DECODER=0 restores the start at step begin)
            from .encoder import SUBSAMPLED_EVENT
            ev_sub = SUBSAMPLED_EVENT.get(signal.device.index)
            side.wait_event(ev_sub if (ev_sub is not None and os.environ.get("IA_DELAY_DECODER", "1") != "0") else inputs_ready)
            from . import cl
            if cl.LAST_UPDATE_EVENT is not None:   # a deferred optimizer update was applied on the main stream inside forward
                side.wait_event(cl.LAST_UPDATE_EVENT)
            with torch.cuda.stream(side):
                decoder, target_length, states = self.decoder(targets=transcript, target_length=transcript_len)
            for t in (transcript, transcript_len):
                t.record_stream(side)
            main.wait_stream(side)
            decoder.record_stream(main)
        else:
            decoder, target_length, states = self.decoder(targets=transcript, target_length=transcript_len)
        # The CTC branch (257-column head, log-softmax, a latency-bound lattice kernel with one workgroup per utterance) only
        # needs the encoder output: it runs on a second side stream under the joint, which fills the GPU on its own.  Issued
        # BEFORE the joint: autograd then replays its backward (on the same side stream) after the joint's backward has
        # been enqueued, i.e. concurrently with it; a stream of its own, so that it never queues behind the LSTM backward.
        side2 = self._side_stream(signal.device, 1) if self.overlap_ctc and signal.is_cuda else None
        if signal.is_cuda and self.cfg.compute_dtype == "bf16":
            # ONE bf16 image of the encoder output for both heads, cast on the main stream before the CTC branch forks off
            from .ops import tail
            enc_btd = encoded.transpose(1, 2)
            if enc_btd.is_contiguous() and enc_btd.dtype == torch.float32:
                tail.share_bf16(enc_btd)      # (withdrawn below, once both heads have been issued)
        want_wer = bool(compute_wer if compute_wer is not None else
                        (self.compute_wer_in_step if self.compute_wer_in_step is not None else True))
        # the CTC head and loss as one node on raw logits (no log-prob tensor, no softmax backward) unless a caller needs the
        # log-probs themselves with a gradient path (LwF: return_probs) or the raw logits stash (MAS: return_logits_)
        ctc_fused = (signal.is_cuda and not return_probs and self.ctc_loss.config_reduction == 'mean_batch'
                     and self.ctc_decoder.fused_loss_supported(encoded, language_ids, transcript))
        ctc_keep = {} if (ctc_fused and want_wer) else None

        def ctc_branch():
            if ctc_fused:
                return None, self.ctc_decoder.forward_loss(encoded, language_ids, transcript, encoded_len, transcript_len,
                                                           zero_infinity=self.ctc_loss.zero_infinity, keep=ctc_keep)
            lp = self.ctc_decoder(encoder_output=encoded, language_ids=language_ids)
            return lp, self.ctc_loss(log_probs=lp, targets=transcript, input_lengths=encoded_len, target_lengths=transcript_len)

        if side2 is not None:
            main = torch.cuda.current_stream(signal.device)
            enc_ready = torch.cuda.Event()
            enc_ready.record(main)
            side2.wait_event(enc_ready)
            with torch.cuda.stream(side2):
                log_probs, ctc_loss = ctc_branch()
            for t in (encoded, encoded_len, transcript, transcript_len):
                t.record_stream(side2)
        wer_pend = None
        if want_wer and signal.is_cuda and self.defer_wer and len(set(language_ids)) == 1:
            from .decoding import device_decode_supported
            if device_decode_supported(self):
                from . import cl as _cl
                _cl.flush_pending_updates()   # (on the compute stream, in front of the event: the decode asks for it too, under its own stream -- a no-op then)
                main_ = torch.cuda.current_stream(signal.device)
                ready_ = torch.cuda.Event()
                ready_.record(main_)
                wer_pend = self.wer_decode_begin(encoded, encoded_len, transcript, language_ids, ready_, self._side_stream(signal.device, 2))
        self.joint.loss_scale_hint = (1.0 - self.ctc_loss_weight) / max(1, signal.shape[0])
        self.joint.dropout_seed = (self.seed * 2654435761 + self._step * 40503) & 0x7FFFFFFF
        self.joint.return_costs = bool(signal.is_cuda)     # per-utterance costs: the combination kernel forms the means
        loss_value, wer, _, _ = self.joint(encoder_outputs=encoded, decoder_outputs=decoder, encoder_lengths=encoded_len,
                                           transcripts=transcript, transcript_lengths=transcript_len, compute_wer=False,
                                           language_ids=language_ids, host_lengths=(h_enc, h_tgt))
        self.joint.return_costs = False
        if side2 is not None:
            main.wait_stream(side2)
            ctc_loss.record_stream(main)
            if log_probs is not None:
                log_probs.record_stream(main)
        else:
            log_probs, ctc_loss = ctc_branch()
        if signal.is_cuda:
            from .ops import tail
            tail.unshare(signal.device)
        costs = getattr(self.joint, "last_costs", None)
        self.joint.last_costs = None
        if signal.is_cuda and ctc_fused and costs is not None and self.loss.reduction == 'mean_batch':
            # loss = (1-w) mean(costs) + w mean(nll), the monitor's three values and the persistent LSTM's timeout flag: one launch
            from .ops import lstm as hip_lstm
            from .ops import tail
            loss_value, vals = tail.loss_combine(costs, ctc_loss, self.ctc_loss_weight, hip_lstm.timeout_words(signal.device))
            keys = ('train_rnnt_loss', 'train_ctc_loss', 'train_loss', '_lstm_timeout')
            monitor = StepMonitor({'training_batch_wer': torch.tensor(float('nan')), 'training_batch_wer_ctc': float('nan')}, keys, vals)
        else:
            if costs is not None:
                loss_value = self.loss.reduce([costs], [transcript_len])
            if ctc_fused:      # per-utterance nll from the fused head + loss node: the wrapper's 'mean_batch'
                ctc_loss = ctc_loss.mean()
            rnnt_only = loss_value
            loss_value = (1 - self.ctc_loss_weight) * loss_value + self.ctc_loss_weight * ctc_loss
            vals = [rnnt_only.detach().float(), ctc_loss.detach().float(), loss_value.detach().float()]
            keys = ['train_rnnt_loss', 'train_ctc_loss', 'train_loss']
            if signal.is_cuda:
                # a lost hand-off of the persistent LSTM (bounded spin) travels to the host with the loss values: the monitor
                # raises when it is read instead of training on a silently wrong prediction network (csrc/lstm.hip)
                from .ops import lstm as hip_lstm
                flag = hip_lstm.timeout_flags(signal.device)
                if flag is not None:
                    vals.append(flag); keys.append('_lstm_timeout')
            monitor = StepMonitor({'training_batch_wer': torch.tensor(float('nan')), 'training_batch_wer_ctc': float('nan')},
                                  tuple(keys), torch.stack(vals))
        if want_wer:
            if log_probs is None:   # greedy CTC reads the raw logits: argmax over the valid columns == argmax of the log-probs
                log_probs = ctc_keep["logits"][:, :, :ctc_keep["V"]]
            if wer_pend is not None and isinstance(monitor, StepMonitor):
                # decode + scoring off the step's critical path: the transducer decode was enqueued on a third side stream right
                # behind the encoder output; the CTC argmax joins it here, the scores are formed when the monitor is read
                main = torch.cuda.current_stream(signal.device)
                ctc_ready = torch.cuda.Event()
                ctc_ready.record(main)      # (side2 -- the CTC branch -- was joined into the main stream above)
                monitor.defer(self.wer_decode_finish(wer_pend, log_probs, encoded_len, language_ids, h_tgt, ctc_ready))
                main.wait_event(wer_pend["snap"])   # the optimizer step that follows must not overtake the decode's weight copies
            else:
                wer, wer_ctc = self.batch_wer(encoded, encoded_len, log_probs, transcript, transcript_len, language_ids)
                monitor['training_batch_wer'], monitor['training_batch_wer_ctc'] = wer, wer_ctc
        self._step += 1
        if return_probs:
            return loss_value, monitor, log_probs
        return loss_value, monitor


EncDecHybridRNNTCTCBPEModel = EncDecHybridRNNTCTCModel


def freeze_layer(model, num_layers):
    """R/utils.py:246-263 verbatim semantics."""
    for p in model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.encoder.layers):
        if i > num_layers:
            for p in layer.parameters():
                p.requires_grad = True
    for mod in (model.decoder, model.ctc_decoder, model.joint):
        for p in mod.parameters():
            p.requires_grad = True
