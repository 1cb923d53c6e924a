"""Prediction network, joint network (fused joint+loss per sub-batch) and the CTC head, with the reference's
module tree / parameter names (A/modules/rnnt.py:524-792,1175-1710; A/modules/conv_asr.py:402-490;
C/parts/rnn.py:151-235,536-561).
"""
from contextlib import nullcontext
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


def label_collate(labels, device=None):
    """C/parts/rnn.py:536-561: list of label lists / tensor -> padded int64 [B,U] (pad 0.0)."""
    if isinstance(labels, torch.Tensor):
        return labels.type(torch.int64)
    if not isinstance(labels, (list, tuple)):
        raise ValueError(f"`labels` should be a list or tensor not {type(labels)}")
    B = len(labels)
    U = max(len(l) for l in labels)
    out = torch.zeros((B, U), dtype=torch.int64, device=device)
    for e, l in enumerate(labels):
        out[e, :len(l)] = torch.as_tensor(l, dtype=torch.int64)
    return out


def _flush_pending():
    from . import cl
    if cl._PENDING_OPTIMIZERS:
        cl.flush_pending_updates()


class LSTMDropout(nn.Module):
    def __init__(self, input_size, hidden_size, dropout, forget_gate_bias=1.0):
        super().__init__()
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=1)
        with torch.no_grad():  # C/parts/rnn.py:212-219
            self.lstm.bias_ih_l0[hidden_size:2 * hidden_size].fill_(forget_gate_bias)
            self.lstm.bias_hh_l0[hidden_size:2 * hidden_size] *= 0.0
        self.dropout = nn.Dropout(dropout) if dropout else None
        self.use_hip = False  # set by RNNTDecoder in the bf16 configuration (persistent HIP LSTM, ops/lstm.py)

    def forward(self, x, h=None, need_state=True):
        """`need_state=False` (the training step: RNNTDecoder.forward discards the final state) selects the persistent HIP
        LSTM, which only produces the output sequence; stateful callers (greedy decoding: predict() with a carried
        (h, c)) get the library LSTM and its (h_n, c_n)."""
        from .ops import lstm as hip_lstm
        if self.use_hip and h is None and not need_state and hip_lstm.lstm_supported(x, self.lstm.hidden_size):
            x, h = hip_lstm.lstm_forward(x, self.lstm), None
        else:
            x, h = self.lstm(x, h)
        if self.dropout:
            x = self.dropout(x)
        return x, h


class RNNTDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        vocab = len(cfg.languages) * cfg.vocab_per_lang
        self.pred_hidden = cfg.pred_hidden
        self.blank_idx = vocab
        self.blank_as_pad = True
        self.prediction = nn.ModuleDict({
            "embed": nn.Embedding(vocab + 1, cfg.pred_hidden, padding_idx=vocab),
            "dec_rnn": LSTMDropout(cfg.pred_hidden, cfg.pred_hidden, cfg.pred_dropout),
        })
        self.prediction["dec_rnn"].use_hip = (cfg.compute_dtype == "bf16")

    def predict(self, y=None, state=None, add_sos=True, batch_size=None, need_state=True):
        p = next(self.parameters())
        if y is not None:
            y = self.prediction["embed"](y.to(p.device))
        else:
            B = batch_size if batch_size is not None else (1 if state is None else state[0].size(1))
            y = torch.zeros((B, 1, self.pred_hidden), device=p.device, dtype=p.dtype)
        if add_sos:
            B, U, H = y.shape
            y = torch.cat([torch.zeros((B, 1, H), device=y.device, dtype=y.dtype), y], dim=1).contiguous()
        g, hid = self.prediction["dec_rnn"](y.transpose(0, 1), state, need_state=need_state)
        return g.transpose(0, 1), hid

    def forward(self, targets, target_length, states=None):
        _flush_pending()
        y = label_collate(targets)
        rnn, emb = self.prediction["dec_rnn"], self.prediction["embed"]
        from .ops import lstm as hip_lstm
        if (states is None and rnn.use_hip and y.is_cuda and y.shape[1] > 0
                and hip_lstm.lstm_supported(emb.weight, rnn.lstm.hidden_size)):
            # training step on the MI355X: zero SOS row + embedding rows written time-major in bf16 by one launch (ops/tail.py),
            # straight into the persistent LSTM; its output [U+1,B,H] is handed on as a [B,H,U+1] VIEW (no transpose copy)
            from .ops import tail
            xb = tail.embed_sos(emb.weight, y, pad_row=emb.padding_idx if emb.padding_idx is not None else -1)
            g = hip_lstm.lstm_forward(xb, rnn.lstm)                  # [U+1, B, H] f32
            if rnn.dropout:
                g = rnn.dropout(g)
            return g.permute(1, 2, 0), target_length, None           # (B, H, U+1)
        g, states = self.predict(y, state=states, add_sos=True, need_state=False)  # (B, U+1, H)
        return g.transpose(1, 2), target_length, states          # (B, H, U+1)


class RNNTJoint(nn.Module):
    """joint_net = [ReLU, Dropout, ModuleDict{lang: Linear(H, 257)}] exactly as rnnt.py:1694-1703 so parameter
    names match (`joint.joint_net.2.hi.weight`)."""

    def __init__(self, cfg, loss=None):
        super().__init__()
        self.cfg = cfg
        self.pred = nn.Linear(cfg.pred_hidden, cfg.joint_hidden)
        self.enc = nn.Linear(cfg.d_model, cfg.joint_hidden)
        final = nn.ModuleDict({l: nn.Linear(cfg.joint_hidden, cfg.vocab_per_lang + 1) for l in cfg.languages})
        layers = [nn.ReLU(inplace=True)] + ([nn.Dropout(p=cfg.joint_dropout)] if cfg.joint_dropout else []) + [final]
        self.joint_net = nn.Sequential(*layers)
        self._fused_batch_size = cfg.fused_batch_size
        self._loss = loss
        self._fuse_loss_wer = True
        self.store_sub_enc = False
        self.store_sub_logits = False
        self.detach_sub_enc = False
        self.store_list: List[torch.Tensor] = []
        self.temp_logits = None
        # MI355X: whole-batch fused joint+loss on the matrix cores (ops/joint.py) whenever the step does not need
        # the per-sub-batch logits stash (MAS importance / LwF passes keep the stash semantics on the unfused path)
        self.use_fused = True
        self.loss_scale_hint = 1.0   # expected |d loss / d cost_b| (set by the model: (1-ctc_w)/B)
        self.dropout_seed = 0
        # training_step sets return_costs: the fused path then leaves the per-utterance costs [B] in last_costs and returns no
        # reduced loss -- ops/tail.loss_combine forms (1-w) mean(costs) + w mean(nll) and the monitor's values in one launch
        self.return_costs = False
        self.last_costs = None

    @property
    def loss(self):
        return self._loss

    def set_loss(self, loss):
        self._loss = loss

    @property
    def fused_batch_size(self):
        return self._fused_batch_size

    def set_fused_batch_size(self, n):
        self._fused_batch_size = n

    def project_encoder(self, x):
        return self.enc(x)

    def project_prednet(self, x):
        return self.pred(x)

    def joint_after_projection(self, f, g, language_ids=None):
        inp = f.unsqueeze(2) + g.unsqueeze(1)  # [B,T,U,H]
        for m in self.joint_net[:-1]:
            inp = m(inp)
        heads = self.joint_net[-1]
        if language_ids is None:
            raise ValueError("language_ids are required by the multilingual joint (rnnt.py:1624-1640)")
        if len(set(language_ids)) == 1:
            res = heads[language_ids[0]](inp)
        else:
            res = torch.stack([heads[l](x) for x, l in zip(inp, language_ids)])
        if self.store_sub_logits:
            self.temp_logits = res.clone()
        return res  # raw logits: the accelerator branch of rnnt.py:1651-1656

    def joint(self, f, g, language_ids=None):
        return self.joint_after_projection(self.project_encoder(f), self.project_prednet(g), language_ids)

    def forward(self, encoder_outputs, decoder_outputs, encoder_lengths=None, transcripts=None,
                transcript_lengths=None, compute_wer=False, language_ids=None, host_lengths=None):
        """Fused joint + loss over sub-batches (rnnt.py:1403-1561).  Returns (loss, wer, wer_num, wer_denom);
        `host_lengths` = (enc_lens list, tgt_lens list) lets the loop narrow each sub-batch without a
        device->host sync (the reference calls .max() on device tensors per sub-batch, :1440-1441)."""
        _flush_pending()
        enc = encoder_outputs.transpose(1, 2)
        dec = decoder_outputs.transpose(1, 2)
        if (encoder_lengths is None) or (transcript_lengths is None):
            raise ValueError("`fuse_loss_wer` is set, therefore encoder and target lengths must be provided as well!")
        if self._loss is None:
            raise ValueError("`fuse_loss_wer` flag is set, but `loss` and `wer` modules were not provided! ")
        if host_lengths is None:
            host_lengths = (encoder_lengths.tolist(), transcript_lengths.tolist())
        h_enc, h_tgt = host_lengths
        B = int(enc.size(0))
        if self._fused_eligible(enc, language_ids, max(h_tgt) + 1):
            return self._forward_fused(enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids,
                                       h_enc, h_tgt), None, None, None
        amp = (torch.autocast(device_type="cuda", dtype=torch.bfloat16)
               if self.cfg.compute_dtype == "bf16" and enc.is_cuda else nullcontext())
        losses, target_lengths, stash = [], [], []
        for begin in range(0, B, self._fused_batch_size):
            end = min(begin + self._fused_batch_size, B)
            max_t = max(h_enc[begin:end])
            max_u = max(h_tgt[begin:end])
            sub_enc = enc[begin:end, :max_t]
            sub_dec = dec[begin:end, :max_u + 1]
            sub_tr = transcripts[begin:end, :max_u]
            with amp:
                sub_joint = self.joint(sub_enc, sub_dec, language_ids=language_ids[begin:end])
            if self.store_sub_enc:
                stash.append(sub_joint.detach().clone() if self.detach_sub_enc else sub_joint.clone())
            if self.store_sub_logits:
                lg = self.temp_logits
                stash.append(lg.detach().clone() if self.detach_sub_enc else lg.clone())
            red = self._loss.reduction
            self._loss.reduction = None
            loss_batch = self._loss(log_probs=sub_joint, targets=sub_tr, input_lengths=encoder_lengths[begin:end],
                                    target_lengths=transcript_lengths[begin:end], max_T=max_t, max_U=max_u)
            self._loss.reduction = red
            losses.append(loss_batch)
            target_lengths.append(transcript_lengths[begin:end])
        losses = self._loss.reduce(losses, target_lengths)
        if self.store_sub_enc or self.store_sub_logits:
            self.store_list = stash
        return losses, None, None, None


    # ------------------------------------------------------------------ fused path
    def _fused_eligible(self, enc, language_ids, U1):
        from . import _lib
        from .ops.joint import fused_joint_supported
        lk = self._loss._loss
        H, V = self.cfg.joint_hidden, self.cfg.vocab_per_lang + 1
        # several languages in one batch (the reference picks the head per sample, A/modules/rnnt.py:1632-1640): the fused
        # kernels take ONE head, so _forward_fused runs them once per language over that language's utterances -- unless
        # the MAS / LwF stash is requested (its sub-batch boxes are defined on the batch order)
        mixed = language_ids is not None and len(set(language_ids)) > 1
        ok = (self.use_fused and self.cfg.compute_dtype == "bf16"
              and language_ids is not None and lk.clamp <= 0.0
              and not (mixed and (self.store_sub_enc or self.store_sub_logits or self._loss.reduction != 'mean_batch'))
              and fused_joint_supported(H, V, enc.device))
        if ok and (self.store_sub_enc or self.store_sub_logits):
            # the stash of the MAS / LwF passes stays on the lattice (ops.joint.LatticeStash); its terms' gradient needs the
            # fused hidden- and weight-gradient kernels
            L = _lib.lib()
            LD = L.ia_joint_ld(V)
            ok = bool(L.ia_joint_dh_fused_supported(U1, H, LD) and L.ia_joint_dw_fused_supported(U1, H, LD))
        return ok

    def _forward_fused(self, enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids, h_enc, h_tgt):
        if len(set(language_ids)) > 1:
            return self._forward_fused_mixed(enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids, h_enc, h_tgt)
        return self._forward_fused_one(enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids, h_enc, h_tgt)

    def _forward_fused_mixed(self, enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids, h_enc, h_tgt):
        """Mixed-language batch: one fused joint + loss pass per language over that language's utterances (gathered with
        index_select: autograd scatters their gradients back), the per-utterance costs returned to batch order.  The mean over
        the batch does not depend on the order the utterances were scored in."""
        B = len(language_ids)
        order = []
        for lang in language_ids:
            if lang not in order:
                order.append(lang)
        keep_rc, keep_seed = self.return_costs, self.dropout_seed
        costs = None
        self.return_costs = True
        try:
            for gi, lang in enumerate(order):
                idx_h = [i for i in range(B) if language_ids[i] == lang]
                idx = torch.tensor(idx_h, dtype=torch.long, device=enc.device)
                self.dropout_seed = (keep_seed + 0x9E3779B1 * gi) & 0xFFFFFFFF   # unrelated masks per group
                self.last_costs = None
                self._forward_fused_one(enc.index_select(0, idx), dec.index_select(0, idx), encoder_lengths.index_select(0, idx),
                                        transcripts.index_select(0, idx), transcript_lengths.index_select(0, idx), [lang] * len(idx_h),
                                        [h_enc[i] for i in idx_h], [h_tgt[i] for i in idx_h])
                c = self.last_costs
                costs = c.new_zeros(B) if costs is None else costs
                costs = costs.index_copy(0, idx, c)
        finally:
            self.return_costs, self.dropout_seed = keep_rc, keep_seed
        if self.return_costs and self._loss.reduction == 'mean_batch':
            self.last_costs = costs
            return None
        self.last_costs = None
        return self._loss.reduce([costs], [transcript_lengths])

    def _forward_fused_one(self, enc, dec, encoder_lengths, transcripts, transcript_lengths, language_ids, h_enc, h_tgt):
        from .ops.joint import fused_joint_rnnt
        max_t, max_u = max(h_enc), max(h_tgt)
        req = None
        if self.store_sub_enc or self.store_sub_logits:
            req = {"sub": self._fused_batch_size, "h_enc": h_enc, "h_tgt": h_tgt,
                   "detach": bool(self.detach_sub_enc) or not torch.is_grad_enabled()}
        head = self.joint_net[-1][language_ids[0]]
        p = 0.0
        for m in self.joint_net[:-1]:
            if isinstance(m, nn.Dropout) and self.training:
                p = float(m.p)
        from .ops import fast
        lk = self._loss._loss
        encn, decn = enc[:, :max_t], dec[:, :max_u + 1]
        d_in, h_in = self.enc.weight.shape[1], self.pred.weight.shape[1]
        if (encn.is_contiguous() and encn.dtype == torch.float32 and decn.shape[1] == dec.shape[1]
                and decn.dtype in (torch.float32, torch.bfloat16) and fast.gemm_supported(d_in, self.enc.weight.shape[0])
                and fast.gemm_supported(h_in, self.pred.weight.shape[0]) and self.enc.bias is not None and self.pred.bias is not None):
            # the two projections run INSIDE the fused node (ops/joint.py, projection mode): f16 operands straight out of the
            # HIP GEMMs, the prediction network's time-major output gathered by one launch, no casts / copies in between
            costs = fused_joint_rnnt(encn, decn, head.weight, head.bias, transcripts[:, :max_u].contiguous().long(),
                                     encoder_lengths.long(), transcript_lengths.long(), lk.blank, dropout_p=p,
                                     seed=self.dropout_seed, fastemit_lambda=lk.fastemit_lambda,
                                     scale_hint=self.loss_scale_hint, stash_req=req,
                                     enc_proj=(self.enc.weight, self.enc.bias), pred_proj=(self.pred.weight, self.pred.bias))
        else:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                # (fast.linear: HIP GEMM forward, weight + bias gradient on csrc/gemm_tn.hip instead of a 16-workgroup library
                # TN GEMM and a bf16 column reduction; falls back to F.linear outside its shape limits)
                f = fast.linear(encn, self.enc.weight, self.enc.bias)
                g = fast.linear(decn, self.pred.weight, self.pred.bias)
            costs = fused_joint_rnnt(f, g, head.weight, head.bias, transcripts[:, :max_u].contiguous().long(),
                                     encoder_lengths.long(), transcript_lengths.long(), lk.blank, dropout_p=p,
                                     seed=self.dropout_seed, fastemit_lambda=lk.fastemit_lambda,
                                     scale_hint=self.loss_scale_hint, stash_req=req)
        if req is not None:
            self.store_list = req["out"]
        if self.return_costs and self._loss.reduction == 'mean_batch':
            self.last_costs = costs
            return None
        return self._loss.reduce([costs], [transcript_lengths])


class _CtcHeadHip(torch.autograd.Function):
    """Single-language CTC head y = x W_sel^T + b_sel on the HIP GEMMs: forward = ia_gemm_bf16 with the 257 selected
    rows padded to 264 (16-byte rows), backward = one library GEMM for dx and ia_gemm_tn_bf16 for dW and db (the ATen
    path spends 89 us in a 16-workgroup TN GEMM and 131 us in a bf16 column reduction for the bias)."""

    @staticmethod
    def forward(ctx, x, w_sel, b_sel):
        from .ops import fast
        shp = x.shape
        V, d = w_sel.shape
        Vp = (V + 7) // 8 * 8
        xb = x.reshape(-1, d).to(torch.bfloat16).contiguous()
        wp = torch.zeros(Vp, d, dtype=torch.bfloat16, device=x.device)
        wp[:V] = w_sel.detach()
        bp = torch.zeros(Vp, dtype=torch.float32, device=x.device)
        bp[:V] = b_sel.detach().float()
        out = torch.empty(xb.shape[0], Vp, dtype=torch.float32, device=x.device)
        fast.gemm(xb, wp, bp, out_f32=out, want_bf16=False)
        ctx.save_for_backward(xb, wp)
        ctx.meta = (shp, V, x.dtype, w_sel.dtype, b_sel.dtype)
        return out[:, :V].reshape(*shp[:-1], V)

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        from .ops import fast
        xb, wp = ctx.saved_tensors
        shp, V, xdt, wdt, bdt = ctx.meta
        Vp, d = wp.shape
        M = xb.shape[0]
        L = _lib.lib()
        dyf = dy.reshape(M, V).float()
        if dyf.stride(1) != 1:
            dyf = dyf.contiguous()
        # f32 [M, 257] -> bf16 [M, 264] with zero padding in one pass (an ATen strided copy + a strided fill of the 7
        # padding columns took 0.12 + 0.39 ms)
        dyb = torch.empty(M, Vp, dtype=torch.bfloat16, device=dy.device)
        _lib.check(L.ia_cast_pad_bf16(_lib.ptr(dyf), dyf.stride(0), M, V, _lib.ptr(dyb), Vp, _lib.stream_ptr()), "ia_cast_pad_bf16")
        dx = None
        if ctx.needs_input_grad[0]:   # dX = dY W_sel on the HIP GEMM against W_sel^T [d, 264]
            dx = fast.gemm(dyb, wp.t().contiguous())[1].view(shp).to(xdt)
        dW, db = fast.gemm_tn(dyb, xb)
        return dx, dW[:V].to(wdt), db[:V].to(bdt)


class ConvASRDecoder(nn.Module):
    """Parameter layout of the reference (Conv1d(d -> n_lang*256+1, k=1), conv_asr.py:445) but the projection
    touches only the batch language's 257 rows: the reference computes all 5633 columns and masked_selects
    (:469-480), 22x wasted FLOPs and an H2D mask copy per call."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.vocab_per_lang = cfg.vocab_per_lang
        n = len(cfg.languages) * cfg.vocab_per_lang + 1
        self._num_classes = n
        self.decoder_layers = nn.Sequential(nn.Conv1d(cfg.d_model, n, kernel_size=1, bias=True))
        nn.init.xavier_uniform_(self.decoder_layers[0].weight, gain=1.0)
        self.lang_index = {l: i for i, l in enumerate(cfg.languages)}
        self.language_masks = {}
        for l, i in self.lang_index.items():
            m = [False] * n
            m[i * cfg.vocab_per_lang:(i + 1) * cfg.vocab_per_lang] = [True] * cfg.vocab_per_lang
            m[-1] = True
            self.language_masks[l] = m
        self.return_logits_ = False
        self.decoder_logits = None
        self.temperature = 1.0
        self._rows_cache = {}

    def _rows(self, lang, device):
        key = (lang, str(device))
        r = self._rows_cache.get(key)
        if r is None:
            i = self.lang_index[lang]
            v = self.vocab_per_lang
            r = self._rows_cache[key] = torch.cat([torch.arange(i * v, (i + 1) * v, device=device),
                                                   torch.tensor([self._num_classes - 1], device=device)])
        return r

    def fused_loss_supported(self, encoder_output, language_ids, targets):
        """The head + CTC loss as ONE autograd node on raw logits (ops/tail.ctc_head_loss): single-language batch, bf16 mode,
        nothing asks for the log-probs / logits tensors themselves (LwF's return_probs, MAS's return_logits_)."""
        from .ops import tail
        return (self.cfg.compute_dtype == "bf16" and language_ids is not None and len(set(language_ids)) == 1
                and not self.return_logits_ and self.temperature == 1.0
                and tail.ctc_head_loss_supported(encoder_output, targets, self.cfg.d_model))

    def forward_loss(self, encoder_output, language_ids, targets, input_lengths, target_lengths, zero_infinity=True, keep=None):
        """nll [B] (reduction 'none') of ConvASRDecoder.forward + CTCLoss.forward for a single-language batch."""
        _flush_pending()
        from .ops import tail
        conv = self.decoder_layers[0]
        i, v = self.lang_index[language_ids[0]], self.vocab_per_lang
        x = encoder_output.transpose(1, 2)   # [B,T,d]
        return tail.ctc_head_loss(x, conv.weight, conv.bias, targets, input_lengths, target_lengths, i * v, v, self._num_classes - 1,
                                  blank=v, zero_infinity=zero_infinity, keep=keep)

    def forward(self, encoder_output, language_ids=None):
        _flush_pending()
        w = self.decoder_layers[0].weight.squeeze(-1)  # [n, d]
        b = self.decoder_layers[0].bias
        x = encoder_output.transpose(1, 2)  # [B,T,d]
        amp = (torch.autocast(device_type="cuda", dtype=torch.bfloat16)
               if self.cfg.compute_dtype == "bf16" and x.is_cuda else nullcontext())
        with amp:
            if language_ids is None:
                out = F.linear(x, w, b)
            elif len(set(language_ids)) == 1:
                rows = self._rows(language_ids[0], x.device)
                w_sel, b_sel = w.index_select(0, rows), b.index_select(0, rows)
                if self.cfg.compute_dtype == "bf16" and x.is_cuda and x.shape[-1] % 8 == 0:
                    out = _CtcHeadHip.apply(x, w_sel, b_sel)
                else:
                    out = F.linear(x, w_sel, b_sel)
            else:
                out = torch.stack([F.linear(xi, w.index_select(0, self._rows(l, x.device)),
                                            b.index_select(0, self._rows(l, x.device)))
                                   for xi, l in zip(x, language_ids)])
        out = out.float()
        if self.temperature != 1.0:
            out = out / self.temperature
        if self.return_logits_:
            self.decoder_logits = out.clone()
        return F.log_softmax(out, dim=-1)
