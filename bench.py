#!/usr/bin/env python3
"""bench.py -- utterances/sec of one Conformer-M hybrid RNNT-CTC + EWC training step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = zero_grad -> training_step (features, SpecAugment, Conformer encoder, prediction net, fused joint+RNNT
loss per sub-batch, CTC head+loss) -> EWC penalty pre-loaded into .grad -> backward -> RCCL gradient all-reduce
(N>1) -> fused AdamW, on a synthetic batch already resident in HBM (audio 0.1*N(0,1), lengths
U(0.6,1)*L with one full-length item, 7 tokens/s, language 'hi', seed 1234).  Workload at N=1 = BASELINE
configs[1]: Conformer-medium (d=256, 16 layers), EWC, bs=32 per GPU, 15 s utterances, bf16 projections, encoder
layers <= 12 frozen as in the reference's config.yaml.  Weak scaling: every rank processes its own 32 utterances.

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline       the HBM-bound transducer gradient kernel timed with HIP events inside the timed steps; `traffic` = HBM
                 bytes per launch from this round's PMC passes of the same command (profiles/r03_pmc_traffic.json)
  roofline_mfma  the dominant MFMA kernel of the step (fused joint hidden-gradient kernel), timed the same way
  cpu_baseline   the CPU oracle (oracle/step_ref.py, fp32, torch intra-op threads = host cores) on a bounded sample of the
                 SAME workload + the GPU-vs-oracle loss error on that sample with identical weights (`loss_rel_err`) + the
                 oracle timed on BASELINE configs[0] exactly (`config1`)
  config.with_h2d_prefetch  the same step fed from pinned host memory by a one-batch-ahead asynchronous copy (PCIe
                 inclusive; never `value`)
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the hosts of this pool only support dmabuf IPC (RCCL needs it across ranks)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFS = 2500.0   # same guide: dense bf16 / f16 MFMA peak (the 5 PF headline includes 2:1 sparsity)


def measure_peaks(dev):
    """peaks MEASURED on this box -- a 16-byte-per-lane streaming copy of 1 GiB (read + write bytes / time) and
    back-to-back v_mfma_f32_16x16x32_bf16 on register operands (csrc/peaks.hip), both timed with events over 10 launches after
    2 warm-up launches.  Returns {"hbm_copy_GBs", "mfma_bf16_TFs"}; roofline fractions are reported against these AND the
    datasheet figures."""
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    n = 1 << 30
    src = torch.full((n,), 1, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    sp = _lib.stream_ptr()

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e-3 / reps
    t_copy = timed(lambda: _lib.check(L.ia_peak_stream_copy(_lib.ptr(src), _lib.ptr(dst), n, sp), "ia_peak_stream_copy"))
    del src, dst
    wgs, iters = 256 * 8, 4096
    sink = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    t_mfma = timed(lambda: _lib.check(L.ia_peak_mfma_bf16(_lib.ptr(sink), wgs, iters, sp), "ia_peak_mfma_bf16"))
    flops = float(L.ia_peak_mfma_bf16_flops(wgs, iters))
    return {"hbm_copy_GBs": round(2.0 * n / t_copy / 1e9, 1), "mfma_bf16_TFs": round(flops / t_mfma / 1e12, 1),
            "how": "1 GiB float4 copy (read+write bytes); 2048 workgroups x 4 waves x 4096 x 8 back-to-back 16x16x32 bf16 MFMAs on "
                   "non-trivial register operands; 10 launches each after 2 warm-up launches"}


def synth_batch(B, seconds, device, seed=1234, vocab=256, sr=16000):
    g = torch.Generator().manual_seed(seed)
    L = int(seconds * sr)
    lens = torch.round(L * (0.6 + 0.4 * torch.rand(B, generator=g))).long()
    lens[0] = L
    x = 0.1 * torch.randn(B, L, generator=g)
    x[:, 1:] = 0.5 * (x[:, 1:] + x[:, :-1])  # 2-tap low-pass
    for b in range(B):
        x[b, lens[b]:] = 0
    dur = lens.float() / sr
    from indic_cl_asr_amd.encoder import subsampled_length
    from indic_cl_asr_amd.features import mel_frame_count
    tl = []
    for b in range(B):
        tp = subsampled_length(mel_frame_count(int(lens[b])))
        tl.append(max(1, min(int(round(7 * float(dur[b]))), tp - 1)))
    tl = torch.tensor(tl)
    U = int(tl.max())
    tok = torch.randint(0, vocab, (B, U), generator=g)
    batch = (x.to(device), lens.to(device), tok.to(device), tl.to(device))
    return batch, (lens.tolist(), tl.tolist())


class HipEvents:
    """Pool of hipEvent_t (ctypes on libamdhip64) recorded by ia_rnnt_backward around the gradient kernel."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.pairs, self.shapes, self.free = [], [], []
        self.enabled = False
        self.touched_fraction = None
        self.full_scale = 1.0
        self.kernel_name = "rnnt_grad"

    def _new(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def hook(self, B, T, U1, V, elem_bytes=4, passes=2, kernel=None, skips_dead_frames=False):
        """passes = lattice-sized streams the kernel must move: read logits + write grads (2); the
        library-GEMM fallback of the f16 path also writes G^T for the split-K weight-gradient GEMM (3)."""
        if kernel is not None:
            self.kernel_name = kernel
        if not self.enabled:
            return None
        a, b = self._new(), self._new()
        self.pairs.append((a, b))
        # the kernel leaves the tiles behind frame T_b + 3 of every utterance alone (nothing reads them): the algorithmic
        # bytes are those of the cells it has to touch (self.touched_fraction, from the host-side lengths of the batch)
        frac = self.touched_fraction if (skips_dead_frames and self.touched_fraction is not None) else 1.0
        self.shapes.append((B, T, U1, V, elem_bytes * passes / 2.0 * frac))
        self.full_scale = 1.0 / frac   # full lattice / touched cells (reported next to the roofline, never as `achieved`)
        return a, b

    def summary(self):
        ms, byts = [], []
        for (a, b), (B, T, U1, V, eb) in zip(self.pairs, self.shapes):
            t = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(t), a, b) == 0:
                ms.append(t.value)
                # read logits once + write grads once, eb bytes per lattice element
                byts.append(2.0 * eb * B * T * U1 * V)
        if not ms:
            return None
        return sum(ms) / len(ms), sum(byts) / len(byts), len(ms)


def cpu_baseline(seconds, cfg_kw, sample_bs=4, freeze_till=12, device=None):
    """CPU oracle on a bounded sample: ONE sub-batch (fused_batch_size=4 utterances) of the same workload, one full step
    (fwd + EWC penalty + bwd + AdamW); the product then runs the SAME 4 utterances on the SAME weights (dropout, SpecAugment
    and dither off on both sides) and the relative loss errors are reported; finally the oracle is timed on BASELINE
    configs[0] exactly (Conformer-small d=144, bs 2 x 5 s, naive fine-tune step)."""
    from oracle import step_ref as S
    torch.manual_seed(0)
    o = S.OracleHybridModel(d_model=cfg_kw["d_model"], n_layers=cfg_kw["n_layers"], n_heads=cfg_kw["n_heads"],
                            pred_hidden=cfg_kw["pred_hidden"], joint_hidden=cfg_kw["joint_hidden"])
    S.freeze_layer(o, freeze_till)
    o.train()
    (sig, sl, tok, tl), _ = synth_batch(sample_bs, seconds, "cpu")
    state0 = {k: v.clone() for k, v in o.state_dict().items()}
    params = S.get_params(o)
    fish = {n: torch.rand_like(p) * 1e-3 for n, p in params.items()}
    ck = {n: p.detach().clone() for n, p in params.items()}
    opt = torch.optim.AdamW([p for p in o.parameters() if p.requires_grad], lr=1e-4)
    t0 = time.time()
    opt.zero_grad()
    loss, mon_o = o.training_step((sig, sl, tok, tl), ['hi'] * sample_bs)
    pen, _ = S.ewc_penalty_grads(10.0, fish, S.get_params(o), ck)
    for n, p in o.named_parameters():
        p.grad = pen[n] if n in pen else None
    loss.backward()
    opt.step()
    dt = time.time() - t0
    out = dict(value=sample_bs / dt, unit="utterances/s", cores=torch.get_num_threads(), kind="port",
               sample=f"1 step of {sample_bs} x {seconds:g} s utterances (one fused sub-batch of the same workload), "
                      f"fp32 oracle/step_ref.py + oracle/rnnt_ref.c, {dt:.1f} s wall")
    if device is not None:   # the checker's verdict on the measured path: same utterances, same weights
        from indic_cl_asr_amd.config import model_config
        from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
        m = EncDecHybridRNNTCTCModel(model_config(compute_dtype="bf16", dither=0.0, **cfg_kw))
        m.load_state_dict(state0)
        m = m.disable_dropout().to(device).train()
        m.spec_augment_enabled = False
        freeze_layer(m, freeze_till); m.encoder.encoder_frozen_till = freeze_till
        with torch.no_grad():
            _, mon_p = m.training_step(tuple(t.to(device) for t in (sig, sl, tok, tl)), ['hi'] * sample_bs)
        out["loss_rel_err"] = {k: abs(mon_p[k] - mon_o[k]) / abs(mon_o[k]) for k in ("train_rnnt_loss", "train_ctc_loss", "train_loss")}
        out["oracle_train_loss"] = mon_o["train_loss"]
        del m
    # BASELINE configs[0] on the CPU path, exactly: d=144, 16 L, 4 heads, H=320, bs 2 x 5 s, naive step (no CL term)
    from indic_cl_asr_amd.config import PRESETS
    torch.manual_seed(0)
    o1 = S.OracleHybridModel(**PRESETS["small"])
    S.freeze_layer(o1, freeze_till)
    o1.train()
    (sig1, sl1, tok1, tl1), _ = synth_batch(2, 5.0, "cpu")
    opt1 = torch.optim.AdamW([p for p in o1.parameters() if p.requires_grad], lr=1e-4)
    best = None
    for _ in range(3):
        t0 = time.time()
        opt1.zero_grad()
        l1, _ = o1.training_step((sig1, sl1, tok1, tl1), ['hi'] * 2)
        l1.backward()
        opt1.step()
        d1 = time.time() - t0
        best = d1 if best is None else min(best, d1)
    out["config1"] = dict(value=2 / best, unit="utterances/s", ms_per_step=round(best * 1e3, 1),
                          sample="BASELINE configs[0]: Conformer-small d=144 16L, bs 2 x 5 s, naive step, best of 3")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--preset", default="medium")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--freeze", type=int, default=12, help="config.yaml model.freeze_encoder_till (-1: train everything)")
    ap.add_argument("--cl", default="ewc", choices=["ewc", "mas_importance", "lwf"],
                    help="step recipe: EWC step (the headline, BASELINE configs[1]); MAS importance pass (configs[2], "
                         "R/cl_baseline_mas.py:258-270); LwF teacher + student step (configs[3], R/cl_baseline_lwf.py:212-264)")
    ap.add_argument("--fp8-prefix", action="store_true", help="e4m3 projections in the frozen prefix (BASELINE configs[4]; never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wer", action="store_true", help="timed steps WITHOUT the in-step greedy decode + WER of the reference's "
                                                         "training_step (compute_wer = True there); default: with it, monitor read after every step")
    ap.add_argument("--no-wer-leg", action="store_true", help="skip the extra leg that times the step the other way round (without / with the in-step WER)")
    ap.add_argument("--no-peaks", action="store_true", help="skip the on-box copy / MFMA peak microbenchmarks")
    ap.add_argument("--grad-exchange", default="fp32", choices=["fp32", "bf16"], help="dtype of the data-parallel gradient all-reduce")
    ap.add_argument("--cpu-sample-bs", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    import torch.distributed as dist
    # rehearsal of the multi-rank path on a one-GPU box (developer switch, never the measured configuration): every rank on
    # device 0, gloo instead of RCCL
    rehearsal = os.environ.get("IA_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.losses import rnnt as rnnt_mod
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer

    torch.manual_seed(1234)
    cfg = model_config(args.preset, compute_dtype=args.dtype, fp8_frozen_prefix=bool(args.fp8_prefix))
    model = EncDecHybridRNNTCTCModel(cfg).to(dev)
    if args.freeze >= 0:
        freeze_layer(model, args.freeze)
        model.encoder.encoder_frozen_till = args.freeze
    if world > 1:   # the reference's DDP recipe (R/cl_baseline.py:133): BatchNorm statistics over the global batch
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model.train()
    flat = cl.FlatParams(model)
    opt = cl.FusedAdamW(flat, lr=1e-4, grad_exchange_dtype=args.grad_exchange)
    # EWC state of "task > 0": Fisher from a previous task (synthetic, positive) and the previous optimum
    fisher = cl.get_zero_params(model)
    fisher.flat.copy_(torch.rand(flat.numel, device=dev) * 1e-3)
    checkpoint = cl.get_params_clone(model)
    batch, host_lens = synth_batch(args.batch, args.seconds, dev, seed=1234 + rank)
    langs = ['hi'] * args.batch

    events = HipEvents()
    rnnt_mod.PROFILE_HOOK = events.hook
    try:   # cells the gradient kernel has to touch: frames < T_b + 8 of every utterance (+ the 64-cell tile that straddles the end)
        from indic_cl_asr_amd.encoder import subsampled_length
        from indic_cl_asr_amd.features import mel_frame_count
        h_enc = [int(subsampled_length(mel_frame_count(int(n), cfg.n_fft, cfg.n_window_stride))) for n in host_lens[0]]
        Tq, U1q = max(h_enc), max(int(u) for u in host_lens[1]) + 1
        live = sum(min(Tq * U1q, (min(Tq, t + 8) * U1q + 63) // 64 * 64) for t in h_enc)
        events.touched_fraction = live / float(len(h_enc) * Tq * U1q)
    except Exception:
        events.touched_fraction = None
    from indic_cl_asr_amd.ops import joint as joint_mod
    mfma_events = []          # (start, stop, flops) of the fused hidden-gradient kernel (the dominant MFMA kernel)
    timing = {"on": False}

    def mfma_hook(flops):
        if not timing["on"]:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        mfma_events.append((a, b, flops))
        return a, b
    joint_mod.MFMA_PROFILE_HOOK = mfma_hook

    omega = cl.get_zero_params(model) if args.cl == "mas_importance" else None
    teacher = cl.get_params_clone(model) if args.cl == "lwf" else None
    m_ = getattr(model, "module", model)

    wer_on = args.cl == "ewc" and not args.no_wer
    last_monitor = [None]
    if wer_on:
        m_.wer.log_prediction = False; m_.ctc_wer.log_prediction = False       # R/cl_baseline.py:127-128

    def step():
        opt.zero_grad()
        if args.cl == "ewc":
            # the reference's step decodes every training batch greedily for the monitor's two WERs (compute_wer = True,
            # hybrid_rnnt_ctc_models.py:875) and its loops log the monitor after every optimizer step (R/cl_baseline.py:198-206)
            loss, monitor = model.training_step(batch, langs, host_lengths=host_lens, compute_wer=wer_on)
            monitor['ewc_penalty'] = cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
            loss.backward()
            opt.step()
            if wer_on:
                last_monitor[0] = monitor
                _ = monitor["training_batch_wer"]
        elif args.cl == "mas_importance":   # the importance pass after a task: |d (logit L2 norms) / d theta| accumulated into omega
            m_.joint.store_sub_logits = True; m_.ctc_decoder.return_logits_ = True
            loss, monitor = model.training_step(batch, langs, host_lengths=host_lens, compute_wer=False)   # (non-headline recipes: timed without the in-step WER)
            cl.mas_importance_loss(model, 0.3).backward()
            cl.importance_accumulate(flat, omega)
        else:                                # LwF: teacher pass with the previous task's weights, student step with the KD terms
            prob_, store = cl.lwf_teacher_forward(model, flat, teacher, batch, langs, host_lengths=host_lens)
            m_.joint.store_sub_enc, m_.joint.detach_sub_enc = True, False
            loss, monitor, prob = model.training_step(batch, langs, return_probs=True, host_lengths=host_lens, compute_wer=False)
            loss, _, _ = cl.lwf_kd_loss(loss, prob, prob_, m_.joint.store_list, store, 0.1, 0.3)
            loss.backward()
            opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events.enabled = True
    timing["on"] = True
    opt.profile_exchange = world > 1       # events around the wait for the gradient all-reduce: exposed exchange time
    from indic_cl_asr_amd.ops import fast as fast_mod
    fast_mod.SYNC_BN_COLLECTIVES = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_dt = time.perf_counter() - t0   # host enqueue time of the timed steps (no sync inside a step)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    events.enabled = False
    timing["on"] = False
    opt.profile_exchange = False
    exchange = None
    if world > 1:
        ex_ms = [a.elapsed_time(b) for a, b in opt.exchange_events]
        exchange = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl": dist.get_backend() == "nccl",
                    "gradient_allreduce_bytes_per_step": int(opt.exchange_bytes), "gradient_dtype": args.grad_exchange,
                    "exposed_wait_ms_per_step": round(sum(ex_ms) / max(1, len(ex_ms)), 4) if ex_ms else None,
                    "syncbn_collectives_per_step": round(fast_mod.SYNC_BN_COLLECTIVES / max(1, args.steps), 1),
                    "note": "one flat all-reduce per step on its own communicator, launched at optimizer.step() and waited for in "
                            "front of the first trainable module of the NEXT forward (under the frozen prefix); exposed = how long "
                            "the compute stream stalls there"}
    # ---- the step the other way round: without the in-step decode + WER when the timed steps have it (what it costs), with it
    # when they were run with --no-wer
    wer_leg = None
    if args.cl == "ewc" and not args.no_wer_leg:
        other = not wer_on
        n3 = max(5, args.steps // 2)
        m_.wer.log_prediction = False; m_.ctc_wer.log_prediction = False       # R/cl_baseline.py:127-128

        def other_step():
            opt.zero_grad()
            l3, mon3 = model.training_step(batch, langs, host_lengths=host_lens, compute_wer=other)
            cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
            l3.backward(); opt.step()
            if other:
                _ = mon3["training_batch_wer"]
            return mon3
        for _ in range(2):
            other_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(n3):
            mon3 = other_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d3 = time.perf_counter() - t3
        if world > 1:
            t = torch.tensor([d3], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d3 = float(t.item())
        wer_leg = {"in_step_wer": other, "value": round(world * args.batch * n3 / d3, 1), "ms_per_step": round(d3 / n3 * 1e3, 3), "steps": n3}
        if other:
            wer_leg["training_batch_wer"] = float(mon3["training_batch_wer"])
            wer_leg["training_batch_wer_ctc"] = float(mon3["training_batch_wer_ctc"])
    # ---- the same step fed from pinned host memory (one batch ahead on a copy stream): PCIe-inclusive rate, never `value`
    h2d = None
    if world == 1 and args.cl == "ewc":
        host_batch = tuple(t.cpu().pin_memory() for t in batch)
        copy = torch.cuda.Stream(device=dev)

        def fetch():
            with torch.cuda.stream(copy):
                devb = tuple(t.to(dev, non_blocking=True) for t in host_batch)
            ev = torch.cuda.Event(); ev.record(copy)
            return devb, ev
        nxt = fetch()
        n2 = max(5, args.steps // 4)
        t1 = 0.0
        for it in range(n2 + 2):
            if it == 2:          # two untimed iterations first: the copy stream's buffers and the pinned staging settle
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            (devb, ev), nxt = nxt, fetch()
            torch.cuda.current_stream(dev).wait_event(ev)
            for t in devb:
                t.record_stream(torch.cuda.current_stream(dev))
            opt.zero_grad()
            l2, mon2 = model.training_step(devb, langs, host_lengths=host_lens, compute_wer=wer_on)
            cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
            l2.backward()
            opt.step()
            if wer_on:
                _ = mon2["training_batch_wer"]
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t1
        h2d = {"value": round(args.batch * n2 / d2, 1), "ms_per_step": round(d2 / n2 * 1e3, 3), "steps": n2,
               "h2d_bytes_per_step": int(sum(t.numel() * t.element_size() for t in host_batch))}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        out = {
            "metric": ("utterances/sec (15 s @16 kHz) Conformer-M RNNT-CTC+EWC train step" if args.cl == "ewc" else
                       f"utterances/sec, {args.cl} step (not the headline metric)"),
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype + ("+fp8 frozen prefix" if args.fp8_prefix else ""), "data": "synthetic",
            "config": {"workload": f"{'BASELINE configs[1]' if (args.cl == 'ewc' and args.preset == 'medium' and args.seconds == 15.0 and args.batch == 32) else 'non-headline configuration (' + args.cl + ')'}: Conformer-{args.preset} (d={cfg.d_model}, {cfg.n_layers}L, "
                                   f"H={cfg.joint_hidden}) hybrid RNNT-CTC + EWC, bs={args.batch}/GPU x {args.seconds:g} s, "
                                   f"{args.dtype} projections, freeze_encoder_till={args.freeze}, 22x257 heads, "
                                   f"fused_batch_size={cfg.fused_batch_size}",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "final_loss": round(float(loss.item()), 4),
                       "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
                       "in_step_wer": ({"enabled": True,
                                        "training_batch_wer": float(last_monitor[0]["training_batch_wer"]),
                                        "training_batch_wer_ctc": float(last_monitor[0]["training_batch_wer_ctc"]),
                                        "note": "the timed steps are the reference-complete step: every batch decoded greedily "
                                                "(compute_wer = True, hybrid_rnnt_ctc_models.py:875) on a side stream beside the joint / "
                                                "backward / optimizer, both WERs scored when the monitor is read -- after every optimizer "
                                                "step, as the reference's loops do; token-level rates on random-initialised weights (no "
                                                "tokenizer / checkpoint in the build)"}
                                       if (wer_on and last_monitor[0] is not None) else {"enabled": False}),
                       "with_h2d_prefetch": h2d, ("without_wer" if wer_on else "with_wer"): wer_leg, "exchange": exchange},
        }
        peaks = None
        if not args.no_peaks:
            try:
                peaks = measure_peaks(dev)
            except Exception as e:   # never lose the line to a microbenchmark
                peaks = {"error": repr(e)}
            out["peaks_measured"] = peaks

        def frac_vs(ach, key, sheet):
            d = {"datasheet": round(ach / sheet, 4)}
            if peaks and key in peaks:
                d["measured"] = round(ach / peaks[key], 4)
            return d
        pmc = None
        s = events.summary()
        if s is not None:
            avg_ms, avg_bytes, n = s
            ach = avg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None
            try:  # HBM bytes per launch from this round's PMC passes of the same command (profiles/r03_pmc_traffic.json)
                pmc_file = "r03_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")) else "r02_pmc_traffic.json"
                pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                if args.batch == 32 and args.seconds == 15.0 and args.preset == "medium":
                    traffic = pmc["kernels"].get(events.kernel_name, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
            out["roofline"] = {"kernel": events.kernel_name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "launches": n, "avg_launch_ms": round(avg_ms, 4),
                               "algorithmic_bytes_per_launch": int(avg_bytes),
                               # the same launch priced on ALL B*T'*(U+1) cells (what the kernel streamed before it left the
                               # tiles behind the utterances' ends alone): for comparison with earlier rounds only
                               "full_lattice_bytes_per_launch": int(avg_bytes * events.full_scale),
                               "frac_if_full_lattice": round(ach * events.full_scale / HBM_PEAK_GBS, 4),
                               "frac_of": frac_vs(ach, "hbm_copy_GBs", HBM_PEAK_GBS)}
        if mfma_events:
            ms_l = [a.elapsed_time(b) for a, b, _ in mfma_events]
            fl = sum(f for _, _, f in mfma_events) / len(mfma_events)
            avg = sum(ms_l) / len(ms_l)
            tf = fl / (avg * 1e-3) / 1e12
            out["roofline_mfma"] = {"kernel": "joint_dh_fused_kernel", "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFS,
                                    "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFS, 4),
                                    "traffic": ((pmc or {}).get("kernels", {}).get("joint_dh_fused_kernel", {}).get("hbm_bytes_per_launch")
                                                if (args.batch == 32 and args.seconds == 15.0 and args.preset == "medium") else None),
                                    "launches": len(ms_l),
                                    "avg_launch_ms": round(avg, 4), "algorithmic_flops_per_launch": int(fl),
                                    "frac_of": frac_vs(tf, "mfma_bf16_TFs", MFMA_BF16_PEAK_TFS)}
        # per-kernel table of this round's committed profile passes (tools/make_rooflines.py: share of the step, bound, achieved,
        # fraction) -- the dominant kernels carry the picture, not one 2 % kernel
        try:
            if args.batch == 32 and args.seconds == 15.0 and args.preset == "medium" and args.cl == "ewc":
                out["rooflines"] = json.load(open(os.path.join(ROOT, "profiles", "r03_rooflines.json")))["kernels"]
        except Exception:
            pass
        if not args.no_cpu_baseline and world == 1:
            kw = dict(d_model=cfg.d_model, n_layers=cfg.n_layers, n_heads=cfg.n_heads, pred_hidden=cfg.pred_hidden,
                      joint_hidden=cfg.joint_hidden)
            out["cpu_baseline"] = cpu_baseline(args.seconds, kw, args.cpu_sample_bs, max(args.freeze, 0), device=dev)
        else:
            out["cpu_baseline"] = None       # (--no-cpu-baseline, or N > 1: rank 0 times it at N = 1 only)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
