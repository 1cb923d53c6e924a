"""Developer probe: where the time of the reference-complete step (compute_wer = True, monitor read after every optimizer step)
goes on the HOST: per-phase wall-clock marks of the bench's with_wer leg, averaged over the timed steps.
    python tools/probe_wer_leg.py [--steps 20] [--lag]        (--lag: read the monitor of step k after step k + 1 was enqueued)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lag", action="store_true")
    ap.add_argument("--no-wer", action="store_true")
    ap.add_argument("--blank-bias", type=float, default=None, help="added to the head's blank bias (emission density of the decode)")
    ap.add_argument("--sync-only", action="store_true", help="no decode, but a device synchronize where the monitor would be read")
    args = ap.parse_args()
    import bench
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = EncDecHybridRNNTCTCModel(model_config("medium", compute_dtype="bf16")).to(dev)
    freeze_layer(model, 12)
    model.encoder.encoder_frozen_till = 12
    model.train()
    if args.blank_bias is not None:
        with torch.no_grad():
            model.joint.joint_net[-1]['hi'].bias[-1] += args.blank_bias
    flat = cl.FlatParams(model)
    opt = cl.FusedAdamW(flat, lr=1e-4)
    fisher = cl.get_zero_params(model)
    fisher.flat.copy_(torch.rand(flat.numel, device=dev) * 1e-3)
    checkpoint = cl.get_params_clone(model)
    batch, host_lens = bench.synth_batch(32, 15.0, dev, seed=1234)
    langs = ['hi'] * 32
    model.wer.log_prediction = False; model.ctc_wer.log_prediction = False
    want = not (args.no_wer or args.sync_only)
    marks = ["zero_grad", "training_step", "penalty", "backward", "opt.step", "monitor read"]
    acc = [0.0] * len(marks)
    prev = None

    def one(timed):
        nonlocal prev
        t = [time.perf_counter()]
        opt.zero_grad(); t.append(time.perf_counter())
        loss, mon = model.training_step(batch, langs, host_lengths=host_lens, compute_wer=want); t.append(time.perf_counter())
        cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if args.sync_only:
            torch.cuda.synchronize()
        elif want:
            if args.lag:
                if prev is not None:
                    _ = prev["training_batch_wer"]
                prev = mon
            else:
                _ = mon["training_batch_wer"]
        t.append(time.perf_counter())
        if timed:
            for i in range(len(marks)):
                acc[i] += t[i + 1] - t[i]

    if os.environ.get("IA_PROBE_PER_STEP"):      # wall clock of every step from the first one on (device synchronised after each)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            one(False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("per-step ms:", " ".join(f"{t:.2f}" for t in ts), flush=True)
        return
    for _ in range(5):
        one(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if want:
        with torch.no_grad():
            enc, elen = model(input_signal=batch[0], input_signal_length=batch[1])
            hyp = model.decode(enc, elen, langs)
        print("symbols per utterance now: max", max(len(h) for h in hyp), "total", sum(len(h) for h in hyp))
    print(f"{'lag' if args.lag else ('no-wer' if args.no_wer else ('sync-only' if args.sync_only else 'wer'))}: "
          f"{dt / args.steps * 1e3:.3f} ms per step; host ms per phase: "
          + ", ".join(f"{m} {a / args.steps * 1e3:.2f}" for m, a in zip(marks, acc)), flush=True)


if __name__ == "__main__":
    main()
