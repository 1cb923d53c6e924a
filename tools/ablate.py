"""Step-time ablations on the bench workload (developer tool): which side-stream overlaps pay, what the pieces cost.
    python tools/ablate.py [--steps 30]
Prints ms/step for: the bench configuration; prediction network on the main stream; CTC branch on the main stream; both."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    cfg = model_config("medium", compute_dtype="bf16")
    model = EncDecHybridRNNTCTCModel(cfg).to(dev)
    freeze_layer(model, 12); model.encoder.encoder_frozen_till = 12
    model.train()
    flat = cl.FlatParams(model)
    opt = cl.FusedAdamW(flat, lr=1e-4)
    fisher = cl.get_zero_params(model)
    fisher.flat.copy_(torch.rand(flat.numel, device=dev) * 1e-3)
    checkpoint = cl.get_params_clone(model)
    batch, host_lens = bench.synth_batch(32, 15.0, dev)
    langs = ['hi'] * 32

    def step():
        opt.zero_grad()
        loss, monitor = model.training_step(batch, langs, host_lengths=host_lens)
        cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
        loss.backward()
        opt.step()

    def timed(tag):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{tag:40s} {dt / args.steps * 1e3:7.3f} ms/step  (host enqueue {host / args.steps * 1e3:6.3f})", flush=True)

    timed("bench configuration")
    model.overlap_decoder = False
    timed("prediction network on the main stream")
    model.overlap_decoder, model.overlap_ctc = True, False
    timed("CTC branch on the main stream")
    model.overlap_decoder = False
    timed("both on the main stream")
    model.overlap_decoder = model.overlap_ctc = True
    timed("bench configuration (again)")


if __name__ == "__main__":
    main()
