"""Is the step GPU-bound?  Runs the bench step loop with an artificial host delay per step (busy wait) and prints the step
time for each delay: while the time does not move the GPU is the bottleneck and the host has at least that much slack.
Developer tool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = EncDecHybridRNNTCTCModel(model_config("medium", compute_dtype="bf16")).to(dev)
    freeze_layer(model, 12); model.encoder.encoder_frozen_till = 12
    model.train()
    flat = cl.FlatParams(model)
    opt = cl.FusedAdamW(flat, lr=1e-4)
    fisher = cl.get_zero_params(model)
    fisher.flat.copy_(torch.rand(flat.numel, device=dev) * 1e-3)
    checkpoint = cl.get_params_clone(model)
    batch, host_lens = bench.synth_batch(32, 15.0, dev)
    langs = ['hi'] * 32

    def step(delay_us):
        opt.zero_grad()
        loss, monitor = model.training_step(batch, langs, host_lengths=host_lens)
        cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
        loss.backward()
        opt.step()
        if delay_us:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < delay_us:
                pass

    for _ in range(15):
        step(0)
    for delay in (0, 250, 500, 1000, 2000, 0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            step(delay)
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        print(f"host delay {delay:5d} us/step: {(time.perf_counter() - t0) / n * 1e3:7.3f} ms/step (host loop {host:7.3f} ms/step)", flush=True)


if __name__ == "__main__":
    main()
