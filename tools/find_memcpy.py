"""Where the per-step memcpy / memset launches come from (developer tool): a few bench steps under torch.profiler with
stacks; prints every CPU op that has a device memcpy / memset / rocclr copy kernel under it, with its innermost repo frame."""
import collections
import os
import sys

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

    def step():
        opt.zero_grad()
        loss, monitor = model.training_step(batch, langs, host_lengths=host_lens)
        cl.ewc_penalty_into_grads(flat, fisher, checkpoint, e_lambda=10.0)
        loss.backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 2
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    agg = collections.Counter()
    dur = collections.Counter()
    for ev in prof.events():
        kids = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
        hits = [k for k in kids if ("emcpy" in k.name or "emset" in k.name or "rocclr" in k.name)]
        if not hits:
            continue
        frame = "?"
        for fr in (ev.stack or []):
            if "indic_cl_asr_amd" in fr or "bench.py" in fr:
                frame = fr.split("indic_cl_asr_amd/")[-1]
                break
        key = (ev.name, hits[0].name[:40], frame[:90])
        agg[key] += len(hits)
        dur[key] += sum(k.duration for k in hits)
    for key, c in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{c / n:6.1f} {dur[key] / n:8.1f} us  {key[0]:28s} {key[1]:40s} {key[2]}")


if __name__ == "__main__":
    main()
