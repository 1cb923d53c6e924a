"""Which Python lines still issue small ATen kernels inside the step (developer tool): runs a few bench steps under
torch.profiler with stacks and prints, per (op, innermost repo frame), the calls per step."""
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
    n = 3
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    agg = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace("aten.", "")
            frame = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if ("indic_cl_asr_amd" in fr.filename or fr.filename.endswith("bench.py")) and "find_aten" not in fr.filename:
                    frame = f"{fr.filename.split('indic_cl_asr_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            agg[(name, frame)] += 1
            return func(*args, **(kwargs or {}))

    with Log():
        for _ in range(n):
            step()
    torch.cuda.synchronize()
    skip = ("empty", "view", "_unsafe_view", "reshape", "as_strided", "t.default", "transpose", "permute", "detach", "alias", "select",
            "slice", "expand", "unsqueeze", "squeeze", "split", "unbind", "_local_scalar", "is_pinned", "narrow", "record_stream")
    for (name, frame), c in sorted(agg.items(), key=lambda kv: -kv[1]):
        if any(name.startswith(x) for x in skip):
            continue
        print(f"{c / n:6.1f}  {name:34s} {frame}")


if __name__ == "__main__":
    main()
