"""Greedy transducer decoding of one bench batch (32 x 15 s, Conformer-medium): the device-resident kernel against the
host-driven loop (developer tool; numbers in profiles/)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from indic_cl_asr_amd import decoding as D
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = EncDecHybridRNNTCTCModel(model_config("medium", compute_dtype="bf16")).to(dev).eval()
    lang = 'hi'
    with torch.no_grad():   # random weights emit a label at almost every micro-step: give the head a realistic blank share
        head = model.joint.joint_net[-1][lang]
        head.weight.mul_(3.0); head.bias[-1] += 2.0
    batch, _ = bench.synth_batch(32, 15.0, dev)
    with torch.no_grad():
        enc, enc_len = model.forward(input_signal=batch[0], input_signal_length=batch[1])
    langs = [lang] * 32
    for name, fn in (("device-resident", D.greedy_rnnt_decode_device), ("host-driven", D.greedy_rnnt_decode_host)):
        out = fn(model, enc, enc_len, langs, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3 if name == "device-resident" else 1
        for _ in range(n):
            out = fn(model, enc, enc_len, langs, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:16s} {dt * 1e3:9.1f} ms per batch   frames {int(enc_len.sum())}  symbols {sum(len(o) for o in out)}", flush=True)


if __name__ == "__main__":
    main()
