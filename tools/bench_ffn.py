"""Micro-benchmark of csrc/ffn_fused.hip against the LayerNorm + two-GEMM sequence it replaces (developer probe).
IA_FFN_MODE (environment): 1 = no weight loads inside the loop, 2 = no MFMAs, 3 = both (timing-only builds of the SAME
kernel: outputs are wrong, the durations say where the time goes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indic_cl_asr_amd.ops import fast  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    N, d, dff = 12032, 256, 1024
    torch.manual_seed(0)
    ln, l1, l2, ln2 = torch.nn.LayerNorm(d).cuda(), torch.nn.Linear(d, dff).cuda(), torch.nn.Linear(dff, d).cuda(), torch.nn.LayerNorm(d).cuda()
    x = torch.randn(N, d, device="cuda")

    def unfused():
        y = fast.layernorm(x, ln.weight, ln.bias, ln.eps)
        _, h = fast.gemm(y, fast.bf16_shadow(l1.weight), l1.bias, act=1, dropout_p=0.1, seed=1)
        fast.gemm(h, fast.bf16_shadow(l2.weight), l2.bias, dropout_p=0.1, seed=2, alpha=0.5, residual=x, out_f32=x, want_bf16=False)

    print("unfused LN + 2 GEMM: %.1f us" % timeit(unfused))
    for mode in (0, 1, 2, 3):
        os.environ["IA_FFN_MODE"] = str(mode)
        t = timeit(lambda: fast.ffn_fused(x, ln, l1, l2, 0.5, 0.1, 1, 0.1, 2, ln2=ln2))
        print("ffn_fused mode %d (%s): %.1f us" % (mode, {0: "full", 1: "no loads", 2: "no mfma", 3: "neither", 6: "half the workgroups, no mfma", 8: "contiguous slots", 10: "contiguous slots, no mfma", 14: "contiguous, half the workgroups, no mfma"}[mode], t))
    os.environ["IA_FFN_MODE"] = "0"
    for dff2 in (256, 512, 1024, 2048):   # fixed cost (prologue / epilogue / launch) against the per-chunk cost
        m1, m2 = torch.nn.Linear(d, dff2).cuda(), torch.nn.Linear(dff2, d).cuda()
        t = timeit(lambda: fast.ffn_fused(x, ln, m1, m2, 0.5, 0.1, 1, 0.1, 2, ln2=ln2))
        print("ffn_fused d_ff = %d (%d chunks): %.1f us" % (dff2, dff2 // 128, t))
    t = timeit(lambda: fast.ffn_fused(x, ln, l1, l2, 0.5, 0.0, 1, 0.0, 2))
    print("ffn_fused no dropout, no ln2: %.1f us  -> %.0f TFLOP/s" % (t, 4 * N * d * dff / t / 1e6))


if __name__ == "__main__":
    main()
