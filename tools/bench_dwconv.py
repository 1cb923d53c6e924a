"""GLU + depthwise conv + BN sums kernel at the bench shape (32 x 376 frames, d = 256, k = 31).  Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    for B, T, d, ksz in ((32, 376, 256, 31), (32, 751, 512, 31), (16, 376, 256, 31)):
        x2 = torch.randn(B * T, 2 * d, device="cuda").bfloat16()
        lens = torch.full((B,), T, dtype=torch.long, device="cuda")
        w = torch.randn(d, ksz, device="cuda") * 0.1
        bias = torch.randn(d, device="cuda") * 0.1
        z = torch.empty(B * T, d, device="cuda")
        sums = torch.empty(2 * d, device="cuda")
        scr = torch.empty(L.ia_dwconv_scratch_elems(B, T, d, ksz), device="cuda")

        def run():
            st = L.ia_glu_dwconv(_lib.ptr(x2), _lib.ptr(lens), B, T, d, ksz, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(z), _lib.ptr(sums[:d]),
                                 _lib.ptr(sums[d:]), _lib.ptr(scr), _lib.stream_ptr())
            assert st == 0
        print(f"B {B} T {T} d {d}: glu_dwconv + finish {timeit(run):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
