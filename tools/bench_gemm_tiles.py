"""bf16 projection GEMM: row-tile choice (128 / 96 / 64 rows) at the encoder's shapes.  Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=40):
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
    from indic_cl_asr_amd.ops import fast
    shapes = [(12032, 256, 256), (12032, 512, 256), (12032, 768, 256), (12032, 1024, 256), (12032, 256, 1024),
              (24032, 512, 512), (24032, 1024, 512), (24032, 1536, 512), (24032, 2048, 512), (24032, 512, 2048),
              (3008, 640, 256), (3392, 640, 640), (12032, 256, 5120)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.1)
        wb = fast.bf16_shadow(w)
        row = []
        for bm in ("128", "96", "64", ""):
            if bm:
                os.environ["IA_GEMM_BM"] = bm
            else:
                os.environ.pop("IA_GEMM_BM", None)
            row.append(timeit(lambda: fast.gemm(a, wb)))
        print(f"[{M} x {K}] x [{K} x {N}]: 128 rows {row[0]:6.1f} us | 96 rows {row[1]:6.1f} us | 64 rows {row[2]:6.1f} us | chosen {row[3]:6.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
