"""256 x 256-tile projection GEMM (csrc/gemm_big.hip) against the 128-row kernel: results and rates at the Conformer-large
shapes (developer probe; IA_GEMM_BIG = 0 / 1 forces the choice)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, n=30):
    for _ in range(4):
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
    shapes = [(24032, 2048, 512), (24032, 1536, 512), (24032, 512, 2048), (24032, 512, 512), (24032, 1024, 512),
              (12032, 1024, 256), (12032, 256, 1024), (12032, 768, 256), (16384, 4096, 4096), (300, 256, 128)]
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in shapes:
        a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
        w = torch.nn.Parameter(torch.randn(N, K, device="cuda", generator=g) * 0.05)
        bias = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g)
        wb = fast.bf16_shadow(w)
        outs, ts = [], []
        for mode in ("0", "1"):
            os.environ["IA_GEMM_BIG"] = mode
            of, oh = fast.gemm(a, wb, bias, act=1, dropout_p=0.1, seed=3, alpha=0.5, residual=res, out_f32=torch.empty_like(res))
            outs.append((of.clone(), oh.clone()))
            ts.append(timeit(lambda: fast.gemm(a, wb)))
        os.environ.pop("IA_GEMM_BIG", None)
        t_auto = timeit(lambda: fast.gemm(a, wb))
        d = (outs[0][0] - outs[1][0]).abs().max().item() / outs[0][0].abs().max().item()
        same16 = (outs[0][1] == outs[1][1]).float().mean().item()
        fl = 2.0 * M * N * K
        print(f"[{M} x {K}] x [{K} x {N}]: 128-row tiles {ts[0]:7.1f} us ({fl / ts[0] / 1e6:6.0f} TF/s) | 256 x 256 tiles {ts[1]:7.1f} us "
              f"({fl / ts[1] / 1e6:6.0f} TF/s) | chosen {t_auto:7.1f} us | max rel diff {d:.1e}, equal bf16 outputs {same16:.4f}", flush=True)


if __name__ == "__main__":
    main()
