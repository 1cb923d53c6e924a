"""Key-tiled attention kernels at the bench shape (32 x 376 frames, 4 heads of 64): forward with / without dropout, backward.
Developer tool."""
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
    from indic_cl_asr_amd.ops import fast
    B, T, H, dk = 32, 376, 4, 64
    d = H * dk
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.8).bfloat16().cuda()
    pl = (torch.randn(2 * T - 1, d, generator=g) * 0.8).bfloat16().cuda()
    bu = (torch.randn(H, dk, generator=g) * 0.3).cuda(); bv = (torch.randn(H, dk, generator=g) * 0.3).cuda()
    lens = torch.round(T * (0.6 + 0.4 * torch.rand(B, generator=g))).long(); lens[0] = T
    lens = lens.cuda()
    dctx = torch.randn(B * T, d, device="cuda").bfloat16()
    for p in (0.0, 0.1):
        t_f = timeit(lambda: fast.relpos_attention_flash(qkv, pl, bu, bv, lens, B, T, H, dk, dropout_p=p, seed=3))
        ctx, lse = fast.relpos_attention_flash(qkv, pl, bu, bv, lens, B, T, H, dk, dropout_p=p, seed=3, want_lse=True)
        t_b = timeit(lambda: fast.relpos_attention_flash_bwd(qkv, pl, bu, bv, lens, ctx, dctx, lse, B, T, H, dk, dropout_p=p, seed=3))
        print(f"dropout {p}: forward {t_f:6.1f} us   backward (all launches) {t_b:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
