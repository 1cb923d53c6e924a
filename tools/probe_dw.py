"""Micro-benchmark of the joint weight-gradient kernel at the benchmarked shape (developer probe): flat 64-cell steps with
the frame counts against the 8 x 8 live tiles with frame and label counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indic_cl_asr_amd import _lib  # noqa: E402


def main():
    L = _lib.lib()
    B, T, U1, H, LD, p = 32, 376, 106, 640, 264, 0.2
    g = torch.Generator().manual_seed(0)
    tl = (torch.rand(B, generator=g) * 0.4 + 0.6).mul(T).long().clamp(1, T)
    ul = (torch.rand(B, generator=g) * 0.4 + 0.6).mul(U1 - 1).long().clamp(0, U1 - 1)
    live = (torch.arange(T).view(1, T, 1, 1) < tl.view(B, 1, 1, 1)) & (torch.arange(U1).view(1, 1, U1, 1) <= ul.view(B, 1, 1, 1))
    G = ((torch.randn(B, T, U1, LD, generator=g) * 0.01) * live).half().view(B * T * U1, LD).contiguous().cuda()
    f = torch.randn(B, T, H, generator=g).half().cuda()
    gg = torch.randn(B, U1, H, generator=g).half().cuda()
    scr = torch.empty(L.ia_joint_dw_fused_scratch_elems(B, T, U1, H, LD), device="cuda")
    tld, uld = tl.cuda(), ul.cuda()
    outs = {}
    for name, ll in (("flat steps, dead frames skipped", None), ("8 x 8 live tiles", uld)):
        dW = torch.empty(LD, H, device="cuda")

        def run():
            _lib.check(L.ia_joint_dw_fused_ex(_lib.ptr(G), _lib.ptr(f), _lib.ptr(gg), _lib.ptr(tld), _lib.ptr(ll), B, T, U1, H, LD, p, 7,
                                              _lib.ptr(dW), _lib.ptr(scr), _lib.stream_ptr()), "dw")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        print("%-34s %.1f us (kernel + finishing sum)" % (name, a.elapsed_time(b) / 20 * 1e3))
        outs[name] = dW.clone()
    v = list(outs.values())
    print("max |difference| / max |dW| = %.2e" % ((v[0] - v[1]).abs().max().item() / v[0].abs().max().item()))
    print("live share of the lattice: frames %.3f, frames x labels %.3f" % (tl.float().mean().item() / T,
          (tl.float() * (ul.float() + 1)).mean().item() / (T * U1)))


if __name__ == "__main__":
    main()
