"""Developer probe of the fused hidden-gradient kernel at the benchmarked shape: time of the full kernel and of its timing-only
variants (IA_DH_DIAG bits: 1 no MFMAs / fragment reads, 2 no epilogue, 4 no G loads, 8 no keep table, 16 no f / g row loads)."""
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
    G = (torch.randn(B * T * U1, LD, generator=g) * 0.01).half().cuda()
    Wt = torch.zeros(H, L.ia_joint_dh_k(), dtype=torch.float16, device="cuda")
    Wt[:, :257] = (torch.randn(H, 257, generator=g) * 0.05).half().cuda()
    f = torch.randn(B, T, H, generator=g).half().cuda()
    gg = torch.randn(B, U1, H, generator=g).half().cuda()
    tld, uld = tl.cuda(), ul.cuda()
    df = torch.zeros(B, T, H, device="cuda")
    dg = torch.zeros(B, U1, H, device="cuda")
    scr = torch.empty(L.ia_joint_dh_fused_scratch_bytes(B, T, U1, H), dtype=torch.uint8, device="cuda")

    def run():
        _lib.check(L.ia_joint_dh_fused(_lib.ptr(G), _lib.ptr(Wt), _lib.ptr(f), _lib.ptr(gg), _lib.ptr(tld), _lib.ptr(uld), _lib.ptr(df),
                                       _lib.ptr(dg), B, T, U1, H, LD, 1.0, p, 7, _lib.ptr(scr), _lib.stream_ptr()), "dh")

    def timeit(n=20):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            run()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    names = {0: "full", 1: "no MFMAs", 2: "no epilogue", 3: "no MFMAs, no epilogue", 4: "no G loads", 8: "no keep table", 16: "no f / g loads",
             24: "no keep table, no f / g loads", 26: "no epilogue, no table, no f / g", 27: "only loads + barriers", 31: "barriers only"}
    for d, nm in names.items():
        os.environ["IA_DH_DIAG"] = str(d)
        print("diag %2d (%s): %.1f us" % (d, nm, timeit()), flush=True)
    os.environ.pop("IA_DH_DIAG", None)


if __name__ == "__main__":
    main()
