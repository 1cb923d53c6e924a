"""Developer probe: library GEMM orientations for the joint's hidden gradient (not part of the product)."""
import torch, time
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cells, LD, H, S = 1275392, 264, 640, 64
Kc = ((cells + S - 1) // S + 63) // 64 * 64
dev = "cuda"
G = torch.randn(cells, LD, device=dev, dtype=torch.float16)
Wp = torch.randn(LD, H, device=dev, dtype=torch.float16)
Wt = Wp.t().contiguous()
GT = torch.randn(S, LD, Kc, device=dev, dtype=torch.float16)
print("mm(G, Wp)            [cells,264]x[264,640]      ", t(lambda: torch.mm(G, Wp)))
print("matmul(Wt, GT)       [640,264]x[S,264,Kc]       ", t(lambda: torch.matmul(Wt, GT)))
print("mm(G, Wt.t())        B given K-contiguous        ", t(lambda: torch.mm(G, Wt.t())))
Gp = torch.randn(cells, 320, device=dev, dtype=torch.float16); Wp3 = torch.randn(320, H, device=dev, dtype=torch.float16)
print("mm(G320, W320)       K padded to 320             ", t(lambda: torch.mm(Gp, Wp3)))
Gb = G.bfloat16(); Wb = Wp.bfloat16()
print("mm bf16                                          ", t(lambda: torch.mm(Gb, Wb)))
GTl = GT.view(S * LD, Kc)
print("bmm(GT^T view) per chunk: [S,Kc,264]x[264,640]   ", t(lambda: torch.matmul(GT.transpose(1, 2), Wp)))
