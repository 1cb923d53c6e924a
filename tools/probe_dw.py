"""Developer probe: split-K factor of the batched weight-gradient GEMM (not part of the product)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 12032
for n, k in ((1024, 256), (256, 1024), (768, 256), (256, 256), (512, 256)):
    dy = torch.randn(M, n, device="cuda", dtype=torch.bfloat16); x = torch.randn(M, k, device="cuda", dtype=torch.bfloat16)
    row = [f"dW[{n}x{k}]"]
    for S in (1, 4, 8, 16, 32, 47):
        if M % S: continue
        if S == 1:
            us = t(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
        else:
            us = t(lambda: torch.bmm(dy.view(S, M // S, n).transpose(1, 2), x.view(S, M // S, k), out_dtype=torch.float32).sum(0))
        row.append(f"S={S}: {us:.1f}us")
    print("  ".join(row))

# the in-tree TN kernel on the same shapes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indic_cl_asr_amd import _lib
L = _lib.lib()
for n, k in ((1024, 256), (256, 1024), (768, 256), (256, 256), (512, 256)):
    dy = torch.randn(M, n, device="cuda", dtype=torch.bfloat16); x = torch.randn(M, k, device="cuda", dtype=torch.bfloat16)
    dW = torch.empty(n, k, device="cuda"); db = torch.empty(n, device="cuda")
    scr = torch.empty(L.ia_gemm_tn_scratch_elems(M, n, k), device="cuda")
    f = lambda: L.ia_gemm_tn_bf16(_lib.ptr(dy), n, _lib.ptr(x), k, M, n, k, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(scr), _lib.stream_ptr())
    print(f"ia_gemm_tn_bf16 dW[{n}x{k}] + db: {t(f):.1f}us")
