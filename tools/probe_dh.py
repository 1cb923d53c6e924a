"""Developer probe: one call of the fused hidden-gradient kernel with IA_DEBUG diagnostics on stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IA_DEBUG"] = "1"
import torch
from indic_cl_asr_amd import _lib
L = _lib.lib()
B, T, U1, H, LD = 2, 20, 9, 640, 264
dev = "cuda"
G = torch.zeros(B * T * U1, LD, dtype=torch.float16, device=dev)
Wt = torch.zeros(H, L.ia_joint_dh_k(), dtype=torch.float16, device=dev)
f = torch.zeros(B, T, H, dtype=torch.float16, device=dev); g = torch.zeros(B, U1, H, dtype=torch.float16, device=dev)
al = torch.tensor([20, 11], device=dev); ll = torch.tensor([8, 3], device=dev)
df = torch.zeros(B, T, H, device=dev); dg = torch.zeros(B, U1, H, device=dev)
scr = torch.empty(L.ia_joint_dh_fused_scratch_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
st = L.ia_joint_dh_fused(_lib.ptr(G), _lib.ptr(Wt), _lib.ptr(f), _lib.ptr(g), _lib.ptr(al), _lib.ptr(ll), _lib.ptr(df), _lib.ptr(dg), B, T, U1, H, LD,
                         1.0, 0.2, 1, _lib.ptr(scr), _lib.stream_ptr())
torch.cuda.synchronize()
print("status", st)
