"""Developer timing of the in-place gradient kernel (ia_joint_backward_g with dbias) at BASELINE config-2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indic_cl_asr_amd import _lib
L = _lib.lib()
B, T, U1, V, LD = 32, 376, 106, 257, 264
dev = "cuda"
logits = (torch.randn(B * T * U1, LD, device=dev) * 2).half()
labels = torch.randint(0, V - 1, (B, U1 - 1), device=dev)
al = torch.full((B,), T, device=dev, dtype=torch.int64); ll = torch.full((B,), U1 - 1, device=dev, dtype=torch.int64)
nbytes = L.ia_rnnt_workspace_bytes(B, T, U1)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
cg = torch.ones(B, device=dev)
db = torch.empty(LD, device=dev); scr = torch.empty(L.ia_joint_backward_g_dbias_scratch_elems(LD), device=dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def ev():
    e = ctypes.c_void_p(); hip.hipEventCreate(ctypes.byref(e)); return e
e0, e1 = ev(), ev()
ts = []
for i in range(12):
    st = L.ia_joint_backward_g(_lib.ptr(logits), _lib.ptr(labels), _lib.ptr(al), _lib.ptr(ll), B, T, U1, V, LD, V - 1, 0.0, _lib.ptr(cg),
                               1.0, None, 0, 0, _lib.ptr(db), _lib.ptr(scr), _lib.ptr(ws), nbytes, _lib.stream_ptr(), e0, e1)
    _lib.check(st, "g")
    torch.cuda.synchronize()
    t = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(t), e0, e1); ts.append(t.value)
print(os.environ.get("IA_LIB_PATH", "default"), f"{sum(ts[2:]) / len(ts[2:]):.4f} ms")
