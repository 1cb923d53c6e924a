"""Developer micro-benchmark of the fused joint + transducer loss (forward + backward) at BASELINE config-2 shapes
(not the judged bench):  python tools/bench_joint.py [B T U1 H V p reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indic_cl_asr_amd.ops import joint as J
from indic_cl_asr_amd.ops.joint import fused_joint_rnnt
if os.environ.get('IA_FUSED_DW'):
    J.USE_FUSED_DW = os.environ['IA_FUSED_DW'] == '1'
if os.environ.get('IA_FUSED_DH'):
    J.USE_FUSED_DH = os.environ['IA_FUSED_DH'] == '1'


def main():
    B, T, U1, H, V, p, reps = 32, 376, 106, 640, 257, 0.2, 5
    if len(sys.argv) > 1:
        B, T, U1, H, V = map(int, sys.argv[1:6]); p = float(sys.argv[6]); reps = int(sys.argv[7])
    torch.manual_seed(0)
    f = (torch.randn(B, T, H, device="cuda") * 0.7).requires_grad_(True)
    g = (torch.randn(B, U1, H, device="cuda") * 0.7).requires_grad_(True)
    W = (torch.randn(V, H, device="cuda") * 0.1).requires_grad_(True)
    b = torch.zeros(V, device="cuda", requires_grad=True)
    labels = torch.randint(0, V - 1, (B, U1 - 1), device="cuda")
    fl = torch.full((B,), T, device="cuda"); gl = torch.full((B,), U1 - 1, device="cuda")
    def step():
        for t in (f, g, W, b):
            t.grad = None
        fused_joint_rnnt(f, g, W, b, labels, fl, gl, V - 1, dropout_p=p, seed=3).sum().backward()
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record(); torch.cuda.synchronize()
    print(f"fused joint fwd+bwd B{B} T{T} U{U1} H{H} V{V} p{p}: {e0.elapsed_time(e1) / reps:.3f} ms")


if __name__ == "__main__":
    main()
