"""Developer micro-benchmark of the RNNT loss launch at BASELINE config-2 shapes (not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indic_cl_asr_amd.losses.rnnt import rnnt_loss_hip

def main():
    torch.manual_seed(0)
    B, T, U1, V = 32, 376, 106, 257
    if len(sys.argv) > 1:
        B, T, U1, V = map(int, sys.argv[1:5])
    for ragged in (False, True):
        acts = torch.randn(B, T, U1, V, device="cuda")
        labels = torch.randint(0, V - 1, (B, U1 - 1), device="cuda")
        if ragged:
            fl = (T * (0.6 + 0.4 * torch.rand(B, device="cuda"))).long().clamp(1, T); fl[0] = T
            gl = ((U1 - 1) * (0.6 + 0.4 * torch.rand(B, device="cuda"))).long().clamp(0, U1 - 1); gl[0] = U1 - 1
        else:
            fl = torch.full((B,), T, device="cuda"); gl = torch.full((B,), U1 - 1, device="cuda")
        ws = None
        for inplace in (False, True):
            for _ in range(3):
                c, g, ws = rnnt_loss_hip(acts, labels, fl, gl, V - 1, workspace=ws, inplace=inplace)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                c, g, ws = rnnt_loss_hip(acts, labels, fl, gl, V - 1, workspace=ws, inplace=inplace)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            gb = 2 * 4 * B * T * U1 * V / 1e9
            print(f"ragged={ragged} inplace={inplace} B{B} T{T} U{U1} V{V}: {ms:.3f} ms/call  algorithmic {gb:.2f} GB -> {gb/ms:.2f} TB/s")

if __name__ == "__main__":
    main()
