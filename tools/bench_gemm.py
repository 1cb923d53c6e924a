"""Projection GEMM rates at Conformer-large shapes: bf16 MFMA, e4m3 per-row (bf16-rate MFMA), MX block-scaled fp8 (2x-rate MFMA).
Developer tool; numbers in profiles/."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def main():
    from indic_cl_asr_amd import _lib
    from indic_cl_asr_amd.ops import fast
    L = _lib.lib()
    M = 24032
    for N, K in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 256), (256, 1024)):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.1)
        wb = fast.bf16_shadow(w)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * M * N * K
        t_bf = timeit(lambda: fast.gemm(a, wb))
        aq, asc = fast.quantize_fp8_rows(a); wq, wsc = fast.fp8_shadow(w)
        def f8():
            L.ia_gemm_fp8(_lib.ptr(aq), aq.stride(0), _lib.ptr(asc), _lib.ptr(wq), wq.stride(0), _lib.ptr(wsc), M, N, aq.shape[1], None, 0, 0.0,
                          0, 1.0, None, 0, None, 0, _lib.ptr(out), N, _lib.stream_ptr())
        t_f8 = timeit(f8)
        mq, msc = fast.quantize_mxfp8(a); mwq, mwsc = fast.mxfp8_shadow(w)
        def mx():
            L.ia_gemm_mxfp8(_lib.ptr(mq), K, _lib.ptr(msc), msc.stride(0), _lib.ptr(mwq), K, _lib.ptr(mwsc), mwsc.stride(0), M, N, K, None, 0,
                            0.0, 0, 1.0, None, 0, None, 0, _lib.ptr(out), N, _lib.stream_ptr())
        t_mx = timeit(mx)
        t_q = timeit(lambda: fast.quantize_mxfp8(a))
        print(f"[{M} x {K}] x [{K} x {N}]: bf16 {t_bf:7.1f} us {fl / t_bf / 1e6:6.0f} TF/s | e4m3 rows {t_f8:7.1f} us {fl / t_f8 / 1e6:6.0f} TF/s | "
              f"MX {t_mx:7.1f} us {fl / t_mx / 1e6:6.0f} TF/s (+ quantise A {t_q:6.1f} us)", flush=True)


if __name__ == "__main__":
    main()
