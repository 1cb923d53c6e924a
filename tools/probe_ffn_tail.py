"""Developer probe: the fused feed-forward launch with and without the q|k|v tail projection, and the projection as its own launch
(12 032 frames, d = 256, d_ff = 1024, 768 columns)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


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
    from indic_cl_asr_amd import _lib
    from indic_cl_asr_amd.ops import fast
    L = _lib.lib()
    N = 12032
    torch.manual_seed(0)
    ln, ln2 = torch.nn.LayerNorm(256).cuda(), torch.nn.LayerNorm(256).cuda()
    l1, l2, qkv = torch.nn.Linear(256, 1024).cuda(), torch.nn.Linear(1024, 256).cuda(), torch.nn.Linear(256, 768).cuda()
    w1, w2, wq = fast.bf16_shadow(l1.weight), fast.bf16_shadow(l2.weight), fast.bf16_shadow(qkv.weight)
    x = torch.randn(N, 256, device="cuda")
    y = torch.empty(N, 256, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(N, 768, dtype=torch.bfloat16, device="cuda")
    common = (_lib.ptr(x), N, 256, 1024, _lib.ptr(ln.weight), _lib.ptr(ln.bias), ln.eps, _lib.ptr(w1), _lib.ptr(l1.bias), _lib.ptr(w2),
              _lib.ptr(l2.bias), 0.5, 0.1, 31, 0.1, 32, _lib.ptr(ln2.weight), _lib.ptr(ln2.bias))
    sp = _lib.stream_ptr()
    t_ffn = timeit(lambda: L.ia_ffn_fused(*common, _lib.ptr(y), 1, sp))
    t_tail = timeit(lambda: L.ia_ffn_fused_tail(*common, None, 1, _lib.ptr(wq), _lib.ptr(qkv.bias), _lib.ptr(out), 768, sp))
    t_gemm = timeit(lambda: L.ia_gemm_bf16(_lib.ptr(y), 256, _lib.ptr(wq), 256, N, 768, 256, _lib.ptr(qkv.bias), 0, 0.0, 0, 1.0, None, 0, None, 0,
                                           _lib.ptr(out), 768, sp))
    def both():
        L.ia_ffn_fused(*common, _lib.ptr(y), 1, sp)
        L.ia_gemm_bf16(_lib.ptr(y), 256, _lib.ptr(wq), 256, N, 768, 256, _lib.ptr(qkv.bias), 0, 0.0, 0, 1.0, None, 0, None, 0, _lib.ptr(out), 768, sp)
    t_both = timeit(both)
    print(f"ffn {t_ffn:.1f} us | ffn + tail {t_tail:.1f} us | projection alone {t_gemm:.1f} us | ffn then projection {t_both:.1f} us")


if __name__ == "__main__":
    main()
