"""Log-mel front end at the bench shape (32 x 15 s): one-pass FFT kernels vs the GEMM front end.  Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=30):
    for _ in range(3):
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
    from indic_cl_asr_amd.features import mel_filterbank_slaney
    from indic_cl_asr_amd.ops import frontend
    fb = torch.as_tensor(mel_filterbank_slaney()).float().cuda()
    window = torch.hann_window(400, periodic=False).cuda()
    x = (torch.randn(32, 240000, generator=torch.Generator().manual_seed(0)) * 0.1).cuda()
    for mode in ("fft", "gemm"):
        os.environ["IA_FRONTEND"] = mode
        for dither in (0.0, 1e-5):
            t = timeit(lambda: frontend.log_mel(x, window, fb, dither=dither, seed=5))
            print(f"{mode:5s} dither {dither:g}: {t:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
