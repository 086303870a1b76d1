#!/usr/bin/env python3
"""Forward FFT of the channelizer (hfdl_gpu_fft_forward): error RMS / signal RMS against float64 numpy and the kernels' time from HIP
events, per transform size.  N >= 2^18 runs the register-resident passes (fft_regs.h), smaller sizes the LDS radix-4 passes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import dumphfdl_amd as hf   # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

rows = []
for logn in (12, 15, 17, 18, 19, 20, 21, 22, 23, 24):
    n = 1 << logn
    rng = np.random.default_rng(logn)
    x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
    hf.fft_forward(x[:n], shifted=True)
    got = hf.fft_forward(x, shifted=True)
    ms = F.last_stage_ms()
    want = np.fft.fftshift(np.fft.fft(x.astype(np.complex128)))
    err = float(np.sqrt(np.mean(np.abs(got - want) ** 2) / np.mean(np.abs(want) ** 2)))
    rows.append(dict(n=n, log2n=logn, rel_rms=err, kernel_ms=ms, GBs=3 * 16 * n / (ms * 1e-3) / 1e9 if ms > 0 else None))
    print("N = 2^%d: rel rms %.3e, three passes %.4f ms = %.0f GB/s of pass traffic" % (logn, err, ms, rows[-1]["GBs"] or 0))
print(json.dumps(rows))
