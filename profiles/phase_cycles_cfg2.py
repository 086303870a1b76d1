#!/usr/bin/env python3
"""Demodulator phase cycles per 5400-sps sample on cfg2 without torch (taps on, a launch per block): R resampler, P the three-wave
pipelined phase (wall), W1 timing-recovery wave busy, W2 carrier wave busy (mean over channels; its slowest channel bounds the launch).
HFDL_GPU_LIB selects the build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F
w = bench.WORKLOADS["cfg2"]
freqs = bench.channel_plan(w)
fe = hf.Frontend(w["fs"], w["centerfreq"], freqs)
g = fe.geometry
x, _ = bench.make_input(w, g.input_size, 0, 1)
nb = len(x) // g.input_size
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), x.nbytes) == 0 and hip.hipMemcpy(dev, x.ctypes.data, x.nbytes, 1) == 0
ptr = lambda b: dev.value + 8 * (b % nb) * g.input_size
chans = range(len(freqs))
for b in range(10):
    fe.push_block(ptr(b)); fe.sync()
    if b < 4:
        continue
    t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in chans])
    n5400 = len(fe.read_tap(F.TAP_AGC_LEVEL, 0))
    nsym = np.array([len(fe.read_tap(F.TAP_SYMBOLS, c)) for c in chans])
    m = t.mean(axis=0) / n5400
    print("block %d samples %d symbols %.0f per sample: R %.0f P %.0f W1 %.0f W2 %.0f | per channel W2 min %.0f max %.0f, W1 max %.0f" %
          (b, n5400, nsym.mean(), m[0], m[1], m[2], m[3], t[:, 3].min() / n5400, t[:, 3].max() / n5400, t[:, 2].max() / n5400))
