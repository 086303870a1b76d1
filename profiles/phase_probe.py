#!/usr/bin/env python3
"""Demodulator phase cycle counters (R resampler / P whole three-wave pipelined phase, wall / W1 timing-recovery wave busy /
W2 carrier-equaliser-framer wave busy), alone and beside the fold kernel.
Alone: push + sync per block.  Beside the fold: the last call is channelize_block(), which launches the previous block's
held-back demodulator after its forward FFT (so it runs under that block's fold) and queues no demodulator of its own.
  python profiles/phase_probe.py [cfg3|cfg2|cfg4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import torch
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = bench.WORKLOADS[name]
freqs = bench.channel_plan(w)
fe = hf.Frontend(w["fs"], w["centerfreq"], freqs)
g = fe.geometry
x, bursts = bench.make_input(w, g.input_size, 0, 1)
nb = len(x) // g.input_size
dev = torch.from_numpy(x.view(np.float32)).cuda()
ptr = lambda b: dev.data_ptr() + 8 * (b % nb) * g.input_size
chans = range(0, len(freqs), max(1, len(freqs) // 16))
for b in range(4):
    fe.push_block(ptr(b)); fe.sync()
for b in range(4, 7):
    fe.push_block(ptr(b)); fe.sync()
    t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in chans])
    n5400 = len(fe.read_tap(F.TAP_AGC_LEVEL, 0))
    print(name, "alone       block", b, "cycles R/P/W1/W2 (mean over %d ch):" % len(chans), t.mean(axis=0).astype(int), "max", t.max(axis=0).astype(int),
          "samples/block", n5400, "per sample: P %.0f W1 %.0f W2 %.0f" % tuple(t.mean(axis=0)[1:] / max(n5400, 1)))
for rep in range(3):
    for b in range(7, 12):
        fe.push_block(ptr(b))
    L = F.load()
    F._check(L.hfdl_gpu_frontend_channelize_block(fe._h, ptr(12), g.input_size, 1))
    fe.sync()
    t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in chans])
    print(name, "beside fold block 11 cycles R/P/W1/W2 (mean over %d ch):" % len(chans), t.mean(axis=0).astype(int))
