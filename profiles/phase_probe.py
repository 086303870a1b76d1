#!/usr/bin/env python3
"""Demodulator phase cycle counters (R resampler / A AGC / M matched filter / S symbol loop), alone and beside the fold kernel.
Alone: push + sync per block.  Beside the fold: the last call is channelize_block(), which launches the previous block's
held-back demodulator after its forward FFT (so it runs under that block's fold) and queues no demodulator of its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import torch
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F

w = bench.WORKLOADS["cfg3"]
fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w))
g = fe.geometry
x, bursts = bench.make_input(w, g.input_size, 0, 1)
dev = torch.from_numpy(x.view(np.float32)).cuda()
ptr = lambda b: dev.data_ptr() + 8 * (b % 16) * g.input_size
chans = range(0, 256, 16)
for b in range(4):
    fe.push_block(ptr(b)); fe.sync()
for b in range(4, 7):
    fe.push_block(ptr(b)); fe.sync()
    t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in chans])
    print("alone       block", b, "cycles R/A/M/S (mean over 16 ch):", t.mean(axis=0).astype(int))
for rep in range(3):
    for b in range(7, 12):
        fe.push_block(ptr(b))
    L = F.load()
    F._check(L.hfdl_gpu_frontend_channelize_block(fe._h, ptr(12), g.input_size, 1))
    fe.sync()
    t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in chans])
    print("beside fold block 11 cycles R/A/M/S (mean over 16 ch):", t.mean(axis=0).astype(int))
