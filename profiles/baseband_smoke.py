#!/usr/bin/env python3
"""hfdl_gpu_frontend_push_baseband smoke: the oracle's channelizer output through the device's demodulator stage, small traffic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hfdl_synth as synth
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F
from oracle import pyoracle
fs, cf = 250000, 10_000_000
freqs = [9_915_000, 9_972_000, 10_026_000]
bursts = synth.plan_traffic(freqs, 6.0, seed=41, dense=True, gap_s=0.12, amp=(0.02, 0.05))
x = synth.synth_wideband(fs, cf, int(6.0 * fs), bursts, noise_sigma=0.012, seed=41)
fe = hf.Frontend(fs, cf, freqs)
ora = pyoracle.Frontend(fs, cf, freqs)
n, pd, worst = fe.input_size, [], 0.0
t0 = time.time()
for b in range(len(x) // n):
    ora.push_block(x[b * n:(b + 1) * n])
    fe.push_baseband([ora.channel_view(c)["chan_out"] for c in range(len(freqs))])
    pd += fe.poll_pdus()
    for c in range(len(freqs)):
        a, w = fe.read_tap(F.TAP_SYMBOLS, c), ora.channel_view(c)["symbols"]
        assert len(a) == len(w), (b, c, len(a), len(w))
        if len(w):
            worst = max(worst, float(np.max(np.abs(a - w))))
key = lambda p: (p["freq"], p["sample_index"], p["mode"], p["octets"])
print("baseband smoke: %d blocks in %.1f s, %d PDUs (oracle %d), identical %s, worst symbol difference %.3g" %
      (len(x) // n, time.time() - t0, len(pd), len(ora.pdus), sorted(map(key, pd)) == sorted(map(key, ora.pdus)), worst))
