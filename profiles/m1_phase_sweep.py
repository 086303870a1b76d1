#!/usr/bin/env python3
"""Where the ~1-3 % `A2_found` + `M1_not_found` losses come from: the SAME burst (payload, level 25 dB, carrier offset) sent at 32 start
times one 32nd of a symbol apart, 24 bursts -- oracle only, baseband at fs/4096.  Prints the loss count per timing phase and, for the
default reading and two alternatives, how many of the 768 decodes are lost."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hfdl_synth as synth
import bench
from oracle import pyoracle

RATE = 40e6 / 4096
rng = np.random.default_rng(5)
sigma = 0.05 / np.sqrt(4096)
plan = []
for i in range(24):
    mode = i % 4
    octets, _ = bench.make_payload(rng, mode)
    plan.append(dict(mode=mode, octets=octets, t0=float(rng.uniform(0.1, 0.2)), amp=0.02, cfo=float(rng.uniform(-15, 15))))
out = {}
for name, fields in (("default", {}), ("symsync_reset_both", dict(symsync_reset_both=1)), ("resamp_float_64", dict(resamp_kind=1))):
    pyoracle.set_variant(**fields)
    lost = np.zeros(32, int)
    a2 = 0
    for i, b in enumerate(plan):
        for k in range(32):
            bb = dict(b, t0=b["t0"] + k / 32.0 / 1800.0)
            n = int((bb["t0"] + synth.burst_symbols_len(b["mode"]) / 1800 + 0.25) * RATE)
            x = synth.synth_channel_baseband(RATE, n, [bb], noise_sigma=sigma, seed=1000 + i)
            ch = pyoracle.Channel(40_000_000, 15_000_000, 15_000_000, want_channelizer=False)
            ch.process_baseband(x)
            s = ch.summary()
            a2 += s["a2_found"]
            lost[k] += 0 if any(p["octets"][:len(b["octets"])] == b["octets"] for p in ch.pdus) else 1
            ch.close()
    out[name] = dict(lost_per_phase=lost.tolist(), lost=int(lost.sum()), decodes=24 * 32, a2_found=a2)
pyoracle.set_variant()
print(json.dumps(out))
