#!/usr/bin/env python3
"""Characterisation run: long noise-only stretches (exercise the 13-frame timeout reset and carrier-runaway reset paths,
src/hfdl.c:711-715, 745-752) followed by one burst per channel; GPU vs oracle counters and PDUs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dumphfdl_amd as hf
import hfdl_synth as synth
from oracle import pyoracle

fs, cf = 250_000, 10_000_000
freqs = [int(cf + (i - 4) * 27_000 + 5_000) for i in range(8)]
dur = float(os.environ.get("SOAK_S", "75"))
rng = np.random.default_rng(5)
bursts = [dict(freq=f, mode=int(i % 4), octets=synth.make_pdu(rng, int(i % 4)), t0=dur - 3.0, amp=0.02, cfo=float(rng.uniform(-10, 10))) for i, f in enumerate(freqs)]
fe = hf.Frontend(fs, cf, freqs)
ora = pyoracle.Frontend(fs, cf, freqs, nthreads=8)
n = fe.input_size
chunk = 40 * n
done = 0
total = int(dur * fs) // n * n
seed = 0
while done < total:
    m = min(chunk, total - done)
    t_off = done / fs
    bl = [dict(b, t0=b["t0"] - t_off) for b in bursts if b["t0"] - t_off < m / fs + 1 and b["t0"] - t_off > -4]
    x = synth.synth_wideband(fs, cf, m, bl, noise_sigma=0.01, seed=100 + seed); seed += 1
    for b in range(m // n):
        fe.push_block(x[b * n:(b + 1) * n]); ora.push_block(x[b * n:(b + 1) * n], nthreads=8)
    done += m
got = sorted((p["freq"], p["sample_index"], p["octets"]) for p in fe.poll_pdus())
want = sorted((p["freq"], p["sample_index"], p["octets"]) for p in ora.pdus)
cg = [fe.channel_stats(c) for c in range(8)]
co = [ora.channel_counters(c) for c in range(8)]
oct_same = sorted((a[0], a[2]) for a in got) == sorted((a[0], a[2]) for a in want)
didx = {g[0]: g[1] - w[1] for g in got for w in want if g[0] == w[0]}
print(json.dumps(dict(seconds=dur, octets_identical=oct_same, sample_index_delta=didx, pdus_gpu=len(got), pdus_oracle=len(want), identical=got == want,
                      a2_gpu=[c["a2_found"] for c in cg], a2_oracle=[c["a2_found"] for c in co],
                      m1nf_gpu=[c["m1_not_found"] for c in cg], m1nf_oracle=[c["m1_not_found"] for c in co],
                      nf_db_gpu=[round(c["noise_floor_db"], 2) for c in cg], nf_db_oracle=[round(20 * np.log10(c["noise_floor"]), 2) for c in co])))
