#!/usr/bin/env python3
"""Is the run-to-run spread of the fold kernel (2.58 .. 2.64 ms on one board) tied to the allocation?  One process creates the
40 Msps / 256-channel front end several times and measures 64 blocks each time."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dumphfdl_amd as hf

fs, cf, nch = 40_000_000, 10_000_000, 256
freqs = [int(cf + (i - nch / 2 + 0.5) * 150_000) for i in range(nch)]
out = []
keep = []
for rep in range(int(os.environ.get("REPS", "5"))):
    fe = hf.Frontend(fs, cf, freqs)
    fe.enable_taps(False)
    n = fe.input_size
    x = torch.randn(2 * n * 4, device="cuda") * 0.05
    ptrs = [x.data_ptr() + 8 * n * b for b in range(4)]
    for i in range(8):
        fe.push_block(ptrs[i % 4])
    fe.poll_pdus()
    fe.reset_timers(True)
    t0 = time.perf_counter()
    for i in range(64):
        fe.push_block(ptrs[i % 4])
    fe.poll_pdus()
    dt = time.perf_counter() - t0
    ms, cnt = fe.fold_time_ms()
    out.append(dict(rep=rep, fold_ms=round(ms / cnt, 4), msps=round(64 * n / dt / 1e6, 1)))
    if os.environ.get("HOLD") and rep % 2 == 0:
        keep.append(torch.empty(int(os.environ["HOLD"]) << 20, dtype=torch.uint8, device="cuda"))   # perturb the next allocation
    fe.close()
    del x
print(json.dumps(out))
