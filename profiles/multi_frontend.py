#!/usr/bin/env python3
"""K independent cfg2 receivers (front ends) in ONE process on one MI355X, input resident in HBM, blocks pushed round-robin:
aggregate wideband rate.  A single demodulator-bound receiver occupies 96 of 1024 SIMDs; one process's streams run side by side."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import dumphfdl_amd as hf

w = bench.WORKLOADS["cfg2"]
freqs = bench.channel_plan(w)
out = []
for K in (1, 2, 4, 8):
    fes = [hf.Frontend(w["fs"], w["centerfreq"], freqs, device=0) for _ in range(K)]
    g = fes[0].geometry
    x, bursts = bench.make_input(w, g.input_size, 0, 1)
    nblocks = len(x) // g.input_size
    dev = torch.from_numpy(np.array(x).view(np.float32)).cuda()
    ptrs = [dev.data_ptr() + 8 * b * g.input_size for b in range(nblocks)]
    for fe in fes:
        fe.enable_taps(False)
        for b in range(8):
            fe.push_block(ptrs[b % nblocks])
        fe.poll_pdus()
    torch.cuda.synchronize()
    steps = 256
    t0 = time.perf_counter()
    for i in range(steps):
        for fe in fes:
            fe.push_block(ptrs[(8 + i) % nblocks])
    pd = [len(fe.poll_pdus(16384)) for fe in fes]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out.append(dict(receivers=K, aggregate_Msamples_s=round(K * steps * g.input_size / el / 1e6), per_receiver=round(steps * g.input_size / el / 1e6), pdus=pd))
    for fe in fes:
        fe.close()
    del dev
print(json.dumps(out))
