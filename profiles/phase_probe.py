import os, sys, json
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
for b in range(12):
    fe.push_block(dev.data_ptr() + 8 * b * g.input_size); fe.sync()
    if b >= 8:
        t = np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in range(0, 256, 16)])
        print(b, "cycles R/A/M/S (mean over 16 ch):", t.mean(axis=0).astype(int))
