#!/usr/bin/env python3
"""Host -> device rate of the front end's copy stream as a function of WHEN in the process the front end is created (which hardware
queue / copy engine the runtime hands its streams): the cfg2 host-RAM leg on a first, second and third front end, with and without
torch having touched the device first."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

use_torch_first = "--torch-first" in sys.argv
import torch
if use_torch_first:
    torch.zeros(1 << 20).cuda()
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F
w = bench.WORKLOADS["cfg2"]
freqs = bench.channel_plan(w)
out = []
keep = []
for i in range(3):
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=0)
    fe.enable_taps(False)
    g = fe.geometry
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    nblocks = len(x) // g.input_size
    hbuf, leg = bench.host_ram_leg(torch, hf, F, fe, x, g, nblocks, 256)
    out.append(dict(order=i, torch_first=use_torch_first, value=round(leg["value"]), pcie_GBs=round(leg["pcie_GBs"], 1)))
    if i == 1:
        keep.append((fe, hbuf))             # the third front end is created while the second is alive
    else:
        fe.close(); hf.host_free(hbuf)
print(json.dumps(out))
