#!/usr/bin/env python3
"""The shader clock of fold and demodulator launches IN THE PIPELINE, measured from inside the kernels (laboratory build:
s_memtime against the constant 100 MHz s_memrealtime, hfdl_gpu_lab_clock_probe_read): is a demodulator launch under the fold slow because
it executes more cycles (a neighbour on its SIMD) or because the cycles are slower (the clock the power budget leaves)?

    python profiles/clock_probe.py [cfg3] [blocks]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 160
w = bench.WORKLOADS[wl]
lab = F.load_lab()
fe = F.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w), lib=lab)
g = fe.geometry
fe.enable_taps(False)
x, _ = bench.make_input(w, g.input_size, 0, 1)
nres = len(x) // g.input_size
dev = torch.from_numpy(x.view(np.float32)).cuda()


def read(which):
    buf = (C.c_uint64 * (4 * 4096))()
    n = C.c_int32(0)
    F._check(lab.hfdl_gpu_lab_clock_probe_read(which, buf, 4096, C.byref(n)), lab)
    return [tuple(buf[4 * i + j] for j in range(4)) for i in range(n.value)]


def run(label, blocks, alone_demod=False):
    read(0); read(1)
    for b in range(blocks):
        fe.push_block(dev.data_ptr() + 8 * (b % nres) * g.input_size)
        if alone_demod:
            fe.sync()           # a launch of everything per block: nothing runs beside anything
    fe.sync()
    fold, dem = read(0), read(1)
    print("## %s: %d blocks" % (label, blocks))
    rows = []
    for tag, cyc, ticks, r0 in fold:
        if ticks:
            rows.append(("fold %2d columns, %2d blocks (one workgroup's life)" % (tag // 100, tag % 100), ticks / 100.0, cyc, cyc / ticks * 0.1))
    for tag, cyc, ticks, r0 in dem:
        if ticks:
            rows.append(("demodulator, %d blocks" % (tag - 1000), ticks / 100.0, cyc, cyc / ticks * 0.1))
    # summary by kind: duration, cycles, clock
    kinds = {}
    for k, us, cyc, ghz in rows:
        kinds.setdefault(k, []).append((us, cyc, ghz))
    for k, v in sorted(kinds.items()):
        us = np.array([t[0] for t in v]); cyc = np.array([t[1] for t in v], float); ghz = np.array([t[2] for t in v])
        print("| %s | n %d | us %.0f .. %.0f (median %.0f) | cycles %.0f .. %.0f (median %.0f) | clock GHz %.2f .. %.2f (median %.2f) |"
              % (k, len(v), us.min(), us.max(), np.median(us), cyc.min(), cyc.max(), np.median(cyc), ghz.min(), ghz.max(), np.median(ghz)))
    # demodulator launches, slow against fast half: cycles and clock
    d = [(us, cyc, ghz) for k, us, cyc, ghz in rows if k.startswith("demodulator, %d" % g.demod_batch)]
    if len(d) >= 8:
        d.sort()
        q = len(d) // 4
        fast, slow = d[:q], d[-q:]
        f = lambda part, i: float(np.mean([t[i] for t in part]))
        print("fastest quarter of the %d-block demodulator launches: %.0f us, %.0f cycles, %.2f GHz; slowest quarter: %.0f us, %.0f cycles, %.2f GHz"
              % (g.demod_batch, f(fast, 0), f(fast, 1), f(fast, 2), f(slow, 0), f(slow, 1), f(slow, 2)))
    return rows


run("warm-up", 48)
rows = run("pipeline (blocks pushed without a sync: folds of 16 / 32, demodulators beside them)", nblocks)
run("one block at a time (sync after every block: every kernel alone)", 12, alone_demod=True)
json.dump([dict(kind=k, us=us, cycles=cyc, ghz=ghz) for k, us, cyc, ghz in rows], open(os.path.join(ROOT, "gpurun_out", "clock_probe_%s.json" % wl), "w"))
fe.close()
