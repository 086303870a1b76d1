#!/usr/bin/env python3
"""Which of the fold's resources slows the demodulator down?  The demodulator of cfg3, one block per launch and a sync after every
block (nothing of the pipeline beside it), timed from inside the kernel (laboratory build: s_memtime / s_memrealtime) while a SYNTHETIC
neighbour (profiles/micro/neighbour.hip: four waves on every CU using one of the fold's resources at the fold's rate) runs on a stream
of its own.

    python profiles/neighbour_probe.py [cfg3] [ms per mode]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
ms_mode = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
os.environ.setdefault("HFDL_GPU_DEMOD_BATCH", "1")
nb = C.CDLL(os.path.join(ROOT, "profiles", "micro", "libneighbour.so"))
nb.neighbour_start.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_int]
nb.neighbour_wait.restype = C.c_double
w = bench.WORKLOADS[wl]
lab = F.load_lab()
fe = F.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w), lib=lab)
g = fe.geometry
fe.enable_taps(False)
x, _ = bench.make_input(w, g.input_size, 0, 1)
nres = len(x) // g.input_size
dev = torch.from_numpy(x.view(np.float32)).cuda()
src = torch.empty(1 << 30, dtype=torch.float32, device="cuda").normal_()      # 4 GiB: past the 256 MiB Infinity Cache

MODES = [
    (0, "nothing beside it"),
    (1, "matrix pipe only (64 MFMA per quad, back to back)"),
    (1000, "matrix pipe only, inline-asm instructions, back to back"),
    (2000, "matrix pipe only, the short form: v_mfma_f32_4x4x1_16B_f32 (2 passes) back to back, the same multiply-accumulates per quad"),
    (3000, "the bf16 matrix instruction (v_mfma_f32_16x16x32_bf16) back to back"),
    (1008, "matrix pipe only, 8 idle cycles of the issuing wave behind every instruction"),
    (1016, "matrix pipe only, 16 idle cycles behind every instruction"),
    (1020, "matrix pipe only, 20 idle cycles behind every instruction"),
    (1024, "matrix pipe only, 24 idle cycles behind every instruction"),
    (1028, "matrix pipe only, 28 idle cycles behind every instruction"),
    (33, "matrix pipe + 16 DPP moves per quad"),
    (2, "LDS reads (16 x b128 per quad), paced"),
    (4, "LDS writes (4 x b128 per quad, the fold's 4-way addresses), paced"),
    (14, "LDS reads + writes + barrier, paced"),
    (16, "HBM reads (5 KiB per quad and wave), paced"),
    (15, "matrix pipe + LDS reads + writes + barrier"),
    (31, "matrix pipe + LDS + barrier + HBM reads"),
    (63, "all of it (the synthetic fold)"),
    (66, "LDS reads flat out"),
    (68, "LDS writes flat out"),
    (80, "HBM reads flat out"),
    (96, "DPP moves flat out (a vector-ALU wave on every SIMD)"),
]


def read(which):
    buf = (C.c_uint64 * (4 * 4096))()
    n = C.c_int32(0)
    F._check(lab.hfdl_gpu_lab_clock_probe_read(which, buf, 4096, C.byref(n)), lab)
    return [tuple(buf[4 * i + j] for j in range(4)) for i in range(n.value)]


# warm-up: the pipeline's own allocations, the clocks
for b in range(8):
    fe.push_block(dev.data_ptr() + 8 * (b % nres) * g.input_size)
    fe.sync()
out = []
print("| neighbour | demodulator launches | us (median) | cycles (median) | min .. max cycles | clock GHz | x alone | launch ms (dispatch events) | wall ms per block (whole pipeline, synced) | neighbour's quads per ms and workgroup (64 matrix instructions per quad and wave: 1172 = the pipe full at 2.4 GHz) |")
print("|---|---|---|---|---|---|---|---|---|---|")
base = None
for mode, label in MODES:
    read(0); read(1)
    fe.reset_timers(True)
    if mode:
        rc = nb.neighbour_start(mode, src.data_ptr(), src.numel() * 4, ms_mode + 60.0, 256)
        assert rc == 0, rc
        time.sleep(0.005)
    t0 = time.perf_counter()
    b = 0
    while (time.perf_counter() - t0) * 1e3 < ms_mode and b < 64:
        fe.push_block(dev.data_ptr() + 8 * (b % nres) * g.input_size)
        fe.sync()
        b += 1
    wall_ms = (time.perf_counter() - t0) * 1e3
    dem = [(cyc, ticks) for tag, cyc, ticks, r0 in read(1) if ticks]
    ev_ms, ev_launches, ev_blocks = fe.demod_time_ms()
    quads = nb.neighbour_wait() if mode else 0.0
    quads_per_ms = quads / (ms_mode + 60.0)
    cyc = np.array([d[0] for d in dem], float); us = np.array([d[1] for d in dem], float) / 100.0
    if base is None:
        base = float(np.median(cyc))
    print("| %s | %d | %.0f | %.0f | %.0f .. %.0f | %.2f | %.2f | %.3f | %.2f | %.0f |" % (label, len(dem), np.median(us), np.median(cyc), cyc.min(), cyc.max(),
          float(np.median(cyc / us)) * 1e-3, float(np.median(cyc)) / base, ev_ms / max(ev_blocks, 1), wall_ms / max(b, 1), quads_per_ms), flush=True)
    out.append(dict(mode=mode, label=label, launches=len(dem), us_median=float(np.median(us)), cycles_median=float(np.median(cyc)),
                    cycles_min=float(cyc.min()), cycles_max=float(cyc.max()), x_alone=float(np.median(cyc)) / base,
                    event_ms_per_block=ev_ms / max(ev_blocks, 1), wall_ms_per_block=wall_ms / max(b, 1), neighbour_quads_per_ms=quads_per_ms))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "neighbour_probe_%s.json" % wl), "w"), indent=1)
fe.close()
