#!/usr/bin/env python3
"""A/B of libhfdl_gpu.so builds on the demodulator-bound workload (cfg2: 8 Msps x 32 channels), without torch: device memory
through libamdhip64 by ctypes, one child process per build (the library is chosen at import by HFDL_GPU_LIB), the input synthesised
once.  Per build: wideband rate (wall), the demodulator kernel's time per block from its own dispatch events, the PDU count and a
CRC over every PDU (freq, sample_index, mode, octets) -- equal CRCs = the builds decode the same thing.

  python profiles/ab_demod.py [--steps 256] [--rounds 2] base /root/repo/dumphfdl_amd/libhfdl_gpu_x.so ...
"""
import ctypes
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402

INPUT = "/tmp/ab_demod_input.npy"


def child(steps, workload):
    import bench
    import dumphfdl_amd as hf
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    w = bench.WORKLOADS[workload]
    freqs = bench.channel_plan(w)
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=0)
    g = fe.geometry
    fe.enable_taps(False)
    x = np.load(INPUT, mmap_mode="r")
    x = np.ascontiguousarray(x)
    nblocks = len(x) // g.input_size
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), x.nbytes) == 0
    assert hip.hipMemcpy(dev, x.ctypes.data, x.nbytes, 1) == 0
    ptrs = [dev.value + 8 * b * g.input_size for b in range(nblocks)]
    pdus = []
    for b in range(8):
        fe.push_block(ptrs[b % nblocks])
    pdus += fe.poll_pdus()
    fe.reset_timers(True)
    t0 = time.perf_counter()
    for i in range(steps):
        fe.push_block(ptrs[(8 + i) % nblocks])
        if i % 256 == 255 and i + 1 < steps:
            pdus += fe.poll_pdus(16384, max_in_flight=1)
    pdus += fe.poll_pdus(16384)
    el = time.perf_counter() - t0
    ms, launches, blocks = fe.demod_time_ms()
    crc = 0
    for p in sorted((p["freq"], p["sample_index"], p["mode"], p["octets"]) for p in pdus):
        crc = zlib.crc32(repr(p).encode(), crc)
    print(json.dumps(dict(lib=os.environ.get("HFDL_GPU_LIB", "base"), Msamples_s=round(steps * g.input_size / el / 1e6, 1),
                          demod_ms_per_block=round(ms / max(blocks, 1), 5), launches=launches, batch=g.demod_batch, pdus=len(pdus), pdu_crc="%08x" % crc)))
    fe.close()


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.steps, a.workload)
    import bench
    import dumphfdl_amd as hf
    w = bench.WORKLOADS[a.workload]
    fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w)[:1], device=0)
    n = fe.geometry.input_size
    fe.close()
    x, _ = bench.make_input(w, n, 0, 1)
    np.save(INPUT, x)
    for r in range(a.rounds):
        for lib in a.libs or ["base"]:
            env = dict(os.environ)
            if lib == "base":
                env.pop("HFDL_GPU_LIB", None)
            else:
                env["HFDL_GPU_LIB"] = lib
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(a.steps), "--workload", a.workload], env=env,
                                 capture_output=True, text=True, timeout=300)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr[-400:]
            print(line, flush=True)


if __name__ == "__main__":
    main()
