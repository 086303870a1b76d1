#!/usr/bin/env python3
"""Carrier-wave cycles and timing-recovery outputs per framer state (search / equaliser training / data), per channel, from two
-DHFDL_DM_PROBE=2 / =3 builds run on the same blocks."""
import ctypes, os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import bench
    import dumphfdl_amd as hf
    from dumphfdl_amd import frontend as F
    w = bench.WORKLOADS["cfg2"]
    freqs = bench.channel_plan(w)
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs)
    g = fe.geometry
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    nb = len(x) // g.input_size
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), x.nbytes) == 0 and hip.hipMemcpy(dev, x.ctypes.data, x.nbytes, 1) == 0
    tot = np.zeros((len(freqs), 4))
    for b in range(nb):
        fe.push_block(dev.value + 8 * b * g.input_size); fe.sync()
        tot += np.array([fe.read_tap(F.TAP_PHASE_CYCLES, c) for c in range(len(freqs))])
    print(json.dumps(tot.tolist()))
    sys.exit(0)
out = {}
for v in (2, 3):
    env = dict(os.environ, HFDL_GPU_LIB="/root/repo/dumphfdl_amd/libhfdl_gpu_probe%d.so" % v)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    out[v] = np.array(json.loads(r.stdout.strip().splitlines()[-1]))
cyc, cnt = out[2], out[3]
print("whole stream, per channel: outputs in search / training / data, cycles per output in each, W2 busy cycles total")
for c in range(len(cyc)):
    print("ch %2d outputs %6d %6d %6d  cycles/output %5.0f %5.0f %5.0f   W2 %9.0f  in-state share %.2f" % (
        c, cnt[c, 0], cnt[c, 1], cnt[c, 2], cyc[c, 0] / max(cnt[c, 0], 1), cyc[c, 1] / max(cnt[c, 1], 1), cyc[c, 2] / max(cnt[c, 2], 1), cyc[c, 3], cyc[c, :3].sum() / cyc[c, 3]))
print("mean cycles/output: search %.0f training %.0f data %.0f" % tuple(cyc[:, :3].sum(axis=0) / cnt[:, :3].sum(axis=0)))
