#!/usr/bin/env python3
"""What the board draws and clocks while ONE kind of work loops for some seconds -- sampled with rocm-smi from a thread of this process
(sclk, average socket power) -- to put numbers behind the fold's power roofline (profiles/r06_experiments.md):

  fold4 / fold16 / fold32   the product tiling of that form, alone, launch after launch (laboratory probe, the front end's own taps)
  stream                    a bare read-only pass over the same 16 GiB (laboratory probe)
  pipeline                  the whole front end on resident input, 64 blocks per sync
  idle                      nothing

    python profiles/power_probe.py <mode> [seconds]
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

mode = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
samples = []
stop = False
first_raw = []


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            if not first_raw:
                first_raw.append(out[-700:])
            sclk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
            mclk = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", out)
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            samples.append((time.time(), int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None, float(pw.group(1)) if pw else None))
        except Exception as e:      # noqa: BLE001
            samples.append((time.time(), None, None, None, str(e)))
        time.sleep(0.25)


w = bench.WORKLOADS["cfg3"]
os.environ.setdefault("HFDL_GPU_FOLD_BATCH", "32")
fe = F.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w), lib=F.load_lab())
g = fe.geometry
fe.enable_taps(False)
x = (0.05 * np.random.default_rng(0).standard_normal(2 * 16 * g.input_size)).astype(np.float32)
dev = torch.from_numpy(x).cuda()
for b in range(32):
    fe.push_block(dev.data_ptr() + 8 * (b % 16) * g.input_size)
fe.poll_pdus()
variants = F.fold_variants()
pick = {"fold4": next(i for i, v in enumerate(variants) if v[4] == 4), "fold16": next(i for i, v in enumerate(variants) if v[4] == 16),
        "fold32": next(i for i, v in enumerate(variants) if v[4] == 32)}
th = threading.Thread(target=sampler, daemon=True)
t_start = time.time()
th.start()
done, ms_sum, launches = 0, 0.0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    if mode in pick:
        nb = {"fold4": 4, "fold16": 16, "fold32": 32}[mode]
        avg, best, _ = fe.fold_variant_probe(pick[mode], nb, 20)
        ms_sum += avg * 20
        launches += 20
    elif mode == "stream":
        fe.stream_read_probe()
        launches += 1
    elif mode == "pipeline":
        for i in range(64):
            fe.push_block(dev.data_ptr() + 8 * (i % 16) * g.input_size)
        fe.sync()
        done += 64
    else:
        time.sleep(0.5)
el = time.time() - t_start
stop = True
th.join(timeout=5)
ok = [s for s in samples if len(s) == 4 and s[1] is not None and s[0] > t_start + 1.5]
res = dict(mode=mode, seconds=round(el, 1), samples=len(ok),
           sclk_mhz_avg=round(sum(s[1] for s in ok) / len(ok)) if ok else None, sclk_mhz_min=min((s[1] for s in ok), default=None),
           mclk_mhz=ok[0][2] if ok else None,
           power_w_avg=round(sum(s[3] for s in ok if s[3]) / max(1, sum(1 for s in ok if s[3])), 1) if ok else None,
           power_w_max=max((s[3] for s in ok if s[3]), default=None))
if mode in pick:
    res["avg_launch_ms"] = round(ms_sum / max(launches, 1), 3)
if mode == "pipeline":
    res["ms_per_block"] = round(el / max(done, 1) * 1e3, 4)
if not ok:
    res["raw"] = [s for s in samples[:3]]
    res["rocm_smi_said"] = first_raw[:1]
print(json.dumps(res))
fe.close()
