#!/usr/bin/env python3
"""Experiment driver: run the channelizer alone or the whole pipeline for ~N seconds (for clock / power sampling with rocm-smi)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def run(mode, seconds):
    import torch
    import dumphfdl_amd as hf
    w = bench.WORKLOADS["cfg3"]
    fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w))
    g = fe.geometry
    x = (0.1 * np.random.default_rng(0).standard_normal(2 * 2 * g.input_size)).astype(np.float32)
    dev = torch.from_numpy(x).cuda()
    ptrs = [dev.data_ptr(), dev.data_ptr() + 8 * g.input_size]
    fn = fe.channelize_block if mode == "chan" else fe.push_block
    t_end = time.time() + seconds
    n = 0
    fe.reset_timers(True)
    t0 = time.perf_counter()
    while time.time() < t_end:
        for i in range(64): fn(ptrs[i & 1])
        fe.sync(); n += 64
    el = time.perf_counter() - t0
    ms, k = fe.fold_time_ms()
    print(json.dumps(dict(mode=mode, blocks=n, step_ms=el / n * 1e3, fold_ms=ms / k)))

if __name__ == "__main__":
    run(sys.argv[1], float(sys.argv[2]))
