#!/usr/bin/env python3
"""Experiment driver: time the channelizer alone (no demod stream) for fold-kernel variants / slice counts on cfg3 geometry."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def run():
    import torch
    import dumphfdl_amd as hf
    w = bench.WORKLOADS[os.environ.get("WL", "cfg3")]
    fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w))
    g = fe.geometry
    x = (0.1 * np.random.default_rng(0).standard_normal(2 * 2 * g.input_size)).astype(np.float32)
    dev = torch.from_numpy(x).cuda()
    ptrs = [dev.data_ptr(), dev.data_ptr() + 8 * g.input_size]
    for i in range(3): fe.channelize_block(ptrs[i & 1])
    fe.reset_timers(True)
    t0 = time.perf_counter()
    for i in range(24): fe.channelize_block(ptrs[i & 1])
    fe.sync(); el = time.perf_counter() - t0
    ms, n = fe.fold_time_ms()
    alg = 8 * g.input_size + g.channels * 8 * g.fft_size + g.channels * 8 * (g.post_input_size // g.post_decimation)
    print(json.dumps(dict(variant=os.environ.get("HFDL_EXP_FOLD", "0"), slices=g.fold_slices, fold_ms=ms / n, frac=alg / (ms / n * 1e-3) / 8e12, step_ms=el / 24 * 1e3)))

if __name__ == "__main__":
    run()
