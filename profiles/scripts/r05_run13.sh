#!/bin/bash
# round 5, run 13: the opt-in pruned fold with the cumulative-energy criterion at 3e-7: test, then the leg under a few stream settings (lab library)
mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pruned" > gpurun_out/r5m/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5m/pytest.log
tail -30 gpurun_out/r5m/pytest.log
timeout 900 python - > gpurun_out/r5m/legs.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F
w = bench.WORKLOADS["cfg3"]
freqs = bench.channel_plan(w)
g = F.plan_geometry(4096, 250 / w["fs"])
x, bursts = bench.make_input(w, g.input_size, 0, 1)
lab = F.load_lab()
class Lab:            # the leg's `hf`, creating laboratory front ends
    @staticmethod
    def Frontend(*a, **k):
        return hf.Frontend(*a, lib=lab, **k)
for tol in (3e-7, 1e-6):
    bench.PRUNE_TOL = tol
    for env in ({}, {"HFDL_GPU_FOLD_BOUND": "0"}, {"HFDL_GPU_FOLD_BOUND": "0", "HFDL_GPU_DEMOD_BATCH": "2"}, {"HFDL_GPU_FOLD_BOUND": "0", "HFDL_GPU_FOLD_BATCH": "16", "HFDL_GPU_DEMOD_BATCH": "2"},
                {"HFDL_GPU_FOLD_BATCH": "8"}, {"HFDL_GPU_DEMOD_BATCH": "1"}):
        os.environ.update(env)
        try:
            r = bench.pruned_fold_leg(torch, Lab, F, w, freqs, x, 0, 256, 8, [])
        finally:
            for k in env: os.environ.pop(k)
        s = r["streams"]
        print(tol, env, "rows", r["fold_rows"], "value %.0f" % r["value"], "fold %.3f ms" % r["fold_kernel_avg_ms"], "demod %.3f" % r["demod_kernel_ms_per_block"],
              "A %.2f B %.2f D %.2f" % (s["stream_a_ms"], s["stream_b_ms"], s["stream_d_ms"]), "half", s["half_blocks"], "err %.2e" % r["chan_out_rel_rms_vs_full_fold"], "pdus", r["pdus"], flush=True)
        if tol != 3e-7: break
PY
cat gpurun_out/r5m/legs.txt | tail -20
