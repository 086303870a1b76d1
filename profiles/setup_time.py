#!/usr/bin/env python3
"""What one rank of the 8-GPU run spends before its timed region: creating the cfg3 front end (256 channels' filter taps designed on
cpus / 8 host threads, as bench.py sets HFDL_GPU_HOST_THREADS at world size 8) and synthesising its 16-block input stream.  Printed as
JSON for profiles/r04_setup_time.json; bench.py refuses to start a run whose slowest rank exceeds HFDL_BENCH_SETUP_BUDGET_S (600 s)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cpus = os.cpu_count() or 1
threads = max(1, cpus // 8)
os.environ["HFDL_GPU_HOST_THREADS"] = str(threads)
import bench            # noqa: E402
import dumphfdl_amd as hf   # noqa: E402

w = bench.WORKLOADS["cfg3"]
t0 = time.time()
fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w))
t_create = time.time() - t0
g = fe.geometry
for f in os.listdir("/tmp"):            # a cold synthesis: no cached stream of this seed
    if f.startswith("hfdl_bench_%s_seed5_" % w["fs"]):
        os.remove(os.path.join("/tmp", f))
t0 = time.time()
x, _ = bench.make_input(w, g.input_size, 0, 8)          # rank 0 of 8: stream seed 5
t_gen = time.time() - t0
fe.close()
print(json.dumps(dict(host_cpus=cpus, host_threads_per_rank=threads, frontend_create_s=round(t_create, 2), input_synthesis_s=round(t_gen, 2),
                      predicted_per_rank_setup_s=round(t_create + t_gen, 2),
                      note="one rank alone on this host; eight ranks run their tap design side by side on cpus / 8 threads each (this figure) and "
                           "their syntheses concurrently (different seeds: no lock shared), so the per-rank figure is the prediction for the job; "
                           "bench.py's budget is HFDL_BENCH_SETUP_BUDGET_S = 600 s",
                      input_blocks=len(x) // g.input_size)))
