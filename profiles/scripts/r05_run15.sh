#!/bin/bash
# round 5, run 15: which PDU differs between the full and the pruned fold at 256 steps, and what the oracle says about it
mkdir -p gpurun_out/r5o
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pruned" > gpurun_out/r5o/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5o/pytest.log
tail -12 gpurun_out/r5o/pytest.log
timeout 1200 python - > gpurun_out/r5o/diff.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
import dumphfdl_amd as hf
from dumphfdl_amd import frontend as F
from oracle import pyoracle
w = bench.WORKLOADS["cfg3"]
freqs = bench.channel_plan(w)
g = F.plan_geometry(4096, 250 / w["fs"])
x, bursts = bench.make_input(w, g.input_size, 0, 1)
nblocks = len(x) // g.input_size
by_freq = {}
for b in bursts: by_freq.setdefault(b["freq"], []).append(b)
dev = torch.from_numpy(x.view(np.float32)).cuda()
def run(tol, steps=256, warmup=8):
    if tol: os.environ["HFDL_GPU_FOLD_PRUNE"] = repr(tol)
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs); fe.enable_taps(False)
    os.environ.pop("HFDL_GPU_FOLD_PRUNE", None)
    step = 0
    for _ in range(warmup):
        fe.push_block(dev.data_ptr() + 8 * (step % nblocks) * g.input_size); step += 1
    fe.poll_pdus()
    el, raw, step = bench.timed_blocks(torch, fe, lambda i: fe.push_block(dev.data_ptr() + 8 * i * g.input_size), steps, step, nblocks)
    p = [q for buf, n in raw for q in fe.pdus_to_dicts(buf, n)]
    rows = fe.geometry.fold_rows
    fe.close()
    return p, rows
key = lambda p: (p["freq"], p["mode"], p["octets"], p["fcs_status"])
ref, _ = run(0)
print("resident blocks", nblocks, "full:", len(ref), "PDUs; matching sent", sum(bench.matches_sent(p, by_freq) for p in ref))
odd_freqs = set()
for tol in (1e-7, 3e-7, 1e-6):
    got, rows = run(tol)
    from collections import Counter
    a, b = Counter(key(p) for p in ref), Counter(key(p) for p in got)
    only_full, only_pruned = list((a - b).elements()), list((b - a).elements())
    print("tol %g rows %d: %d PDUs, matching sent %d; only in full %d, only in pruned %d" % (tol, rows, len(got), sum(bench.matches_sent(p, by_freq) for p in got), len(only_full), len(only_pruned)))
    for k in only_full:
        q = [p for p in ref if key(p) == k][0]
        print("   only full  :", q["freq"], q["channel"], q["mode"], q["sample_index"], "fcs", q["fcs_status"], "sent", bench.matches_sent(q, by_freq)); odd_freqs.add(q["freq"])
    for k in only_pruned:
        q = [p for p in got if key(p) == k][0]
        print("   only pruned:", q["freq"], q["channel"], q["mode"], q["sample_index"], "fcs", q["fcs_status"], "sent", bench.matches_sent(q, by_freq)); odd_freqs.add(q["freq"])
sys.stdout.flush()
# the oracle over the same block sequence on the channels in question
odd = sorted(odd_freqs)[:4]
if odd:
    ora = pyoracle.Frontend(w["fs"], w["centerfreq"], odd, nthreads=8)
    for s in range(8 + 256):
        i = s % nblocks
        ora.push_block(x[i * g.input_size:(i + 1) * g.input_size], nthreads=8)
    op = [p for p in ora.pdus]
    print("oracle on", odd, ":", len(op), "PDUs over the warm-up and the timed blocks")
    for f in odd:
        print("  freq", f, "oracle:", sorted((p["sample_index"], p["mode"]) for p in op if p["freq"] == f))
        print("           full  :", sorted((p["sample_index"], p["mode"]) for p in ref if p["freq"] == f))
PY
cat gpurun_out/r5o/diff.txt | tail -30
