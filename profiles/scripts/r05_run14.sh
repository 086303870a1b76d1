#!/bin/bash
# round 5, run 14: pruned fold with contiguous windows: test; distance to the oracle with and without; the leg at 256 and 20 steps
mkdir -p gpurun_out/r5n
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pruned" > gpurun_out/r5n/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5n/pytest.log
tail -12 gpurun_out/r5n/pytest.log
timeout 900 python - > gpurun_out/r5n/legs.txt 2>&1 <<'PY'
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
sub = [3, 77, 128, 250]
ora = pyoracle.Frontend(w["fs"], w["centerfreq"], [freqs[c] for c in sub], nthreads=8)
ora.push_block(x[:g.input_size], nthreads=8)
want = [ora.channel_view(i)["chan_out"].astype(np.complex128) for i in range(len(sub))]
rel = lambda a, b: float(np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2)))
outs = {}
for tol in (0, 1e-7, 2.5e-7, 3e-7, 5e-7, 1e-6):
    if tol: os.environ["HFDL_GPU_FOLD_PRUNE"] = repr(tol)
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs)
    os.environ.pop("HFDL_GPU_FOLD_PRUNE", None)
    fe.channelize_block(x[:g.input_size])
    outs[tol] = [fe.read_tap(F.TAP_CHAN_OUT, c).astype(np.complex128) for c in sub]
    print("tol %g: fold_rows %d of %d; against the oracle %s; against the full fold %s" % (tol, fe.geometry.fold_rows, fe.geometry.pre_decimation,
          ["%.2e" % rel(a, b) for a, b in zip(outs[tol], want)], ["%.2e" % rel(a, b) for a, b in zip(outs[tol], outs[0])]), flush=True)
    fe.close()
# the full run's PDUs as the reference of the leg
def full_pdus(steps, warmup):
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs); fe.enable_taps(False)
    nblocks = len(x) // g.input_size
    dev = torch.from_numpy(x.view(np.float32)).cuda()
    step = 0
    for _ in range(warmup):
        fe.push_block(dev.data_ptr() + 8 * (step % nblocks) * g.input_size); step += 1
    fe.poll_pdus()
    el, raw, step = bench.timed_blocks(torch, fe, lambda i: fe.push_block(dev.data_ptr() + 8 * i * g.input_size), steps, step, nblocks)
    p = [q for buf, n in raw for q in fe.pdus_to_dicts(buf, n)]
    fe.close()
    return p, steps * g.input_size / el / 1e6
for steps, warmup in ((256, 8), (20, 5)):
    ref, v = full_pdus(steps, warmup)
    print("full fold, %d steps: %.0f Msamples/s, %d PDUs" % (steps, v, len(ref)))
    for tol in (3e-7, 1e-6):
        bench.PRUNE_TOL = tol
        r = bench.pruned_fold_leg(torch, hf, F, w, freqs, x, 0, steps, warmup, ref)
        s = r["streams"]
        print(tol, "steps", steps, "rows", r["fold_rows"], "value %.0f" % r["value"], "fold %.3f ms" % r["fold_kernel_avg_ms"], "demod %.3f" % r["demod_kernel_ms_per_block"],
              "A %.2f B %.2f D %.2f" % (s["stream_a_ms"], s["stream_b_ms"], s["stream_d_ms"]), "err %.2e" % r["chan_out_rel_rms_vs_full_fold"], "pdus", r["pdus"],
              "same", r["pdus_same_as_full_fold"], r["detection_sample_max_abs_diff"], flush=True)
PY
cat gpurun_out/r5n/legs.txt | tail -20
