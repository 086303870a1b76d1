#!/usr/bin/env python3
"""Times every compiled register tiling of the fold kernel (dumphfdl_amd/csrc/fold_kernels.hip, fold_variants[]) on one workload's
resident filter taps and prints a markdown table: tiling, blocks per launch, ms per launch, ms per block, algorithmic GB/s, fraction
of the 8 TB/s peak, and whether the partial sums are bit-identical to the first tiling of the same block count.

    HFDL_GPU_FOLD_BATCH=8 python profiles/fold_variants.py [cfg3|cfg2] [reps]
"""
import json
import os
import sys

os.environ.setdefault("HFDL_GPU_FOLD_BATCH", "8")       # a half of 8 spectra so that the 8-block tilings can run
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
import dumphfdl_amd as hf   # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = bench.WORKLOADS[wl]
fe = hf.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w))
g = fe.geometry
fe.enable_taps(False)
rng = np.random.default_rng(1)
x = (rng.standard_normal(2 * 8 * g.input_size).astype(np.float32) * 0.05)
dev = torch.from_numpy(x).cuda()
for b in range(8):      # eight spectra into the half (no sync in between: they stay queued until the half is full)
    fe.push_block(dev.data_ptr() + 8 * b * g.input_size)
fe.poll_pdus()
rows, ref = [], {}
for v, (u, r, cs, nc, nb, wv) in enumerate(F.fold_variants()):
    if wv >= 2:                     # LDS-staged spectra, wv - 2 waves per workgroup: any M that is a multiple of 128 U
        if g.fft_inv_size % (128 * u) or g.channels < (wv - 2) * nc:
            continue
    elif g.fft_inv_size != (128 if wv else 512) * u * cs:
        continue
    try:
        avg, best, chk = fe.fold_variant_probe(v, reps)
    except hf.GpuError as e:
        print("variant %d skipped: %s" % (v, e), file=sys.stderr)
        continue
    ref.setdefault(nb, None)
    byt = bench.alg_bytes_per_launch(g, nb)
    rows.append(dict(variant=v, WV=wv, U=u, R=r, CS=cs, NC=nc, NB=nb, avg_ms=avg, best_ms=best, ms_per_block=avg / nb, GBs=byt / (avg * 1e-3) / 1e9,
                     frac=byt / (avg * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, checksum=chk))
# bit identity: block 0 .. NB-1 of every tiling with the same NB must give the same partial sums
first = {}
for r_ in rows:
    first.setdefault(r_["NB"], r_["checksum"])
    r_["bit_identical_to_first_of_NB"] = r_["checksum"] == first[r_["NB"]]
print("# fold kernel tilings on %s (M = %d, %d channels, %d slices x %d alias rows; %d launches each after one untimed)" %
      (wl, g.fft_inv_size, g.channels, g.fold_slices, g.pre_decimation // g.fold_slices, reps))
print()
print("| W | U | R | CS | NC | NB | ms / launch (avg) | best | ms / block | algorithmic GB/s | of 8 TB/s | same bits as first NB tiling |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r_ in rows:
    print("| %d | %d | %d | %d | %d | %d | %.3f | %.3f | %.3f | %.0f | %.3f | %s |" % (r_["WV"], r_["U"], r_["R"], r_["CS"], r_["NC"], r_["NB"], r_["avg_ms"], r_["best_ms"],
                                                                             r_["ms_per_block"], r_["GBs"], r_["frac"], "yes" if r_["bit_identical_to_first_of_NB"] else "NO"))
print()
print("```json")
print(json.dumps(rows))
print("```")
fe.close()
