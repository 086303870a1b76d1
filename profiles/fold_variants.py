#!/usr/bin/env python3
"""Times every compiled tiling of the matrix-pipe fold kernel (dumphfdl_amd/csrc/fold_kernels.hip, fold_variants[] of the LABORATORY
build, libhfdl_gpu_lab.so) on one workload's resident filter taps and prints a markdown table: tiling, blocks per launch, ms per
launch, ms per block, algorithmic GB/s, fraction of the 8 TB/s peak, and whether the partial sums are bit-identical to the plain-VALU
FMA-chain reference kernel at the same block count.

    python profiles/fold_variants.py [cfg3|cfg2] [reps] [nb,nb,...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
import dumphfdl_amd as hf   # noqa: E402
from dumphfdl_amd import frontend as F   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nbs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 8, 16]
w = bench.WORKLOADS[wl]
fe = F.Frontend(w["fs"], w["centerfreq"], bench.channel_plan(w), lib=F.load_lab())
g = fe.geometry
fe.enable_taps(False)
rng = np.random.default_rng(1)
x = (rng.standard_normal(2 * 16 * g.input_size).astype(np.float32) * 0.05)
dev = torch.from_numpy(x).cuda()
# sixteen spectra into the half (no sync in between: they stay queued until the half is full); thirty-two when the front end was created
# with HFDL_GPU_FOLD_BATCH=32 and the thirty-two-column tilings are asked for (the sixteen resident blocks pushed twice)
for b in range(max(16, min(max(nbs), g.fold_batch))):
    fe.push_block(dev.data_ptr() + 8 * (b % 16) * g.input_size)
fe.poll_pdus()
rows = []
ref = {}
for nb in nbs:
    try:
        ref[nb] = fe.fold_variant_probe(-1, nb, 1)
    except hf.GpuError as e:
        print("reference kernel at %d blocks: %s" % (nb, e), file=sys.stderr)
only = [int(v) for v in os.environ["FOLD_VARIANTS"].split(",")] if os.environ.get("FOLD_VARIANTS") else None      # PMC passes: a few tilings only
for v, (p, q, wv, d, nbmax, layout) in enumerate(F.fold_variants()):
    if only is not None and v not in only:
        continue
    for nb in nbs:
        if nb > nbmax or (nbmax == 32 and nb <= 16 and nb != 16):        # the four-column form takes at most 4 blocks, the sixteen-column form any count up to 16, the thirty-two-column form is timed at 16 (for comparison) and beyond
            continue
        try:
            avg, best, chk = fe.fold_variant_probe(v, nb, reps)
        except hf.GpuError as e:
            print("variant %d skipped: %s" % (v, e), file=sys.stderr)
            continue
        byt = bench.alg_bytes_per_launch(g, nb)
        rows.append(dict(variant=v, family="16x16x4", P=p, Q=q, W=wv, D=d, NB=nb, avg_ms=avg, best_ms=best, ms_per_block=avg / nb, GBs=byt / (avg * 1e-3) / 1e9,
                         frac=byt / (avg * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, checksum=chk,
                         bit_identical_to_fma_reference=(chk == ref[nb][2]) if nb in ref else None))
print("# matrix-pipe fold tilings on %s (M = %d, %d channels, %d slices x %d alias rows; %d launches each after one untimed)" %
      (wl, g.fft_inv_size, g.channels, g.fold_slices, g.pre_decimation // g.fold_slices, reps))
print()
print("P = channel octets per wave, W = waves per workgroup, D = quads of alias rows of loads in flight (Q = 4: up to 16 blocks).  FMA-chain reference kernel: "
      + ", ".join("%d blocks %.1f ms" % (nb, r[0]) for nb, r in sorted(ref.items())))
print()
print("| MFMA | P | Q | W | D | NB | ms / launch (avg) | best | ms / block | algorithmic GB/s | of 8 TB/s | same bits as the FMA-chain reference |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r_ in rows:
    print("| %s | %d | %d | %d | %d | %d | %.3f | %.3f | %.3f | %.0f | %.3f | %s |" % (r_["family"], r_["P"], r_["Q"], r_["W"], r_["D"], r_["NB"], r_["avg_ms"], r_["best_ms"],
                                                                        r_["ms_per_block"], r_["GBs"], r_["frac"], {True: "yes", False: "NO", None: "?"}[r_["bit_identical_to_fma_reference"]]))
print()
print("```json")
print(json.dumps(rows))
print("```")
fe.close()
