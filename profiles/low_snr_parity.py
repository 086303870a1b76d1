#!/usr/bin/env python3
"""GPU vs oracle on traffic binned by in-channel SNR: where does "the decoded-frame set is identical" hold?

64 channels at 1 Msps; per 2 dB bin every channel carries four bursts (all eight modes over the channels), each at the bin's SNR
+- 1 dB.  The same samples go through the GPU front end and through the strict oracle; per bin: PDUs either side, the
(freq, sample_index, mode, octets) multisets' intersection, GPU-only / oracle-only PDUs, sent payloads recovered by either.

In-channel SNR = burst power over the noise power of the whole post-channelizer band (fs / 128 = 7812.5 Hz).

  python profiles/low_snr_parity.py [--bins -8:10:2] [--bursts-per-channel 4] > gpurun_out/low_snr.json

sweep() is also what tests/test_gpu_low_snr.py asserts on and what bench.py's `parity` block embeds (fewer bins).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hfdl_synth as synth          # noqa: E402

FS, CF, NCH, DECIM = 1_000_000, 10_000_000, 64, 128
FREQS = [int(CF + (i - NCH // 2) * 14_000 + 3_000) for i in range(NCH)]
NOISE_SIGMA = 0.02
IN_CHANNEL_NOISE_RMS = NOISE_SIGMA * np.sqrt(2.0) / np.sqrt(DECIM)


def plan_bin(snr_db, bursts_per_channel, seed):
    rng = np.random.default_rng(seed)
    bursts = []
    for i, f in enumerate(FREQS):
        t = float(rng.uniform(0.3, 0.9))
        for k in range(bursts_per_channel):
            mode = (i + k * 3) % 8                                    # all eight modes, spread over channels and positions
            amp = float(10 ** ((snr_db + rng.uniform(-1, 1)) / 20) * IN_CHANNEL_NOISE_RMS)
            bursts.append(dict(freq=f, mode=mode, octets=synth.make_pdu(rng, mode), t0=t, amp=amp, cfo=float(rng.uniform(-25, 25))))
            t += synth.burst_symbols_len(mode) / 1800 + float(rng.uniform(0.3, 0.5))
    dur = max(b["t0"] + synth.burst_symbols_len(b["mode"]) / 1800 for b in bursts) + 0.4
    return bursts, dur


def synth_bin(args):
    snr_db, bursts_per_channel, seed = args
    bursts, dur = plan_bin(snr_db, bursts_per_channel, seed)
    return bursts, synth.synth_wideband(FS, CF, int(dur * FS), bursts, noise_sigma=NOISE_SIGMA, seed=seed)


def bin_seed(snr_db):
    return 1000 + int(round(snr_db * 10))


def run_bin(hf, pyoracle, snr_db, bursts_per_channel=4, seed=None, threads=None, device=0, made=None):
    seed = bin_seed(snr_db) if seed is None else seed
    threads = threads or max(1, min(os.cpu_count() or 1, 64))
    bursts, x = made if made is not None else synth_bin((snr_db, bursts_per_channel, seed))
    fe = hf.Frontend(FS, CF, FREQS, device=device)
    fe.enable_taps(False)
    ora = pyoracle.Frontend(FS, CF, FREQS, nthreads=threads)
    n = fe.input_size
    gpu_pdus = []
    for b in range(len(x) // n):
        fe.push_block(x[b * n:(b + 1) * n])
        ora.push_block(x[b * n:(b + 1) * n], nthreads=threads)
        if b % 8 == 7:
            gpu_pdus += fe.poll_pdus(max_in_flight=1)
    gpu_pdus += fe.poll_pdus()
    key = lambda p: (p["freq"], p["sample_index"], p["mode"], p["octets"])
    got, want = sorted(map(key, gpu_pdus)), sorted(map(key, ora.pdus))
    gs, ws = set(got), set(want)
    sent = {}
    for b in bursts:
        sent.setdefault(b["freq"], []).append(b)
    good = lambda s: {(f, si, m, o) for f, si, m, o in s if any(o[:len(b["octets"])] == b["octets"] and m == b["mode"] for b in sent[f])}
    ok = lambda s: len(good(s))
    # same frame found at the same place with other octets, or the same octets found elsewhere (+-3 samples)?
    changed = sum(1 for f, si, m, o in gs - ws if any(f == f2 and m == m2 and abs(si - s2) <= 3 and o != o2 for f2, s2, m2, o2 in ws - gs))
    moved = sum(1 for f, si, m, o in gs - ws if any(f == f2 and m == m2 and 0 < abs(si - s2) <= 3 and o == o2 for f2, s2, m2, o2 in ws - gs))
    fe.close()
    ora.close()
    return dict(snr_db=snr_db, bursts=len(bursts), gpu_pdus=len(got), oracle_pdus=len(want), common=len(gs & ws), gpu_only=len(gs - ws),
                oracle_only=len(ws - gs), identical=got == want, same_place_other_octets=changed, same_octets_other_place=moved,
                gpu_recovered=ok(gs), oracle_recovered=ok(ws), recovered_sets_identical=good(gs) == good(ws), samples=len(x))


def sweep(hf, pyoracle, bins, bursts_per_channel=4, device=0):
    """The bins' streams are synthesised side by side on the host's cores (numpy, ~10 s each), then decoded one after the other."""
    from multiprocessing import get_context
    jobs = max(1, min(len(bins), (os.cpu_count() or 1) // 2))
    if jobs > 1:
        with get_context("spawn").Pool(jobs) as pool:
            made = pool.map(synth_bin, [(s, bursts_per_channel, bin_seed(s)) for s in bins])
    else:
        made = [None] * len(bins)
    return [run_bin(hf, pyoracle, s, bursts_per_channel, device=device, made=m) for s, m in zip(bins, made)]


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", default="-8:10:2")
    ap.add_argument("--bursts-per-channel", type=int, default=4)
    a = ap.parse_args()
    lo, hi, st = (int(v) for v in a.bins.split(":"))
    import dumphfdl_amd as hf
    from oracle import pyoracle
    rows = sweep(hf, pyoracle, list(range(lo, hi + 1, st)), a.bursts_per_channel)
    print(json.dumps(dict(lib=os.environ.get("HFDL_GPU_LIB", "libhfdl_gpu.so"), fs=FS, channels=NCH, rows=rows)))
