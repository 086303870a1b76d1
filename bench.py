#!/usr/bin/env python3
"""bench.py -- wideband I/Q Msamples/s (+ HFDL frames/s) of the MI355X HFDL front end at a fixed channel count.

  python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4]

One "step" = one block of `input_size` wideband cf32 samples through the WHOLE hot path: overlap assembly, forward
FFT, per-channel fold + inverse FFT + NCO (fastddc), per-channel demodulator, burst decoder, PDU read-back.
The synthetic input is resident in HBM before the timed region.  For N > 1 the driver launches one rank per GPU;
every rank owns an independent wideband stream with the same channel count (BASELINE.json config 5, channels /
streams sharded, no data-path collective); `value` is the whole-job aggregate.

The JSON line carries, next to the driver's contract fields:
  roofline      the fold kernel (spectrum x per-channel filter, >99% of the block's algorithmic bytes), timed with
                HIP events on the front end's own stream; bytes = SURVEY.md section 8(d) canonical model
  cpu_baseline  the plain-C oracle (a restatement -- FFTW / liquid-dsp are not installable here) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]: 40 Msps synthetic cf32, 256 channels on a 150 kHz grid, 1 MI355X
    "cfg3": dict(fs=40_000_000, centerfreq=15_000_000, nch=256, grid=150_000, blocks=16, seed=3, noise=0.05,
                 name="40 Msps cf32, 256 HFDL channels (BASELINE.json configs[2]; per rank at N>1 = configs[4])"),
    # BASELINE.json configs[3]: same geometry, every channel carries back-to-back bursts cycling all 8 modes (Viterbi batch stress)
    "cfg4": dict(fs=40_000_000, centerfreq=15_000_000, nch=256, grid=150_000, blocks=32, seed=4, noise=0.05, dense=True,
                 name="40 Msps cf32, 256 HFDL channels, burst-dense: back-to-back 300/600/1200/1800 bps single+double-slot bursts (BASELINE.json configs[3])"),
    # BASELINE.json configs[1]: 8 Msps, 32 channels on a 200 kHz grid
    "cfg2": dict(fs=8_000_000, centerfreq=10_000_000, nch=32, grid=200_000, blocks=26, seed=2, noise=0.02,
                 name="8 Msps cf32, 32 HFDL channels (BASELINE.json configs[1])"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def channel_plan(w):
    nch, grid, cf = w["nch"], w["grid"], w["centerfreq"]
    return [int(cf + (i - nch // 2) * grid + grid // 2 - 1440) for i in range(nch)]


def make_input(w, geom_input_size, rank, world):
    """Seeded synthetic wideband stream: one single-slot burst per channel (modes cycle 300/600/1200/1800 bps) + AWGN."""
    from dumphfdl_amd import synth
    from dumphfdl_amd import shard
    seed = shard.stream_seed(w["seed"], rank, world)
    nsamp = w["blocks"] * geom_input_size
    cache = "/tmp/hfdl_bench_%s_seed%d_%d%s.npy" % (w["fs"], seed, nsamp, "_dense" if w.get("dense") else "")
    freqs = channel_plan(w)
    dur = nsamp / w["fs"]
    bursts = plan_bursts(w, freqs, dur, seed)
    if os.path.exists(cache):
        x = np.load(cache, mmap_mode="r")
        if x.shape == (nsamp,):
            return np.ascontiguousarray(x), bursts
    x = synth.synth_wideband(w["fs"], w["centerfreq"], nsamp, bursts, noise_sigma=w["noise"], seed=seed)
    try:
        np.save(cache, x)
    except OSError:
        pass
    return x, bursts


def plan_bursts(w, freqs, dur, seed):
    """The traffic of a workload: one single-slot burst per channel, or (dense) as many back-to-back bursts as fit."""
    from dumphfdl_amd import synth
    rng = np.random.default_rng(seed)
    bursts = []
    for i, f in enumerate(freqs):
        if w.get("dense"):
            # as many bursts as fit the resident stretch, modes cycling 0..7 from a per-channel offset
            t, k = float(rng.uniform(0.02, 0.2)), 0
            while True:
                mode = (i + k) % 8
                length = synth.burst_symbols_len(mode) / 1800
                if t + length > dur - 0.05:
                    if k == 0 and mode >= 4:            # a double-slot burst does not fit any more: try the single-slot one
                        k += 4
                        continue
                    break
                bursts.append(dict(freq=f, mode=mode, octets=synth.make_pdu(rng, mode), t0=t,
                                   amp=float(rng.uniform(0.01, 0.03)), cfo=float(rng.uniform(-15, 15))))
                t += length + float(rng.uniform(0.05, 0.15))
                k += 1
            continue
        mode = i % 4
        t0 = float(rng.uniform(0.02, max(0.03, dur - synth.burst_symbols_len(mode) / 1800 - 0.05)))
        bursts.append(dict(freq=f, mode=mode, octets=synth.make_pdu(rng, mode), t0=t0,
                           amp=float(rng.uniform(0.01, 0.03)),        # ~19..29 dB in-channel SNR
                           cfo=float(rng.uniform(-15, 15))))
    return bursts


def cpu_baseline(w, x, input_size, target_seconds=20.0):
    """Time the oracle (plain-C restatement of the reference path, one worker thread per channel like the reference)
    on this host: C_s channels of the same geometry, a few blocks; scale the per-channel part to the full channel count."""
    from oracle import pyoracle
    cores = max(1, min(os.cpu_count() or 1, 64))
    freqs = channel_plan(w)
    cs = min(len(freqs), cores)
    sel = freqs[:: max(1, len(freqs) // cs)][:cs]
    t0 = time.time()
    fe = pyoracle.Frontend(w["fs"], w["centerfreq"], sel, nthreads=cores)
    t_init = time.time() - t0
    # one untimed block (page-in, twiddle tables), then timed blocks
    fe.push_block(x[:input_size], nthreads=cores)
    nblk, t_all, t_fft = 0, 0.0, 0.0
    L = pyoracle.lib()
    import ctypes as C
    spec = np.empty(fe.ddc.fft_size, np.complex64)
    buf = np.zeros(fe.ddc.fft_size, np.complex64)
    while nblk < 2 or (t_all < target_seconds / 2 and nblk < w["blocks"] - 1):
        blk = np.ascontiguousarray(x[(nblk + 1) * input_size:(nblk + 2) * input_size])
        t1 = time.time()
        fe.push_block(blk, nthreads=cores)
        t_all += time.time() - t1
        t1 = time.time()            # the shared forward FFT alone (1 thread, like --fft-threads 1)
        L.orc_forward_block(buf.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), C.byref(fe.ddc), spec.ctypes.data_as(C.c_void_p))
        t_fft += time.time() - t1
        nblk += 1
    per_blk, fft_blk = t_all / nblk, t_fft / nblk
    chan_blk = max(per_blk - fft_blk, 1e-9)          # cs channels on `cores` threads
    full = fft_blk + chan_blk * (len(freqs) / cs)
    frames = len(fe.pdus)
    fe.close()
    return dict(value=input_size / full / 1e6, unit="Msamples/s", cores=cores, kind="port",
                sample="oracle (C restatement; FFTW/liquid-dsp binaries unavailable): %d of %d channels x %d blocks of %d samples on %d "
                       "threads, %.2f s/block measured (forward FFT %.2f s), channel part scaled x%.1f to %d channels; init %.1f s untimed; %d PDUs"
                       % (cs, len(freqs), nblk, input_size, cores, per_blk, fft_blk, len(freqs) / cs, len(freqs), t_init, frames))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="timed blocks (one step = one block of input_size samples through the whole path); 256 blocks of the 40 Msps geometry = 0.75 s, so pipeline fill and drain stay below 1 %")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-input", action="store_true",
                    help="feed the blocks from page-locked HOST memory (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--sample-format", default="cf32", choices=["cf32", "cs16"], help="with --host-input: raw format pushed over PCIe")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / final reduction at N > 1 (nccl = RCCL; gloo for a 1-GPU smoke of the N>1 path)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    import torch            # plumbing only: device memory for the resident input, barrier / max-reduce across ranks
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HFDL front end has no CPU path")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    red_device = "cuda"
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
            red_device = "cpu"
    import dumphfdl_amd as hf

    freqs = channel_plan(w)
    t0 = time.time()
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=dev_index)
    g = fe.geometry
    fe.enable_taps(False)            # per-stage debug taps (DATADUMPS analogue) are a test facility, not part of the path
    t_create = time.time() - t0
    t0 = time.time()
    x, bursts = make_input(w, g.input_size, rank, world)
    t_gen = time.time() - t0
    nblocks = len(x) // g.input_size
    if args.host_input:
        import ctypes
        from dumphfdl_amd import frontend as F
        if args.sample_format == "cs16":
            raw = np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16)
            fmt, bps = F.SFMT_CS16, 4
        else:
            raw, fmt, bps = x.view(np.float32), F.SFMT_CF32, 8
        hbuf = hf.host_alloc(raw.nbytes)
        ctypes.memmove(hbuf, raw.ctypes.data, raw.nbytes)
        hptrs = [hbuf + bps * b * g.input_size for b in range(nblocks)]
        push = lambda i: fe.push_host_ptr(hptrs[i], fmt)
        dev = None
    else:
        dev = torch.from_numpy(x.view(np.float32)).cuda()          # resident in HBM before the timed region
        ptrs = [dev.data_ptr() + 8 * b * g.input_size for b in range(nblocks)]
        push = lambda i: fe.push_block(ptrs[i])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        push(step % nblocks); step += 1
    fe.poll_pdus()
    fe.reset_timers(True)
    barrier()
    t0 = time.perf_counter()
    raw = []
    for i in range(args.steps):
        push(step % nblocks); step += 1
        if i % 256 == 255 and i + 1 < args.steps:      # long runs: empty the device PDU ring now and then, pipeline kept running
            raw.append(fe.poll_pdus_raw(16384, max_in_flight=1))
    raw.append(fe.poll_pdus_raw(16384))     # sync + device->host of every PDU struct produced by the timed blocks (what the C host gets)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pdus = [p for buf, n in raw for p in fe.pdus_to_dicts(buf, n)]     # Python-side unpacking for the checks below: not part of the path
    npdus = len(pdus)
    fold_ms, fold_n = fe.fold_time_ms()
    stream_gbs = fe.stream_read_probe() if rank == 0 else None       # after the timed region: the board's own read ceiling
    barrier()
    from dumphfdl_amd import shard
    elapsed_max, total_samples, total_pdus = shard.reduce_job(elapsed, args.steps * g.input_size, npdus, dist, device=red_device)

    if rank == 0:
        good = sum(1 for p in pdus if any(p["octets"][:len(b["octets"])] == b["octets"] for b in bursts if b["freq"] == p["freq"]))
        samples = total_samples
        # B = 8*input_size + C*8*N + C*8*(post_input_size/post_decimation) algorithmic bytes per block
        alg_bytes = 8 * g.input_size + g.channels * 8 * g.fft_size + g.channels * 8 * (g.post_input_size // g.post_decimation)
        fold_avg_ms = fold_ms / max(fold_n, 1)
        achieved = alg_bytes / (fold_avg_ms * 1e-3) / 1e9 if fold_n else None
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "fold_traffic_%s.json" % args.workload)
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "wideband I/Q Msamples/s (cf32 ingest -> decoded HFDL PDUs) at fixed channel count",
            "value": samples / elapsed_max / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.host_input else "synthetic, fed from page-locked host memory as %s (PCIe-inclusive)" % args.sample_format,
            "config": {"workload": w["name"], "sample_rate": w["fs"], "channels": g.channels, "fft_size": g.fft_size,
                       "fft_inv_size": g.fft_inv_size, "block_samples": g.input_size, "resident_blocks": nblocks,
                       "parallelism": "1 independent %d-channel stream per GPU, no collectives" % g.channels},
            "frames_per_s": total_pdus / elapsed_max, "pdus_in_timed_region": total_pdus,
            "pdus_rank0_matching_sent_payload": good,
            "pdus_rank0_fcs_good_on_device": sum(1 for p in pdus if p["fcs_status"] == 0),
            "roofline": {"bound": "hbm", "kernel": "fold_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fold_avg_ms, "launches": fold_n,
                         "literal_bytes_per_launch": 16 * g.fft_size * (g.channels + 1),      # SURVEY 8(d) secondary figure
                         "stream_read_GBs": stream_gbs,
                         "frac_of_stream_read": (achieved / stream_gbs) if (achieved and stream_gbs) else None},
            "setup_s": {"frontend_create": round(t_create, 2), "input_synthesis": round(t_gen, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            fe.close()
            del dev
            out["cpu_baseline"] = cpu_baseline(w, x, g.input_size)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
