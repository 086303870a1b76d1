#!/usr/bin/env python3
"""bench.py -- wideband I/Q Msamples/s (+ HFDL frames/s) of the MI355X HFDL front end at a fixed channel count.

  python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4] [--shard streams|channels]

One "step" = one block of `input_size` wideband cf32 samples through the WHOLE hot path: overlap assembly, forward
FFT, per-channel fold + inverse FFT + NCO (fastddc), per-channel demodulator, burst decoder, PDU read-back.
`value` is measured with the synthetic input resident in HBM before the timed region.  For N > 1 the driver launches one
rank per GPU; with --shard streams (default) every rank owns an independent wideband stream with the same channel count
(BASELINE.json configs[4], weak scaling); with --shard channels all ranks ingest the SAME stream and each decodes a
round-robin subset of its channels (SURVEY.md 8e, strong scaling).  No data-path collective either way.

The JSON line carries, next to the driver's contract fields:
  roofline        the fold kernel (spectrum x per-channel filter, >99% of the block's algorithmic bytes), timed with
                  HIP events on the front end's own stream; bytes = SURVEY.md section 8(d) canonical model
  host_ram_input  the same workload fed from page-locked HOST memory (SURVEY.md 8(d) "input pre-loaded in host RAM"):
                  PCIe-inclusive, never `value`
  host_path       the same workload through the C host library (file input -> block graph -> GPU front end ->
                  pdu_decoder_queue_push), i.e. what a dumphfdl user gets (dumphfdl_amd/hfdl_replay --bench)
  pdus_*          every PDU of the timed region is checked against the sent traffic: payload + mode, on-device header FCS,
                  and the device's LPDU walk (every announced LPDU found with a good FCS; the traffic carries real LPDU lists)
  fec             trellis steps/s: demanded by the run, and the burst decoder's own capacity on a resident batch
  parity          same-run gate: the channels the CPU baseline decodes, decoded by the GPU from the same blocks --
                  channelizer error RMS / signal RMS and the (freq, sample_index, mode, octets) multisets
  cpu_baseline    the plain-C oracle built with the reference's release flags, timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[0] as SURVEY.md 8(d) words it: 250 ksps cf32, ONE channel at centre + 37 kHz, 60 s (523 blocks of 28672 samples), one single-slot 300 bps
    # SPDU on the 32 s frame grid, AWGN at 15 dB Es/N0, seed 1.  (Es/N0 = in-channel SNR + 10 log10(7812.5 Hz / 1800 Bd) = SNR + 6.4 dB:
    # amplitude 0.0135 over sigma 0.02 per component.)  The reference's CPU-runnable plumbing case; here it exercises the
    # single-channel shape of the fold (one channel padded to a pair, the single-wave workgroups) through the same kernels.
    "cfg1": dict(fs=250_000, centerfreq=10_000_000, nch=1, freqs=[10_037_000], blocks=523, seed=1, noise=0.02, slot_grid_s=32.0, es_n0_db=15.0,
                 name="250 ksps cf32 --iq-file, 1 HFDL channel at centre + 37 kHz, 60 s, one 300 bps SPDU per 32 s frame, 15 dB Es/N0 (BASELINE.json configs[0])"),
    # BASELINE.json configs[2]: 40 Msps synthetic cf32, 256 channels on a 150 kHz grid, 1 MI355X
    "cfg3": dict(fs=40_000_000, centerfreq=15_000_000, nch=256, grid=150_000, blocks=16, seed=3, noise=0.05,
                 name="40 Msps cf32, 256 HFDL channels (BASELINE.json configs[2]; per rank at N>1 = configs[4])"),
    # BASELINE.json configs[3]: same geometry, every channel carries back-to-back bursts cycling all 8 modes (Viterbi batch stress)
    "cfg4": dict(fs=40_000_000, centerfreq=15_000_000, nch=256, grid=150_000, blocks=32, seed=4, noise=0.05, dense=True,
                 name="40 Msps cf32, 256 HFDL channels, burst-dense: back-to-back 300/600/1200/1800 bps single+double-slot bursts (BASELINE.json configs[3])"),
    # BASELINE.json configs[1]: 8 Msps, 32 channels on a 200 kHz grid.  SURVEY.md 8(d) words it with 30 s of signal: the bench keeps 26 blocks
    # (3 s) resident in HBM and replays them (it measures a rate); the thirty seconds themselves -- burst-dense, all eight modes, a cs16 file
    # through hfdl_replay against the oracle on every channel -- are tests/test_gpu_parity.py::test_cfg2_thirty_seconds_through_the_c_host_program
    "cfg2": dict(fs=8_000_000, centerfreq=10_000_000, nch=32, grid=200_000, blocks=26, seed=2, noise=0.02,
                 name="8 Msps cf32, 32 HFDL channels (BASELINE.json configs[1])"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (v_mfma_f32_*_f32: f32 in, f32 accumulate), same guide
NBITS = [540, 1080, 2160, 3240, 1260, 2520, 5040, 7560]      # decoded bits = trellis steps per frame, mode 0..7 (src/hfdl.c:81-138)


def channel_plan(w):
    if "freqs" in w:
        return list(w["freqs"])
    nch, grid, cf = w["nch"], w["grid"], w["centerfreq"]
    return [int(cf + (i - nch // 2) * grid + grid // 2 - 1440) for i in range(nch)]


def make_input(w, geom_input_size, rank, world, shard_mode="streams"):
    """Seeded synthetic wideband stream: one single-slot burst per channel (modes cycle 300/600/1200/1800 bps) + AWGN."""
    import hfdl_synth as synth
    from dumphfdl_amd import shard
    seed = shard.stream_seed(w["seed"], rank, world) if shard_mode == "streams" else w["seed"]
    nsamp = w["blocks"] * geom_input_size
    freqs = channel_plan(w)
    dur = nsamp / w["fs"]
    bursts = plan_bursts(w, freqs, dur, seed)
    # the cached stream belongs to THIS traffic plan: a file left by another version of the plan (other payloads) is not reused
    import zlib
    tag = zlib.crc32(b"".join(b["octets"] + bytes([b["mode"]]) + np.float64([b["t0"], b["amp"], b["cfo"]]).tobytes() for b in bursts))
    cache = "/tmp/hfdl_bench_%s_seed%d_%d%s_%08x.npy" % (w["fs"], seed, nsamp, "_dense" if w.get("dense") else "", tag)
    def cached():
        if os.path.exists(cache):
            x = np.load(cache, mmap_mode="r")
            if x.shape == (nsamp,):
                return np.ascontiguousarray(x)
        return None

    x = cached()
    if x is not None:
        return x, bursts
    # one synthesis per seed on a node: ranks that share a stream (--shard channels) wait for the first one's file instead of
    # each running the same ~8 s of numpy beside the others
    import fcntl
    lock = None
    try:
        lock = open(cache + ".lock", "w")
        fcntl.flock(lock, fcntl.LOCK_EX)
        x = cached()
    except OSError:
        lock = None
    try:
        if x is None:
            x = synth.synth_wideband(w["fs"], w["centerfreq"], nsamp, bursts, noise_sigma=w["noise"], seed=seed)
            try:
                tmp = cache + ".%d.tmp.npy" % os.getpid()
                np.save(tmp, x)
                os.replace(tmp, cache)
            except OSError:
                pass
    finally:
        if lock is not None:
            fcntl.flock(lock, fcntl.LOCK_UN)
            lock.close()
    return x, bursts


def make_payload(rng, mode):
    """What a burst carries: an SPDU (half of the 66-octet frames) or an MPDU with a REAL LPDU list -- down- or uplink, every
    LPDU ending in its own FCS -- so that the device's LPDU walk (hfdl_gpu_pdu.lpdus_*) has something to count and the run can
    check it.  Returns (octets, LPDUs sent or None for an SPDU)."""
    import hfdl_synth as synth
    n = synth.mode_sizes(mode)["max_payload"]
    if n == 66 and rng.random() < 0.5:
        return synth.make_spdu(rng), None
    octets, cnt = synth.make_mpdu_with_lpdus(rng, n, uplink=bool(rng.random() < 0.25))
    return octets, cnt


def plan_bursts(w, freqs, dur, seed):
    """The traffic of a workload: one single-slot burst per channel, or (dense) as many back-to-back bursts as fit."""
    import hfdl_synth as synth
    rng = np.random.default_rng(seed)
    bursts = []

    def burst(f, mode, t0):
        octets, lpdus = make_payload(rng, mode)
        return dict(freq=f, mode=mode, octets=octets, lpdus=lpdus, t0=t0,
                    amp=float(rng.uniform(0.01, 0.03)),        # ~19..29 dB in-channel SNR
                    cfo=float(rng.uniform(-15, 15)))

    if "slot_grid_s" in w:
        # one single-slot 300 bps SPDU per frame of 32 s (13 slots of 2.461 s, ARINC 635), the first one second into the stretch
        noise_rms = w["noise"] * np.sqrt(2.0) / np.sqrt(2 ** int(np.floor(np.log2(w["fs"] / 5400.0))))        # in the channel's fs / decimation band
        amp = float(noise_rms * 10 ** ((w["es_n0_db"] - 10 * np.log10((w["fs"] / 2 ** int(np.floor(np.log2(w["fs"] / 5400.0)))) / 1800.0)) / 20))
        t = 1.0
        while t + synth.burst_symbols_len(0) / 1800 < dur - 0.05:
            for f in freqs:
                bursts.append(dict(freq=f, mode=0, octets=synth.make_spdu(rng), lpdus=None, t0=t, amp=amp, cfo=float(rng.uniform(-15, 15))))
            t += w["slot_grid_s"]
        return bursts
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
                bursts.append(burst(f, mode, t))
                t += length + float(rng.uniform(0.05, 0.15))
                k += 1
            continue
        mode = i % 4
        t0 = float(rng.uniform(0.02, max(0.03, dur - synth.burst_symbols_len(mode) / 1800 - 0.05)))
        bursts.append(burst(f, mode, t0))
    return bursts


def pdu_key(p):
    return (p["freq"], p["sample_index"], p["mode"], p["octets"].hex())


def matches_sent(p, bursts_by_freq):
    return any(p["octets"][:len(b["octets"])] == b["octets"] and p["mode"] == b["mode"] for b in bursts_by_freq.get(p["freq"], ()))


def lpdu_walk_matches_sent(p, bursts_by_freq):
    """The device's LPDU walk of this PDU = what was put on the air: every announced LPDU found with a good FCS (none for an SPDU)."""
    for b in bursts_by_freq.get(p["freq"], ()):
        if p["octets"][:len(b["octets"])] == b["octets"] and p["mode"] == b["mode"]:
            want = (0, 0, 0, 0, 0) if b.get("lpdus") is None else (b["lpdus"], b["lpdus"], 0, 0, 0)
            return tuple(p["lpdus"]) == want
    return False


def probe_cpu_libs():
    """say whether the reference's CPU libraries exist on this box (src/fft_fftw.c:22-41 binds FFTW3f)."""
    found = {}
    for key, names in (("fftw3f", ("libfftw3f.so.3", "libfftw3f.so")), ("liquid", ("libliquid.so", "libliquid.so.1"))):
        found[key] = None
        for n in names:
            try:
                ctypes.CDLL(n)
                found[key] = n
                break
            except OSError:
                continue
    return found


def fftw_forward_seconds(n, name, reps=3):
    """csdr_make_fft_c2c + csdr_fft_execute as the reference drives FFTW (src/fft_fftw.c:22-41): FFTW_ESTIMATE plan, 1 thread."""
    L = ctypes.CDLL(name)
    L.fftwf_plan_dft_1d.restype = ctypes.c_void_p
    L.fftwf_plan_dft_1d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
    L.fftwf_execute.argtypes = [ctypes.c_void_p]
    L.fftwf_destroy_plan.argtypes = [ctypes.c_void_p]
    a = (np.random.default_rng(0).standard_normal(2 * n).astype(np.float32)).view(np.complex64)
    b = np.empty_like(a)
    plan = L.fftwf_plan_dft_1d(n, a.ctypes.data, b.ctypes.data, -1, 1 << 6)      # FFTW_FORWARD, FFTW_ESTIMATE
    L.fftwf_execute(plan)
    t0 = time.time()
    for _ in range(reps):
        L.fftwf_execute(plan)
    dt = (time.time() - t0) / reps
    L.fftwf_destroy_plan(plan)
    return dt


def parity_gate(w, x, input_size, hf, dev_index, cores, max_seconds=25.0):
    """Same-run parity: the channel subset the CPU baseline decodes, pushed through a fresh GPU front end
    and through the strict oracle block by block from the start of the stream.  Returns the measured channelizer error and
    whether the decoded PDU multisets are identical."""
    from oracle import pyoracle
    from dumphfdl_amd import frontend as F
    pyoracle.select_build("strict")
    freqs = channel_plan(w)
    cs = min(len(freqs), cores)
    try:
        ram_gib = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2.0 ** 30
    except (ValueError, OSError):
        ram_gib = 0.0
    if (os.cpu_count() or 1) >= 32 and ram_gib >= 128:
        # a host with the cores and the memory for the oracle's filters (8 N bytes per channel: 16 GiB at cfg3) checks EVERY channel
        cs, cores = len(freqs), max(cores, min(os.cpu_count(), 256))
    sel = freqs[:: max(1, len(freqs) // cs)][:cs]
    ora = pyoracle.Frontend(w["fs"], w["centerfreq"], sel, nthreads=cores)
    fe = hf.Frontend(w["fs"], w["centerfreq"], sel, device=dev_index)
    fe.enable_taps(False)
    nblk_max = len(x) // input_size
    t0, nblk, worst = time.time(), 0, 0.0
    check = sorted(set([0, cs // 2, cs - 1]))
    while nblk < nblk_max and (nblk < 2 or time.time() - t0 < max_seconds):
        blk = np.ascontiguousarray(x[nblk * input_size:(nblk + 1) * input_size])
        fe.push_block(blk)
        ora.push_block(blk, nthreads=cores)
        if nblk in (0, 1) or nblk % 5 == 0:
            for c in check:
                a = fe.read_tap(F.TAP_CHAN_OUT, c).astype(np.complex128)
                b = ora.channel_view(c)["chan_out"].astype(np.complex128)
                worst = max(worst, float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), 1e-300))))
        nblk += 1
    got = sorted(pdu_key(p) for p in fe.poll_pdus(16384))
    want = sorted(pdu_key(p) for p in ora.pdus)
    fe.close()
    ora.close()
    # the same comparison with the detection sample allowed to differ by <= 3 (of 5400 per second): after tens of seconds of noise only
    # -- cfg1's 32 s between bursts -- the two implementations' timing loops have random-walked with different last-ulp roundings and
    # can find the same frame, octet for octet, a sample or two apart (tests/test_gpu_parity.py::test_long_idle_then_burst)
    loose = lambda ks: sorted((k[0], k[2], k[3], k[1]) for k in ks)
    near = len(got) == len(want) and all(a[:3] == b[:3] and abs(a[3] - b[3]) <= 3 for a, b in zip(loose(got), loose(want)))
    # ... and where "identical" stops: the same comparison on traffic binned by in-channel SNR (64 channels at 1 Msps, 128 bursts per
    # bin, all eight modes; the full sweep -8 .. +10 dB is tests/test_gpu_low_snr.py)
    low = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import low_snr_parity
        low = [{k: r[k] for k in ("snr_db", "bursts", "gpu_pdus", "oracle_pdus", "common", "gpu_only", "oracle_only", "identical",
                                   "gpu_recovered", "oracle_recovered", "recovered_sets_identical")}
               for r in low_snr_parity.sweep(hf, pyoracle, [-6, -2, 2, 6], bursts_per_channel=2, device=dev_index)]
    except Exception as e:                  # noqa: BLE001 -- an extra, never fatal to the line
        low = dict(error="%s: %s" % (type(e).__name__, e))
    return dict(channels=cs, blocks=nblk, oracle_build="strict (-O3 -ffp-contract=off, no fast-math; bit-pinned parts see tests/golden)",
                low_snr_bins=low,
                chan_out_rel_rms=worst, chan_out_rel_rms_limit=1e-4, chan_out_within_limit=bool(worst <= 1e-4),
                gpu_pdus=len(got), cpu_pdus=len(want), pdu_multisets_identical=bool(got == want),
                pdu_multisets_identical_up_to_3_samples_of_detection=bool(near),
                compared="(freq, sample_index, mode, octets) of every PDU both sides dispatched on these channels and blocks")


def cpu_baseline(w, x, input_size, target_seconds=20.0):
    """Time the oracle (plain-C restatement of the reference path, one worker thread per channel like the reference, built
    with the reference's release flags) on this host: C_s channels of the same geometry, a few blocks; the per-channel part
    is scaled to the full channel count."""
    from oracle import pyoracle
    libs = probe_cpu_libs()
    pyoracle.select_build("fast")
    # 64 threads at most, the channel part scaled to the full channel count.  (Measured once on a 256-thread box with all 256 channels
    # on 256 threads: 14.8 Msamples/s against 44.7 by this estimate -- the oracle's filters, 16 GiB initialised by one thread, then sit
    # on one memory node; the scaled 64-thread figure is the one that does not flatter the GPU.  profiles/r06_experiments.md)
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
    C = ctypes
    spec = np.empty(fe.ddc.fft_size, np.complex64)
    buf = np.zeros(fe.ddc.fft_size, np.complex64)
    # dumphfdl runs its forward FFT on FFT_THREAD_CNT_DEFAULT = 4 FFTW threads (src/fft.h:15, src/fft_fftw.c:9-20): the restated
    # transform is timed on 1 and on 4 threads (six-step split, oracle/csdr_restated.c) and the faster of the two is the baseline's
    fft_try = {}
    first = np.ascontiguousarray(x[:input_size])
    for nt in (1, 4):
        L.orc_set_fft_threads(nt)
        best = 1e9
        for _ in range(3):
            t1 = time.time()
            L.orc_forward_block(buf.ctypes.data_as(C.c_void_p), first.ctypes.data_as(C.c_void_p), C.byref(fe.ddc), spec.ctypes.data_as(C.c_void_p))
            best = min(best, time.time() - t1)
        fft_try[nt] = best
    fft_threads = min(fft_try, key=fft_try.get)
    L.orc_set_fft_threads(fft_threads)
    buf[:] = 0
    while nblk < 2 or (t_all < target_seconds / 2 and nblk < w["blocks"] - 1):
        blk = np.ascontiguousarray(x[(nblk + 1) * input_size:(nblk + 2) * input_size])
        t1 = time.time()
        fe.push_block(blk, nthreads=cores)
        t_all += time.time() - t1
        t1 = time.time()            # the shared forward FFT alone, on the thread count chosen above
        L.orc_forward_block(buf.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), C.byref(fe.ddc), spec.ctypes.data_as(C.c_void_p))
        t_fft += time.time() - t1
        nblk += 1
    per_blk, fft_blk = t_all / nblk, t_fft / nblk
    chan_blk = max(per_blk - fft_blk, 1e-9)          # cs channels on `cores` threads
    fft_used, fft_kind = fft_blk, "oracle radix-4 FFT, %d thread%s (1 thread %.3f s, 4 threads %.3f s)" % (fft_threads, "s" if fft_threads > 1 else "", fft_try[1], fft_try[4])
    if libs["fftw3f"]:
        try:
            # the baseline takes the FASTER of the two: a single-threaded FFTW_ESTIMATE plan must not replace a quicker 4-thread
            # transform (the reference runs FFTW on 4 threads, src/fft.h:15, src/fft_fftw.c:9-20)
            t_fftw = fftw_forward_seconds(fe.ddc.fft_size, libs["fftw3f"])
            if t_fftw < fft_used:
                fft_used, fft_kind = t_fftw, "FFTW3f (%s), FFTW_ESTIMATE, 1 thread (%.3f s; oracle FFT %.3f s)" % (libs["fftw3f"], t_fftw, fft_blk)
            else:
                fft_kind += "; FFTW3f (%s) 1 thread measured slower: %.3f s" % (libs["fftw3f"], t_fftw)
        except Exception as e:            # the probe is best effort: the restated FFT stays the fallback
            fft_kind += " (FFTW found but unusable: %s)" % e
    full = fft_used + chan_blk * (len(freqs) / cs)
    frames = len(fe.pdus)
    fft_size = fe.ddc.fft_size
    fe.close()
    L.orc_set_fft_threads(1)
    pyoracle.select_build("strict")
    return dict(value=input_size / full / 1e6, unit="Msamples/s", cores=cores, kind="port", fft_threads=fft_threads,
                fftw_found=libs["fftw3f"], liquid_found=libs["liquid"],
                build="gcc -O3 -DNDEBUG -ffast-math (the reference's cmake Release flags, CMakeLists.txt:12-15, src/CMakeLists.txt:39-42)",
                forward_fft="%s: %.3f s per %d-point block -- the reference's shape: ONE forward FFT per block (FFTW on 4 threads there) shared by "
                            "all channels, ahead of the per-channel part, which runs on all threads; the FFT is %.0f %% of the CPU block time"
                            % (fft_kind, fft_used, fft_size, 100.0 * fft_used / full),
                sample="oracle (C restatement; dlopen probe: FFTW3f %s, liquid-dsp %s): %d of %d channels x %d blocks of %d samples on %d "
                       "threads, %.3f s/block measured (forward FFT %.3f s, channel part %.3f s), channel part scaled x%.1f to %d channels; "
                       "init %.1f s untimed; %d PDUs decoded in the sample"
                       % ("found" if libs["fftw3f"] else "not found", "found" if libs["liquid"] else "not found",
                          cs, len(freqs), nblk, input_size, cores, per_blk, fft_blk, chan_blk, len(freqs) / cs, len(freqs), t_init, frames))


def fec_capacity(hf, dev_index):
    """The burst decoder on its own: a resident batch of the longest frames (mode 7: 7560 trellis steps, 15120 coded bits),
    one wavefront per frame, kernel time from HIP events (libfec work unit: src/libfec/viterbi27_port.c:166-221)."""
    from dumphfdl_amd import frontend as F
    rng = np.random.default_rng(1)
    nframes, nbits = 1024, 7560
    soft = rng.integers(0, 256, (nframes, 2 * nbits), dtype=np.uint8)
    hf.viterbi27(soft[:8], nbits, device=dev_index)            # first launch: code load, LDS attribute
    hf.viterbi27(soft, nbits, device=dev_index)
    ms = F.last_stage_ms()
    out = dict(viterbi_kernel_frames=nframes, viterbi_kernel_ms=ms,
               viterbi_kernel_trellis_steps_per_s=(nframes * nbits / (ms * 1e-3)) if ms > 0 else None,
               viterbi_kernel_acs_per_s=(nframes * nbits * 64 / (ms * 1e-3)) if ms > 0 else None)
    return out


def host_path_leg(w, x, freqs, fmt="CF32", dev_index=0, seconds_cap=120):
    """The C host library end to end (what a dumphfdl user runs): a raw I/Q file in page cache -> file input -> block graph
    -> GPU front-end block -> pdu_decoder_queue_push.  hfdl_replay --bench prints one JSON object with its own clock."""
    exe = os.path.join(ROOT, "dumphfdl_amd", "hfdl_replay")
    if not os.path.exists(exe):
        return dict(error="dumphfdl_amd/hfdl_replay not built")
    path = None
    try:
        raw = np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16) if fmt == "CS16" else x.view(np.float32)
        for d in ("/dev/shm", "/tmp"):              # page cache either way; /dev/shm can be tiny inside containers
            try:
                path = "%s/hfdl_bench_%d.%s" % (d, os.getpid(), fmt.lower())
                raw.tofile(path)
                break
            except OSError:
                try:
                    os.remove(path)
                except OSError:
                    pass
                path = None
        if path is None:
            return dict(error="no room for the I/Q file in /dev/shm or /tmp")
        loops = max(1, int(np.ceil(1.5 * 2.5e9 / len(x))))        # >= ~1.5 s of work at 2.5 Gsamples/s
        cmd = [exe, "--bench", "--loop", str(loops), "--iq-file", path, "--sample-rate", str(w["fs"]), "--sample-format", fmt, "--device", str(dev_index),
               "--centerfreq", "%.3f" % (w["centerfreq"] / 1e3)] + ["%.3f" % (f / 1e3) for f in freqs]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=seconds_cap)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return dict(error="hfdl_replay --bench failed (rc %d): %s" % (out.returncode, out.stderr[-300:]))
        r = json.loads(line[-1])
        r["sample_format"] = fmt
        r["file_loops"] = loops
        return r
    except Exception as e:
        return dict(error=str(e))
    finally:
        try:
            if path:
                os.remove(path)
        except OSError:
            pass


# the sources that decide the fold launch's memory traffic: the kernel and its launcher, the tap / spectrum layouts and the geometry
FOLD_SOURCES = ("fold_kernels.hip", "kernels.h", "fft_core.h")


def csrc_hash():
    """sha256 (16 hex digits) over the sources of the ROOFLINE KERNEL -- dumphfdl_amd/csrc/{fold_kernels.hip, kernels.h, fft_core.h} in
    that order: identifies the fold a traffic measurement was made with (profiles/fold_traffic.py stamps it into the traffic record; this
    run compares it with its own tree).  Until round 6 the hash ran over every file of csrc/: an edit of the demodulator then withheld a
    traffic figure it cannot have changed."""
    import hashlib
    d = os.path.join(ROOT, "dumphfdl_amd", "csrc")
    h = hashlib.sha256()
    for name in FOLD_SOURCES:
        h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def stream_budget(stages, steps, half_blocks):
    """Kernel time per stream for one half of `half_blocks` blocks: stream A = forward FFTs + fold + inverse FFT / NCO, stream B =
    demodulators, stream D = burst decoders (hfdl_gpu_frontend_stage_times over the timed region, scaled to a half)."""
    if not steps:
        return None
    per_block = {k: (v[0] / steps) for k, v in stages.items()}
    a = per_block["fft"] + per_block["fold"] + per_block["ifft"]
    return dict(half_blocks=half_blocks,
                stream_a_ms=a * half_blocks, stream_b_ms=per_block["demod"] * half_blocks, stream_d_ms=per_block["decode"] * half_blocks,
                per_block_ms={k: round(v, 5) for k, v in per_block.items()},
                launches={k: v[1] for k, v in stages.items()},
                note="A = forward FFT (%.3f ms / block) + fold + inverse FFT / NCO; B = demodulators; D = burst decoders; the streams run beside each other"
                     % per_block["fft"])


def stream_read_leg(w, freqs, dev_index):
    """What the board's HBM delivers to a bare read-only kernel over the same 16 GiB of filter taps: a probe of the LABORATORY build
    (libhfdl_gpu_lab.so, include/hfdl_gpu_lab.h) on a front end of its own -- the product library carries no probes."""
    try:
        from dumphfdl_amd import frontend as F
        fe = F.Frontend(w["fs"], w["centerfreq"], freqs, device=dev_index, lib=F.load_lab())
        try:
            return fe.stream_read_probe()
        finally:
            fe.close()
    except Exception as e:                      # noqa: BLE001 -- the probe is an extra: the line says why it is missing
        sys.stderr.write("bench.py: stream-read probe unavailable: %s\n" % e)
        return None


def traffic_record(workload, shapes=None, launches=None):
    """HBM bytes per fold launch from the PMC passes kept under profiles/ (separate rocprofv3 --pmc runs, corrected as the
    MI355X guide prescribes): a bench run cannot collect counters itself, so the line names where the figure comes from -- and
    whether the kernels it was measured on are the ones this run executes (csrc_matches_head).  `shapes` = this run's timed launches
    by blocks per launch, AS THE LIBRARY COUNTED THEM (hfdl_gpu_frontend_fold_launch_shapes): the figure is the average over them of
    the record's per-shape measurements (a launch of 4 blocks moves other bytes than one of 16).  A record that fails a check is
    reported as stale and `traffic` is null."""
    tfile = os.path.join(ROOT, "profiles", "fold_traffic_%s.json" % workload)
    if not os.path.exists(tfile):
        return None, None
    try:
        t = json.load(open(tfile))
        now = csrc_hash()
        same_code = t.get("csrc_sha16") == now
        per = t.get("per_shape") or ({str(t["blocks_per_launch"]): t} if t.get("blocks_per_launch") else {})
        src = dict(file="profiles/fold_traffic_%s.json" % workload, measured_at_commit=t.get("measured_at_commit"),
                   csrc_sha16_at_measurement=t.get("csrc_sha16"), csrc_sha16_now=now, csrc_matches_head=bool(same_code),
                   shapes_measured=sorted(int(k) for k in per), kernel=t.get("kernel"),
                   collected_by=t.get("command"), note="replayed from that file, not observed by this run")
        if shapes is None:
            shapes = [t.get("blocks_per_launch")]
        src["launch_shapes_of_this_run"] = {str(nb): shapes.count(nb) for nb in sorted(set(shapes))}
        ok = same_code and all(str(nb) in per for nb in shapes) and (launches is None or launches == len(shapes))
        if not ok:
            src["stale"] = "measured on other device code, or this run launched shapes the record does not hold: traffic withheld"
            return None, src
        src["traffic_over_algorithmic"] = {str(nb): per[str(nb)]["traffic_over_algorithmic"] for nb in sorted(set(shapes))}
        return sum(per[str(nb)]["hbm_bytes_per_launch"] for nb in shapes) / len(shapes), src
    except Exception:
        return None, None


def dominant_shape(shape_times):
    """{blocks per launch: (launches, kernel ms)} -> (blocks, launches, average ms) of the launch shape the run spent most of its fold
    time in -- what `roofline` is priced on -- or (0, 0, None) when nothing was timed.  Ties go to the larger shape."""
    if not shape_times:
        return 0, 0, None
    nb = max(shape_times, key=lambda k: (shape_times[k][1], k))
    n, ms = shape_times[nb]
    return nb, n, ms / n


def alg_bytes_per_block(g):
    """B = 8*input_size + C*8*N + C*8*(post_input_size/post_decimation) algorithmic bytes per block."""
    return 8 * g.input_size + g.channels * 8 * g.fft_size + g.channels * 8 * (g.post_input_size // g.post_decimation)


def alg_bytes_per_launch(g, nb):
    """The same model per FOLD LAUNCH when a launch serves `nb` queued blocks with one pass over the filter taps:
    nb blocks of input, the C*N taps ONCE, nb blocks of channel outputs.  nb = 1 gives alg_bytes_per_block."""
    return 8 * nb * g.input_size + g.channels * 8 * g.fft_size + g.channels * 8 * nb * (g.post_input_size // g.post_decimation)


class stdout_to_stderr:
    """Anything the communication libraries print to file descriptor 1 while they come up (RCCL's version banner, gloo's connection
    report) goes to stderr instead: rank 0's stdout carries ONE JSON line and nothing else."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_dist(torch, dist, backend, dev_index, world):
    """torch.distributed carries the timing barrier and the final reductions only (no data-path collective).  `nccl` = RCCL for
    CUDA tensors (with gloo beside it for CPU tensors); if RCCL cannot be brought up the run continues on gloo and says so."""
    note = None
    if backend == "nccl":
        try:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", dev_index))
            t = torch.ones(1, device="cuda")
            dist.all_reduce(t)                      # first RCCL collective: communicator really up, on every rank
            torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError("all_reduce over %d ranks returned %s" % (world, t.item()))
            return "nccl", "cuda", None
        except Exception as e:                      # noqa: BLE001 -- anything RCCL throws: the measurement itself does not need it
            note = "nccl (RCCL) could not be used: %s: %s -- barrier / reductions fell back to gloo" % (type(e).__name__, str(e)[:200])
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:                       # noqa: BLE001
                pass
            # a second rendezvous: next port (rank 0's first store may still hold the old one); every rank computes the same number
            if os.environ.get("MASTER_PORT", "").isdigit():
                os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
    dist.init_process_group("gloo")
    return "gloo", "cpu", note


def timed_blocks(torch, fe, push_fn, steps, first_step, nblocks, prefetch_fn=None):
    """K blocks through the whole path, PDUs back on the host, device idle at the end.  prefetch_fn(i): queue the upload of block i
    (host-memory legs) while block i-1 is pushed, as the C host path does."""
    step = first_step
    raw = []
    t_start = time.perf_counter()
    for i in range(steps):
        push_fn(step % nblocks); step += 1
        if prefetch_fn is not None and i + 1 < steps:
            prefetch_fn(step % nblocks)
        if i % 256 == 255 and i + 1 < steps:      # long runs: empty the device PDU ring now and then, pipeline kept running
            raw.append(fe.poll_pdus_raw(16384, max_in_flight=1))
    raw.append(fe.poll_pdus_raw(16384))     # sync + device->host of every PDU struct produced by the timed blocks (what the C host gets)
    torch.cuda.synchronize()
    return time.perf_counter() - t_start, raw, step


def host_feed(hf, F, fe, x, g, nblocks, fmt_name):
    """The stream in page-locked host memory, one pointer per block; returns (buffer, push(i), prefetch(i))."""
    if fmt_name == "cs16":
        raw = np.clip(np.round(x.view(np.float32) * 20000), -32768, 32767).astype(np.int16)
        fmt, bps = F.SFMT_CS16, 4
    else:
        raw, fmt, bps = x.view(np.float32), F.SFMT_CF32, 8
    hbuf = hf.host_alloc(raw.nbytes)
    ctypes.memmove(hbuf, raw.ctypes.data, raw.nbytes)
    hptrs = [hbuf + bps * b * g.input_size for b in range(nblocks)]
    return hbuf, (lambda i: fe.push_host_ptr(hptrs[i], fmt)), (lambda i: fe.prefetch_host_ptr(hptrs[i], fmt))


def host_ram_leg(torch, hf, F, fe, x, g, nblocks, steps):
    """SURVEY.md 8(d) "input pre-loaded in host RAM": the same blocks from page-locked host memory, the upload of block k+1
    queued while block k is pushed (hfdl_gpu_frontend_prefetch_block_raw) -- PCIe-inclusive."""
    hbuf, hpush, hprefetch = host_feed(hf, F, fe, x, g, nblocks, "cf32")
    k2 = 256          # whatever --steps says: the leg is PCIe-bound and has a ~8 ms drain (the last half's kernels after the last upload) to amortise
    # one untimed pass over EVERY block first: the first DMA out of a freshly page-locked page costs more than the later ones
    # (address translation for the device is set up as pages are first touched), and the stream is replayed several times below
    warm = max(4, nblocks)
    for i in range(warm):
        hpush(i % nblocks)
    fe.poll_pdus()
    hprefetch(warm % nblocks)
    el2, raw2, _ = timed_blocks(torch, fe, hpush, k2, warm, nblocks, prefetch_fn=hprefetch)
    return hbuf, dict(value=k2 * g.input_size / el2 / 1e6, unit="Msamples/s", steps=k2, ms_per_step=el2 / k2 * 1e3,
                      path="cf32 blocks in page-locked host RAM -> hfdl_gpu_frontend_prefetch_block_raw / push_block_raw (copy stream, a ring of "
                           "up to 18 staging buffers in HBM: uploads run up to 17 blocks ahead of the kernels) -> same kernels; PCIe-inclusive",
                      pcie_GBs=k2 * g.input_size * 8 / el2 / 1e9, pdus=sum(n for _, n in raw2))


def cfg2_leg(torch, hf, F, dev_index, steps=256, warmup=8):
    """BASELINE.json configs[1] (8 Msps x 32 channels, the demodulator-bound geometry) next to the headline workload, so that the
    driver's line shows it: resident and host-RAM rates, the demodulator kernel's average from its own dispatch events, the
    steady-state step against the HBM roofline, and every PDU checked against the sent traffic."""
    w = WORKLOADS["cfg2"]
    freqs = channel_plan(w)
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=dev_index)
    g = fe.geometry
    fe.enable_taps(False)
    x, bursts = make_input(w, g.input_size, 0, 1)
    nblocks = len(x) // g.input_size
    by_freq = {}
    for b in bursts:
        by_freq.setdefault(b["freq"], []).append(b)
    dev = torch.from_numpy(x.view(np.float32)).cuda()
    ptrs = [dev.data_ptr() + 8 * b * g.input_size for b in range(nblocks)]
    push = lambda i: fe.push_block(ptrs[i])
    step = 0
    for _ in range(warmup):
        push(step % nblocks); step += 1
    fe.poll_pdus()
    fe.reset_timers(True)
    torch.cuda.synchronize()
    el, raw, step = timed_blocks(torch, fe, push, steps, step, nblocks)
    pdus = [p for buf, n in raw for p in fe.pdus_to_dicts(buf, n)]
    fold_ms, fold_n = fe.fold_time_ms()
    fold_blk = fe.fold_blocks()
    dm_ms, dm_n, dm_blk = fe.demod_time_ms()
    period = fe.step_period_ms()
    fe.reset_timers(False)
    hbuf, host = host_ram_leg(torch, hf, F, fe, x, g, nblocks, steps)
    fe.close()
    del dev
    hf.host_free(hbuf)
    nb = (fold_blk / fold_n) if fold_n else 1.0
    ab = alg_bytes_per_launch(g, nb) / nb          # per block, the taps shared by the blocks of a launch
    return dict(workload=w["name"], value=steps * g.input_size / el / 1e6, unit="Msamples/s", steps=steps, warmup=warmup, ms_per_step=el / steps * 1e3,
                steady_state_ms_per_step=period, block_samples=g.input_size, channels=g.channels,
                demod_kernel_ms_per_block=(dm_ms / dm_blk) if dm_blk else None, demod_kernel_launches=dm_n, demod_blocks_per_launch=g.demod_batch,
                fold_kernel_avg_ms=(fold_ms / fold_n) if fold_n else None, fold_blocks_per_launch=nb,
                algorithmic_bytes_per_block=ab, algorithmic_bytes_per_block_unbatched=alg_bytes_per_block(g),
                whole_step_frac_of_hbm_peak=(ab / (period * 1e-3) / 1e9 / HBM_PEAK_GBS) if period else None,
                bound="demod_kernel: a serial recurrence per channel (latency), not HBM",
                pdus=len(pdus), pdus_matching_sent_payload=sum(1 for p in pdus if matches_sent(p, by_freq)),
                pdus_lpdu_walk_matching_sent=sum(1 for p in pdus if lpdu_walk_matches_sent(p, by_freq)),
                value_host_ram=host["value"], host_ram_input=host)


PRUNE_TOL = 3e-7          # what the skipped alias rows may hold of a filter's energy, as an amplitude ratio (the taps' own rounding noise: 2.1e-7)


def pruned_fold_leg(torch, hf, F, w, freqs, x, dev_index, steps, warmup, ref_pdus):
    """The OPT-IN pruned fold (HFDL_GPU_FOLD_PRUNE, include/hfdl_gpu.h) next to the headline: the same workload, the same blocks, the
    same timed loop with a front end that folds only the alias rows around each channel's pass band.  `value` of the bench line is
    NOT this figure: the default folds every row, as the reference does.  Checked here, in the same run: the channelizer output of
    ten channels against a full-fold front end on the same block (relative RMS), and the PDUs of the timed region against those of
    the headline run (ref_pdus: same blocks in the same order)."""
    watch = [c for c in (0, 1, 7, 8, 100, 127, 128, 200, 254, 255) if c < len(freqs)]

    def create(tol):
        if tol:
            os.environ["HFDL_GPU_FOLD_PRUNE"] = repr(tol)
        try:
            fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=dev_index)
        finally:
            os.environ.pop("HFDL_GPU_FOLD_PRUNE", None)
        return fe

    fe = create(0)
    g = fe.geometry
    fe.channelize_block(x[:g.input_size])
    full = [fe.read_tap(F.TAP_CHAN_OUT, c).astype(np.complex128) for c in watch]
    fe.close()
    fe = create(PRUNE_TOL)
    g = fe.geometry
    fe.channelize_block(x[:g.input_size])
    err = max(float(np.sqrt(np.mean(np.abs(fe.read_tap(F.TAP_CHAN_OUT, c) - f) ** 2) / np.mean(np.abs(f) ** 2))) for c, f in zip(watch, full))
    fe.close()
    fe = create(PRUNE_TOL)
    fe.enable_taps(False)
    nblocks = len(x) // g.input_size
    dev = torch.from_numpy(x.view(np.float32)).cuda()
    ptrs = [dev.data_ptr() + 8 * b * g.input_size for b in range(nblocks)]
    push = lambda i: fe.push_block(ptrs[i])
    step = 0
    for _ in range(warmup):
        push(step % nblocks); step += 1
    fe.poll_pdus()
    fe.reset_timers(True)
    torch.cuda.synchronize()
    el, raw, step = timed_blocks(torch, fe, push, steps, step, nblocks)
    pdus = [p for buf, n in raw for p in fe.pdus_to_dicts(buf, n)]
    fold_ms, fold_n = fe.fold_time_ms()
    dm_ms, dm_n, dm_blk = fe.demod_time_ms()
    period = fe.step_period_ms()
    stages = fe.stage_times()
    fe.reset_timers(False)
    fe.close()
    del dev
    key = lambda p: (p["freq"], p["mode"], p["octets"], p["fcs_status"])
    a = sorted(ref_pdus, key=lambda p: (p["freq"], p["sample_index"]))
    b = sorted(pdus, key=lambda p: (p["freq"], p["sample_index"]))
    same = len(a) == len(b) and [key(p) for p in a] == [key(p) for p in b]
    return dict(what="opt-in (HFDL_GPU_FOLD_PRUNE=%g): only the alias rows outside which a channel's filter holds < tol^2 of its energy are folded; "
                     "NOT the headline -- `value` folds every row, as src/fastddc.c:123-150 does" % PRUNE_TOL,
                tolerance=PRUNE_TOL, fold_rows=g.fold_rows, of_rows=g.pre_decimation,
                value=steps * g.input_size / el / 1e6, unit="Msamples/s", steps=steps, ms_per_step=el / steps * 1e3, steady_state_ms_per_step=period,
                fold_kernel_avg_ms=(fold_ms / fold_n) if fold_n else None, demod_kernel_ms_per_block=(dm_ms / dm_blk) if dm_blk else None,
                streams=stream_budget(stages, steps, g.fold_batch),
                chan_out_rel_rms_vs_full_fold=err, pdus=len(pdus), pdus_same_as_full_fold=same,
                detection_sample_max_abs_diff=(max([abs(p["sample_index"] - q["sample_index"]) for p, q in zip(a, b)] or [0]) if same else None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="timed blocks (one step = one block of input_size samples through the whole path); 256 blocks of the 40 Msps geometry = 0.75 s")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline and the same-run parity gate")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip host_ram_input / host_path / fec / cfg2 (profiling runs)")
    ap.add_argument("--host-input", action="store_true",
                    help="feed the TIMED blocks from page-locked HOST memory (PCIe-inclusive rate; then `value` is not the headline figure)")
    ap.add_argument("--sample-format", default="cf32", choices=["cf32", "cs16"], help="with --host-input: raw format pushed over PCIe")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / final reduction (nccl = RCCL; gloo for a 1-GPU smoke of the N>1 path)")
    ap.add_argument("--shard", default="streams", choices=["streams", "channels"],
                    help="N > 1: independent stream per rank (weak scaling, BASELINE configs[4]) or ONE stream with its channels split round-robin over the ranks (strong scaling, SURVEY 8e)")
    ap.add_argument("--dump-pdus", default=None, help="write this rank's PDU keys (freq, sample_index, mode, octets) to PATH.rankR.json")
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
    # launched by torch.distributed.run (any world size, 1 included): the process group is brought up and every barrier /
    # reduction below runs through it, so the RCCL path executes on a 1-GPU box exactly as it will on 8
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    dist_info = None
    if world > 1 or launched:
        backend, shared_note = args.backend, None
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if backend == "nccl" and local_world > torch.cuda.device_count():
            # RCCL does not refuse two ranks on one device, it hangs in the first collective (measured: profiles/r03_experiments.md):
            # ranks that share a GPU talk through gloo, and the line says so
            backend = "gloo"
            shared_note = "%d local ranks on %d visible GPU(s): RCCL cannot run ranks that share a device -- gloo used" % (local_world, torch.cuda.device_count())
        with stdout_to_stderr():
            used, red_device, note = init_dist(torch, dist, backend, dev_index, world)
        dist_info = dict(backend=used, requested=args.backend, world_size=world, fallback=note or shared_note)
        # all ranks build their filter taps on the host at once: share the cores
        os.environ.setdefault("HFDL_GPU_HOST_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, world))))
    import dumphfdl_amd as hf
    from dumphfdl_amd import shard
    from dumphfdl_amd import frontend as F
    use_dist = dist if dist_info else None

    all_freqs = channel_plan(w)
    freqs = shard.shard_channels(all_freqs, rank, world) if args.shard == "channels" else all_freqs
    t0 = time.time()
    fe = hf.Frontend(w["fs"], w["centerfreq"], freqs, device=dev_index)
    g = fe.geometry
    fe.enable_taps(False)            # per-stage debug taps (DATADUMPS analogue) are a test facility, not part of the path
    t_create = time.time() - t0
    t0 = time.time()
    x, bursts = make_input(w, g.input_size, rank, world, args.shard)
    my_seed = shard.stream_seed(w["seed"], rank, world) if args.shard == "streams" else w["seed"]
    t_gen = time.time() - t0
    # A rank whose set-up crawls (eight front ends designing filter taps on one host's cores, eight syntheses behind one lock) would
    # otherwise surface as a barrier time-out minutes later: every rank learns the slowest rank's times, and the job stops with a message.
    budget = float(os.environ.get("HFDL_BENCH_SETUP_BUDGET_S", "600"))
    worst_create, worst_gen = (t_create, t_gen)
    if use_dist is not None:
        t = torch.tensor([t_create, t_gen], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst_create, worst_gen = float(t[0]), float(t[1])
    if worst_create + worst_gen > budget:
        if rank == 0:
            sys.stderr.write("bench.py: set-up took %.0f s on the slowest rank (front end %.0f s + input synthesis %.0f s), over the %.0f s budget "
                             "(HFDL_BENCH_SETUP_BUDGET_S): not starting the timed region. Fewer ranks per host, or more host cores per rank "
                             "(HFDL_GPU_HOST_THREADS), bring it down; profiles/r04_setup_time.json has the single-rank figures.\n"
                             % (worst_create + worst_gen, worst_create, worst_gen, budget))
        fe.close()
        if use_dist is not None:
            with stdout_to_stderr():
                dist.barrier()
                dist.destroy_process_group()
        raise SystemExit(3)
    nblocks = len(x) // g.input_size
    bursts_by_freq = {}
    for b in bursts:
        bursts_by_freq.setdefault(b["freq"], []).append(b)

    hbuf = None
    if args.host_input:
        hbuf, push, _ = host_feed(hf, F, fe, x, g, nblocks, args.sample_format)
        dev = None
    else:
        dev = torch.from_numpy(x.view(np.float32)).cuda()          # resident in HBM before the timed region
        ptrs = [dev.data_ptr() + 8 * b * g.input_size for b in range(nblocks)]
        push = lambda i: fe.push_block(ptrs[i])

    def barrier():
        torch.cuda.synchronize()
        if use_dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        push(step % nblocks); step += 1
    fe.poll_pdus()
    fe.reset_timers(True)
    barrier()
    elapsed, raw, step = timed_blocks(torch, fe, push, args.steps, step, nblocks)
    pdus = [p for buf, n in raw for p in fe.pdus_to_dicts(buf, n)]     # Python-side unpacking for the checks below: not part of the path
    npdus = len(pdus)
    fold_ms, fold_n = fe.fold_time_ms()
    fold_blk = fe.fold_blocks()
    dm_ms, dm_n, dm_blk = fe.demod_time_ms()
    period_ms = fe.step_period_ms()
    shape_times = fe.fold_launch_times()                   # {blocks per launch: (timed launches, their kernel ms)}
    shapes = {nb: c for nb, (c, _) in shape_times.items()}
    stages = fe.stage_times()                              # kernel time by stage, from the dispatches' own events
    barrier()
    fe.reset_timers(False)
    good = sum(1 for p in pdus if matches_sent(p, bursts_by_freq))
    lpdu_ok = sum(1 for p in pdus if lpdu_walk_matches_sent(p, bursts_by_freq))
    lpdus_good = sum(p["lpdus"][1] for p in pdus)
    trellis = sum(NBITS[p["mode"]] for p in pdus)
    my_samples = args.steps * g.input_size
    if args.shard == "channels" and rank != 0:
        my_samples = 0                               # ONE stream: its samples count once
    elapsed_max, total_samples, total_pdus = shard.reduce_job(elapsed, my_samples, npdus, use_dist, device=red_device)
    total_good, total_trellis, total_lpdu_ok, total_lpdus = shard.reduce_sums([good, trellis, lpdu_ok, lpdus_good], use_dist, device=red_device)
    seeds = shard.gather_ints(my_seed, use_dist, device=red_device)
    fold_avg_ms = fold_ms / max(fold_n, 1)
    # every rank's own numbers, in rank order: a straggler GPU shows up here, not only in the max
    per_rank_cols = shard.gather_floats([my_seed, elapsed / args.steps * 1e3, period_ms, fold_avg_ms, (dm_ms / dm_blk) if dm_blk else 0.0,
                                         npdus, good, g.channels, t_create, t_gen], use_dist, device=red_device)
    if args.dump_pdus:
        json.dump(sorted(pdu_key(p) for p in pdus), open("%s.rank%d.json" % (args.dump_pdus, rank), "w"))

    # ---- extra legs (untimed with respect to `value`; rank 0 at N = 1)
    extra = {}
    solo = world == 1 and rank == 0
    if solo:
        # a live receiver's latency: ONE block pushed into an idle pipeline until its PDUs are on the host (forward FFT, a one-block
        # fold in the four-column form, inverse FFT, demodulator, burst decoder, collection): the median of five
        lat = []
        for i in range(5):
            t0 = time.perf_counter()
            push((step + i) % nblocks)
            fe.poll_pdus_raw(16384)
            lat.append((time.perf_counter() - t0) * 1e3)
        extra["block_to_pdus_latency_ms"] = sorted(lat)[2]
        fe.reset_timers(False)
    if solo and not args.no_extra_legs and not args.host_input:
        hbuf, extra["host_ram_input"] = host_ram_leg(torch, hf, F, fe, x, g, nblocks, args.steps)
    if solo and not args.no_extra_legs:
        extra["fec"] = fec_capacity(hf, dev_index)
    geom = dict(channels=g.channels, fft_size=g.fft_size, fft_inv_size=g.fft_inv_size, input_size=g.input_size)
    demod_batch = g.demod_batch
    # the roofline is priced on ONE launch shape: the one the run spent most of its fold time in (the full halves; the ragged end of a
    # run is a launch of its own -- up to 4 blocks: the four-column form of the kernel -- and listed beside it)
    dom, dom_n, dom_ms = dominant_shape(shape_times)
    fold_nb = float(dom) if dom else 1.0
    alg_bytes = alg_bytes_per_launch(g, fold_nb)
    alg_bytes_block = alg_bytes / fold_nb
    alg_bytes_unbatched = alg_bytes_per_block(g)
    fold_batch = g.fold_batch
    # every rank releases its front end (16 GiB of filter taps each) before the legs that build others / before leaving
    fe.close()
    del dev
    if hbuf:
        hf.host_free(hbuf)
    stream_gbs = stream_read_leg(w, freqs, dev_index) if rank == 0 else None       # after the timed regions: the board's own read ceiling

    if rank == 0:
        samples = total_samples
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom else None
        fold_flops = 8.0 * geom["channels"] * geom["fft_size"] * fold_nb
        tflops = (fold_flops / (dom_ms * 1e-3) / 1e12) if dom else None
        # both sides of the launch; `bound` / `frac` = the side that is nearer its peak (round 5 always priced the wide forms on the
        # matrix pipe, which understated a launch whose operational intensity lies left of the ridge)
        mfma_frac = (tflops / FP32_MFMA_PEAK_TFLOPS) if tflops else None
        hbm_frac = (achieved / HBM_PEAK_GBS) if achieved else None
        wide = bool(dom) and mfma_frac >= hbm_frac
        intensity = (fold_flops / alg_bytes) if dom else None
        ridge = FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        fold_form = lambda nb: "4 columns (4x4x1_16B)" if nb <= 4 else "16 columns (16x16x4)" if nb <= 16 else "32 columns (16x16x4, two spectrum operands per tap operand)"
        traffic, traffic_src = traffic_record(args.workload, [dom] * dom_n, dom_n) if dom else (None, None)
        fold_total_ms = sum(ms for _, ms in shape_times.values())
        par = ("%d independent %d-channel streams, one per GPU, no collectives" % (world, geom["channels"])) if args.shard == "streams" else \
              ("ONE %d-channel stream, channels round-robin over %d GPUs (%d on rank 0), every GPU ingests the same block; no collectives"
               % (len(all_freqs), world, geom["channels"]))
        names = ("stream_seed", "ms_per_step", "steady_state_ms_per_step", "fold_avg_ms", "demod_ms_per_block", "pdus", "pdus_matching_sent_payload",
                 "channels", "frontend_create_s", "input_synthesis_s")
        per_rank = [dict([("rank", r)] + [(n, (int(v) if n in ("stream_seed", "pdus", "pdus_matching_sent_payload", "channels") else round(v, 4)))
                                           for n, v in zip(names, row)]) for r, row in enumerate(per_rank_cols)]
        host_ram_value = extra.get("host_ram_input", {}).get("value")
        out = {
            "metric": "wideband I/Q Msamples/s (cf32 ingest -> decoded HFDL PDUs) at fixed channel count",
            "value": samples / elapsed_max / 1e6, "unit": "Msamples/s",
            # which of SURVEY.md 8(d)'s two regimes `value` is, and the other one next to it as a first-class key
            "value_definition": ("input resident in HBM before the timed region (the bench contract's definition); SURVEY.md 8(d) quotes the "
                                 "metric with the input pre-loaded in HOST RAM: that PCIe-inclusive figure is `value_host_ram`") if not args.host_input
                                else "input in page-locked host RAM, PCIe-inclusive (--host-input): SURVEY.md 8(d)'s own definition; not the HBM-resident headline",
            "value_host_ram": (samples / elapsed_max / 1e6) if args.host_input else host_ram_value,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.shard == "streams" else "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic, resident in HBM before the timed region" if not args.host_input
                    else "synthetic, fed from page-locked host memory as %s (PCIe-inclusive)" % args.sample_format,
            "config": {"workload": w["name"], "sample_rate": w["fs"], "channels": len(all_freqs) if args.shard == "channels" else geom["channels"],
                       "channels_rank0": geom["channels"], "fft_size": geom["fft_size"],
                       "fft_inv_size": geom["fft_inv_size"], "block_samples": geom["input_size"], "resident_blocks": nblocks,
                       "stream_seeds": seeds, "shard": args.shard, "parallelism": par},
            "distributed": dist_info,
            "per_rank": per_rank,
            "frames_per_s": total_pdus / elapsed_max, "pdus_in_timed_region": total_pdus,
            "pdus_matching_sent_payload": total_good,
            "pdus_lpdu_walk_matching_sent": total_lpdu_ok,      # device-side parse_lpdu_list + per-LPDU FCS = what was sent
            "lpdus_good_on_device": total_lpdus,
            "pdus_rank0_fcs_good_on_device": sum(1 for p in pdus if p["fcs_status"] == 0),
            # fill / drain kept visible at any --steps: the steady-state period comes from the fold launches' own start events
            "steady_state_ms_per_step": period_ms,
            "fill_drain_ms": max(0.0, elapsed * 1e3 - period_ms * args.steps) if period_ms else None,
            "trellis_steps_per_s_in_run": total_trellis / elapsed_max,
            "demod_kernel_ms_per_block": (dm_ms / dm_blk) if dm_blk else None, "demod_blocks_per_launch": demod_batch,
            # who bounds a half: kernel time per stream and half of `fold_batch` blocks, summed from the dispatches'
            # own events over the timed region (streams overlap: the largest is the bound, the sum is not the step)
            "streams": stream_budget(stages, args.steps, fold_batch),
            # The dominant kernel: the fold.  It multiplies ONE pass over the filter taps into the spectra of up to 32 queued blocks on the
            # fp32 matrix pipe (sixteen columns of the instruction per group of up to 16 blocks, all of them always computed).  Both sides of
            # the launch are priced -- `mfma` (dense fp32 matrix peak) and `hbm` -- and `bound` / `frac` name the side nearer its peak: the
            # 32-block launches of a long run lie right of the ridge (28.7 flop/B), the 16-block launch of a short run left of it, launches
            # of at most four blocks (the four-column form) are bound by the HBM reads of the taps.
            "roofline": {"bound": "mfma" if wide else "hbm",
                         "kernel": "fold_mfma16_kernel, %s" % fold_form(dom) if dom else None,
                         "achieved": tflops if wide else achieved, "peak": FP32_MFMA_PEAK_TFLOPS if wide else HBM_PEAK_GBS, "unit": "TFLOP/s" if wide else "GB/s",
                         "frac": (mfma_frac if wide else hbm_frac) if dom else None,
                         "frac_is": "max(mfma.frac, hbm.frac): the side of the launch nearer its peak names the bound",
                         "operational_intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                         # The launch sits on the board's POWER budget: its time is the matrix time plus the memory time at the clock the two
                         # leave each other (four / sixteen / thirty-two columns: 0.5 / 1.9 / 3.8 ms of multiplies + 2.2 ms of taps;
                         # profiles/r06_experiments.md), so the two utilisations ADD towards ~1 instead of either reaching it.
                         "utilisation_sum": (mfma_frac + hbm_frac) if dom else None,
                         "priced_on": ("the %d-block launches: %d of the timed region's %d fold launches, %.0f %% of its fold kernel time"
                                       % (dom, dom_n, fold_n, 100.0 * dom_ms * dom_n / max(fold_total_ms, 1e-9))) if dom else None,
                         "algorithmic_flops_per_launch": fold_flops,
                         "flops_model": "8 flops per complex multiply-accumulate x channels x fft_size x blocks_per_launch (src/fastddc.c:114-150 run for that many blocks)",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": dom_ms, "launches": dom_n, "blocks_per_launch": dom, "fold_batch": fold_batch,
                         "launch_shapes": {str(nb): {"launches": c, "avg_ms": ms / c, "form": fold_form(nb)}
                                           for nb, (c, ms) in sorted(shape_times.items())},
                         "mfma": {"achieved": tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_frac},
                         "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                                 "algorithmic_bytes_per_launch": alg_bytes,
                                 "model": "8*NB*input_size + C*8*N + C*8*NB*outputs_per_block per launch: NB queued blocks share ONE pass over the C*N filter "
                                          "taps (NB = blocks_per_launch; NB = 1 is SURVEY.md 8(d)'s per-block figure, algorithmic_bytes_per_block_unbatched)",
                                 "algorithmic_bytes_per_block": alg_bytes_block, "algorithmic_bytes_per_block_unbatched": alg_bytes_unbatched,
                                 "literal_bytes_per_launch": 16 * geom["fft_size"] * (geom["channels"] + fold_nb),      # SURVEY 8(d) secondary figure, NB spectra
                                 "stream_read_GBs": stream_gbs,
                                 "frac_of_stream_read": (achieved / stream_gbs) if (achieved and stream_gbs) else None,
                                 "whole_step_frac": (alg_bytes_block / (period_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if period_ms else None}},
            "setup_s": {"frontend_create": round(t_create, 2), "input_synthesis": round(t_gen, 2)},
        }
        out.update(extra)
        if solo and not args.no_extra_legs:
            out["host_path"] = host_path_leg(w, x, all_freqs, "CF32", dev_index)
            if args.workload != "cfg2" and not args.host_input:
                try:
                    out["cfg2"] = cfg2_leg(torch, hf, F, dev_index)
                except Exception as e:                  # noqa: BLE001 -- a secondary leg never takes the headline line down
                    out["cfg2"] = dict(error="%s: %s" % (type(e).__name__, e))
            if args.workload in ("cfg3", "cfg4") and not args.host_input:
                try:
                    out["pruned_fold"] = pruned_fold_leg(torch, hf, F, w, freqs, x, dev_index, args.steps, args.warmup, pdus)
                except Exception as e:                  # noqa: BLE001
                    out["pruned_fold"] = dict(error="%s: %s" % (type(e).__name__, e))
        if solo and not args.no_cpu_baseline:
            cores = max(1, min(os.cpu_count() or 1, 64))
            out["parity"] = parity_gate(w, x, geom["input_size"], hf, dev_index, cores)
            out["cpu_baseline"] = cpu_baseline(w, x, geom["input_size"])
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist is not None:
        with stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
