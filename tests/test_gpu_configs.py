"""BASELINE.json configurations at their full workload size, through the C ABI on one MI355X:
configs[3] (burst-dense, 256 channels x 40 Msps) and the N > 1 launch path of configs[4] (two ranks sharing the one GPU of
the test box, gloo for the barrier / reduction -- the data path has no collective)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hfdl_synth as synth
from dumphfdl_amd import frontend as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-4


def rel_rms(a, b):
    a = np.asarray(a, np.complex128)
    b = np.asarray(b, np.complex128)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), 1e-300)))


def test_cfg1_through_the_c_host_program(gpu, oracle, tmp_path):
    """BASELINE.json configs[0] exactly as SURVEY.md 8(d) words it (bench.WORKLOADS["cfg1"]: 250 ksps cf32 file, ONE channel at centre
    + 37 kHz, 60 s, a single-slot 300 bps SPDU per 32 s frame, 15 dB Es/N0, seed 1) through `hfdl_replay --iq-file` -- the reference's
    own invocation shape (README.md:876-902, src/main.c:626-651) -- against the oracle on the same file: the same PDUs, both SPDUs
    recovered.  One channel: the fold runs its single-pair shape (the channel padded to a pair, single-wave workgroups only)."""
    sys.path.insert(0, ROOT)
    import bench
    from test_gpu_parity import _replay
    w = bench.WORKLOADS["cfg1"]
    freqs = bench.channel_plan(w)
    assert freqs == [w["centerfreq"] + 37_000]
    g = F.plan_geometry(32, 250 / w["fs"])
    x, bursts = bench.make_input(w, g.input_size, 0, 1)
    assert len(bursts) == 2 and abs(len(x) / w["fs"] - 60.0) < 0.2
    got = _replay(tmp_path, x.view(np.float32), "CF32", w["fs"], w["centerfreq"], freqs)
    ora = oracle.Frontend(w["fs"], w["centerfreq"], freqs)
    n = ora.ddc.input_size
    assert n == g.input_size
    for b in range(len(x) // n):
        ora.push_block(x[b * n:(b + 1) * n])
    want = [(p["freq"], p["bit_rate"], p["slot"], p["octets"]) for p in ora.pdus]
    assert sorted(got) == sorted(want)
    assert sorted(o[:66] for _, _, _, o in got) == sorted(b["octets"] for b in bursts) and all(r == 300 and s == "S" for _, r, s, _ in got)
    # and through the C ABI directly with the bench's own checks: every PDU a sent payload, FCS good on the device
    fe = gpu.Frontend(w["fs"], w["centerfreq"], freqs)
    assert fe.geometry.channels == 1
    for b in range(len(x) // n):
        fe.push_block(x[b * n:(b + 1) * n])
    pdus = fe.poll_pdus()
    by_freq = {freqs[0]: bursts}
    assert len(pdus) == 2 and all(bench.matches_sent(p, by_freq) and p["fcs_status"] == 0 for p in pdus)
    # octets, mode and frequency exactly; the detection sample of the SECOND burst may differ by a sample or two: it follows 32 s of
    # noise through which the two implementations' timing loops have random-walked with different last-ulp roundings
    # (tests/test_gpu_parity.py::test_long_idle_then_burst states the same bound)
    a = sorted((p["freq"], p["mode"], p["octets"], p["sample_index"]) for p in pdus)
    b = sorted((p["freq"], p["mode"], p["octets"], p["sample_index"]) for p in ora.pdus)
    assert [t[:3] for t in a] == [t[:3] for t in b] and all(abs(x[3] - y[3]) <= 3 for x, y in zip(a, b)), (a, b)
    fe.close()


def test_full_size_cfg4_burst_dense(gpu, oracle):
    """BASELINE.json configs[3] as bench.py runs it: 256 channels x 40 Msps, every channel back-to-back bursts cycling all
    eight modes, 32 blocks (235 M samples).  Every PDU carries a sent payload with its mode and a good on-device FCS,
    nothing is dropped, each channel delivers all of its bursts that end inside the stretch, and the oracle -- on all 256 channels where
    the host has the cores and the memory, on a 16-channel subset elsewhere -- yields the identical (freq, sample_index, mode, octets) set
    and channelizer output."""
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["cfg4"])
    freqs = bench.channel_plan(w)
    fe = gpu.Frontend(w["fs"], w["centerfreq"], freqs)
    g = fe.geometry
    assert (g.channels, g.fft_size, g.fft_inv_size, g.input_size) == (256, 1 << 23, 4096, 7340032)
    x, bursts = bench.make_input(w, g.input_size, 0, 1)
    nblk = len(x) // g.input_size
    assert nblk >= 32 and {b["mode"] for b in bursts} == set(range(8))
    sub = [0, 17, 37, 60, 90, 111, 127, 128, 150, 171, 190, 205, 222, 239, 254, 255]        # sixteen since round 6 (round 5: eight)
    cores = os.cpu_count() or 8
    try:
        ram_gib = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2.0 ** 30
    except (ValueError, OSError):
        ram_gib = 0.0
    if cores >= 32 and ram_gib >= 128:
        sub = list(range(256))         # a host large enough (the GPU boxes: 256 cores, 3 TiB) runs the oracle on ALL channels: 16 GiB of oracle filters
    nthr = max(8, min(128, cores))
    ora = oracle.Frontend(w["fs"], w["centerfreq"], [freqs[c] for c in sub], nthreads=nthr)
    fe.enable_taps(False)
    pdus, worst = [], 0.0
    for b in range(nblk):
        blk = x[b * g.input_size:(b + 1) * g.input_size]
        fe.push_block(blk)
        ora.push_block(blk, nthreads=nthr)
        if b in (0, 9, nblk - 1):
            for i, c in enumerate(sub):
                worst = max(worst, rel_rms(fe.read_tap(F.TAP_CHAN_OUT, c), ora.channel_view(i)["chan_out"]))
        elif b % 4 == 3:
            pdus += fe.poll_pdus(max_in_flight=1)          # the pipelined collection path, as the C host uses it
    pdus += fe.poll_pdus()
    cnt = fe.counters()
    assert cnt["pdus_dropped"] == 0 and cnt["pdus_taken"] == len(pdus) and cnt["blocks"] == nblk
    assert worst < RMS_TOL, worst
    by_freq = {}
    for b in bursts:
        by_freq.setdefault(b["freq"], []).append(b)
    seen_modes = set()
    per_chan = {f: 0 for f in freqs}
    for p in pdus:
        match = [b for b in by_freq[p["freq"]] if p["octets"][:len(b["octets"])] == b["octets"]]
        assert len(match) == 1 and match[0]["mode"] == p["mode"], (p["freq"], p["mode"])
        assert len(p["octets"]) == synth.mode_sizes(p["mode"])["octets"]
        assert p["fcs_status"] == F.FCS_GOOD
        assert p["slot"] == ("D" if p["mode"] >= 4 else "S")
        seen_modes.add(p["mode"])
        per_chan[p["freq"]] += 1
    assert seen_modes == set(range(8))
    # No burst is decoded twice, and nearly every burst whose last symbol (+ the demodulator's pipeline delay) lies inside
    # the stretch is decoded: the reference's preamble search itself misses a burst now and then (the oracle reports
    # "M1_not_found" on channel 127 of this very input), which is why the exact gate is the oracle comparison below.
    dur = len(x) / w["fs"]
    due = 0
    for f in freqs:
        due += sum(1 for b in by_freq[f] if b["t0"] + synth.burst_symbols_len(b["mode"]) / 1800 < dur - 0.06)
        assert per_chan[f] <= len(by_freq[f]), (f, per_chan[f])
    assert len(pdus) >= 0.97 * due and len(pdus) >= 300, (len(pdus), due)
    key = lambda p: (p["freq"], p["sample_index"], p["mode"], p["octets"])
    got16 = sorted(key(p) for p in pdus if p["channel"] in sub)
    assert got16 == sorted(key(p) for p in ora.pdus) and len(got16) >= 16
    # trellis work of the run (decoded bits per frame x 64 ACS)
    steps = sum(synth.mode_sizes(p["mode"])["nbits"] for p in pdus)
    assert steps > 500_000
    fe.close()


def test_pruned_fold_is_opt_in_and_within_an_ulp(gpu, oracle, monkeypatch):
    """HFDL_GPU_FOLD_PRUNE (include/hfdl_gpu.h): off by default -- geometry.fold_rows = pre_decimation, every alias row is folded, the
    reference's sum term for term (src/fastddc.c:123-150).  With a tolerance of 3e-7 (the taps' own rounding noise is 2.1e-7) at the cfg4 workload's full
    size (256 channels x 40 Msps, every channel busy in all eight modes) a workgroup folds under a sixth of the 2048 rows; the
    channelizer output of every channel watched moves by < 1.5e-6 relative RMS against the full fold (folding ALL rows in one slice
    instead of four moves it by 5e-7 already: a 2048-term fp32 sum; the distance to the float64 oracle is 1e-6 and does not grow) and stays within the oracle gate, and the PDUs -- (freq, mode, octets, FCS) -- are those of the full fold, the
    detection sample within the +-3 samples two timing loops with different last-ulp roundings are allowed.  A second geometry with a
    left-over octet (2.4 Msps x 132 channels = 16 pairs of octets + one octet) covers the single-octet window."""
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["cfg4"])
    freqs = bench.channel_plan(w)
    watch = [0, 1, 7, 8, 100, 127, 128, 200, 254, 255]

    def run(tol, fs, cf, fr, xs, nblk_out=(0, 5)):
        if tol:
            monkeypatch.setenv("HFDL_GPU_FOLD_PRUNE", repr(tol))
        else:
            monkeypatch.delenv("HFDL_GPU_FOLD_PRUNE", raising=False)
        fe = gpu.Frontend(fs, cf, fr)
        g = fe.geometry
        n = g.input_size
        outs = {}
        for b in range(len(xs) // n):
            fe.push_block(xs[b * n:(b + 1) * n])
            if b in nblk_out:
                outs[b] = [fe.read_tap(F.TAP_CHAN_OUT, c).copy() for c in watch if c < len(fr)]
        pdus = fe.poll_pdus()
        rows = (g.fold_rows, g.pre_decimation)
        fe.close()
        return outs, pdus, rows

    g = F.plan_geometry(4096, 250 / w["fs"])
    assert g.input_size == 7340032
    x, bursts = bench.make_input(w, g.input_size, 0, 1)
    full, pdus_full, rows_full = run(0, w["fs"], w["centerfreq"], freqs, x, (0, 9, 31))
    pruned, pdus_pruned, rows_pruned = run(3e-7, w["fs"], w["centerfreq"], freqs, x, (0, 9, 31))
    assert rows_full == (2048, 2048) and rows_pruned[1] == 2048 and 16 <= rows_pruned[0] <= 320, (rows_full, rows_pruned)
    worst = max(rel_rms(a, b) for blk in full for a, b in zip(pruned[blk], full[blk]))
    assert 0 < worst < 1.5e-6, worst
    key = lambda p: (p["freq"], p["mode"], p["octets"], p["fcs_status"], p["lpdus"])
    a, b = sorted(pdus_full, key=lambda p: (p["freq"], p["sample_index"])), sorted(pdus_pruned, key=lambda p: (p["freq"], p["sample_index"]))
    assert len(a) >= 300 and [key(p) for p in a] == [key(p) for p in b]
    assert all(abs(p["sample_index"] - q["sample_index"]) <= 3 for p, q in zip(a, b))
    # against the oracle, like every other channelizer gate
    sub = [0, 127, 128, 255]
    ora = oracle.Frontend(w["fs"], w["centerfreq"], [freqs[c] for c in sub], nthreads=8)
    ora.push_block(x[:g.input_size], nthreads=8)
    for i, c in enumerate(sub):
        assert rel_rms(pruned[0][watch.index(c)], ora.channel_view(i)["chan_out"]) < RMS_TOL
    ora.close()
    # a left-over octet, small geometry
    fs, cf, nch = 2_400_000, 10_000_000, 132
    fr = [int(cf + (i - nch // 2) * 15_000 + 4_000) for i in range(nch)]
    rng = np.random.default_rng(5)
    bl = [dict(freq=fr[c], mode=int(rng.integers(0, 4)), octets=b"", t0=float(rng.uniform(0.1, 0.5)), amp=0.03, cfo=float(rng.uniform(-10, 10)))
          for c in (0, 64, 127, 128, 131)]
    for q in bl:
        q["octets"] = synth.make_pdu(rng, q["mode"])
    xs = synth.synth_wideband(fs, cf, int(3.4 * fs), bl, noise_sigma=0.012, seed=5)
    watch = [0, 64, 127, 128, 129, 131]
    full, pf, r0 = run(0, fs, cf, fr, xs, (0, 3))
    pruned, pp, r1 = run(3e-7, fs, cf, fr, xs, (0, 3))
    assert r0[0] == r0[1] and r1[0] < r1[1], (r0, r1)
    worst = max(rel_rms(a, b) for blk in full for a, b in zip(pruned[blk], full[blk]))
    assert 0 < worst < 1.5e-6, worst
    assert len(pf) == len(bl) and sorted(key(p) for p in pf) == sorted(key(p) for p in pp)


def test_two_rank_bench_on_one_gpu():
    """The `bench.py --gpus N` launch path exactly as the driver starts it (torch.distributed.run, one process per rank),
    with N = 2 on the single GPU of the test box: ranks take stream seeds 5 and 6 (BASELINE.json configs[4]), rank 0 prints
    ONE JSON line whose sample count is the sum over ranks and whose time is the max; every PDU of both ranks matches a
    payload its own stream carried."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2"]      # default backend nccl: two ranks on one GPU -> gloo, and the line says so
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 16 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["config"]["stream_seeds"] == [5, 6]
    block = r["config"]["block_samples"]
    assert r["config"]["channels"] == 256 and block == 7340032
    samples = 2 * 16 * block            # 16 timed steps = once around each rank's resident stretch: every channel's burst ends in it
    assert abs(r["value"] * 1e6 * (r["ms_per_step"] * 16e-3) - samples) < 1e-6 * samples
    assert r["pdus_in_timed_region"] > 0
    assert r["pdus_matching_sent_payload"] == r["pdus_in_timed_region"]
    # 16 timed blocks per rank, folded fold_batch at a time (one pass over the filter taps per launch)
    assert r["roofline"]["launches"] * r["roofline"]["blocks_per_launch"] == pytest.approx(16) and r["roofline"]["blocks_per_launch"] >= 2
    assert 0.05 < r["roofline"]["frac"] < 1.0
    assert "cpu_baseline" not in r                       # rank 0 at N = 1 only
    # every rank's own numbers, in rank order: a straggler shows here, not only in the max
    pr = r["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and [p["stream_seed"] for p in pr] == [5, 6] and all(p["channels"] == 256 for p in pr)
    assert max(p["ms_per_step"] for p in pr) == pytest.approx(r["ms_per_step"], rel=1e-3)
    assert sum(p["pdus"] for p in pr) == r["pdus_in_timed_region"] and all(p["fold_avg_ms"] > 0 and p["demod_ms_per_block"] > 0 for p in pr)
    assert r["distributed"]["requested"] == "nccl" and r["distributed"]["world_size"] == 2
    import torch
    if torch.cuda.device_count() >= 2:       # a multi-GPU box: the two ranks sit on two devices and talk through RCCL
        assert r["distributed"]["backend"] == "nccl" and r["distributed"]["fallback"] is None
    else:                                    # the 1-GPU test box: ranks share the device, RCCL is not asked (it would hang), the line says so
        assert r["distributed"]["backend"] == "gloo" and "share a device" in r["distributed"]["fallback"]


def test_rccl_path_at_world_size_one():
    """The backend the driver's 8-GPU run uses, executed here: `torch.distributed.run --nproc-per-node 1 bench.py --backend nccl`.  The
    process group is brought up with device_id (eager RCCL communicator), and the barrier, the max / sum reductions and the per-rank
    gather of the run all go through RCCL on CUDA tensors -- at world size 1, but through the same calls."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg2", "--steps", "26", "--warmup", "0",
           "--backend", "nccl", "--no-cpu-baseline", "--no-extra-legs"]            # 26 steps = once around the resident stretch: every burst ends in it
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # stdout = ONE JSON line: no RCCL banner, no gloo chatter
    r = json.loads(lines[0])
    d = r["distributed"]
    assert d == dict(backend="nccl", requested="nccl", world_size=1, fallback=None), d      # RCCL came up: no gloo fall-back
    assert r["n_gpus"] == 1 and r["steps"] == 26 and r["config"]["stream_seeds"] == [2]
    assert len(r["per_rank"]) == 1 and r["per_rank"][0]["rank"] == 0 and r["per_rank"][0]["pdus"] == r["pdus_in_timed_region"]
    assert r["per_rank"][0]["ms_per_step"] == pytest.approx(r["ms_per_step"], rel=1e-3)
    assert r["pdus_in_timed_region"] > 0 and r["pdus_matching_sent_payload"] == r["pdus_in_timed_region"]


def test_poll_sequence_drain_then_snapshot(gpu):
    """hfdl_gpu_frontend_poll_pdus (drain) followed by hfdl_gpu_frontend_poll_pdus_ready(.., 1) with no push in between: the
    snapshot is older than what was already taken -- nothing may be delivered twice and the ring must keep working."""
    fs, cf = 250000, 10_000_000
    freqs = [9_930_000, 10_037_000]
    bursts = synth.plan_traffic(freqs, 9.0, seed=31, dense=True)
    x = synth.synth_wideband(fs, cf, int(9.0 * fs), bursts, noise_sigma=0.01, seed=31)
    fe = gpu.Frontend(fs, cf, freqs)
    n = fe.input_size
    got = []
    for b in range(len(x) // n):
        fe.push_block(x[b * n:(b + 1) * n])
        got += fe.poll_pdus()                         # drain: taken == produced
        again = fe.poll_pdus(max_in_flight=1)         # stale snapshot (the block before): must yield nothing
        assert again == []
    cnt = fe.counters()
    assert cnt["pdus_dropped"] == 0 and cnt["pdus_taken"] == len(got) and len(got) >= len(bursts) - 1
    keys = [(p["freq"], p["sample_index"]) for p in got]
    assert len(set(keys)) == len(keys)
    # a NULL buffer with max > 0 is an argument error and discards nothing
    import ctypes as C
    L = F.load()
    k = C.c_int32(0)
    assert L.hfdl_gpu_frontend_poll_pdus(fe._h, None, 4, C.byref(k)) == -1
    assert L.hfdl_gpu_frontend_poll_pdus_ready(fe._h, None, 4, C.byref(k), 1) == -1
    fe.close()


def _run_bench(args, nproc, port, tmp_path, tag, backend="gloo"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dump = str(tmp_path / tag)
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--no-cpu-baseline", "--no-extra-legs", "--dump-pdus", dump] + args
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base + ["--backend", backend]
    else:
        cmd = [sys.executable] + base
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    keys = []
    for r in range(nproc):
        keys.append([tuple(k) for k in json.load(open("%s.rank%d.json" % (dump, r)))])
    return json.loads(line[0]), keys


def test_single_stream_channel_sharded_union_equals_unsharded(tmp_path):
    """SURVEY.md 8(e), one stream over G GPUs: `bench.py --shard channels` gives every rank the SAME wideband stream and a
    round-robin subset of the channels (here 2 ranks on the one GPU of the test box).  The union of the ranks' PDUs must be
    the unsharded run's PDU set -- same (freq, sample_index, mode, octets) -- with no channel decoded twice."""
    args = ["--workload", "cfg2", "--steps", "26", "--warmup", "0", "--shard", "channels"]
    whole, k1 = _run_bench(args, 1, 0, tmp_path, "whole")
    parts, k2 = _run_bench(args, 2, 29547, tmp_path, "parts")
    assert whole["config"]["channels"] == parts["config"]["channels"] == 32 and parts["config"]["channels_rank0"] == 16
    assert parts["scaling"] == "strong" and parts["config"]["stream_seeds"] == [2, 2]
    assert parts["steps"] == 26 and abs(parts["value"] * 1e6 * parts["ms_per_step"] * 26e-3 - 26 * 917504) < 30     # ONE stream's samples
    f0, f1 = {k[0] for k in k2[0]}, {k[0] for k in k2[1]}
    assert not f0 & f1                                            # channel partition
    assert sorted(k2[0] + k2[1]) == sorted(k1[0]) and len(k1[0]) >= 30
    assert parts["pdus_in_timed_region"] == whole["pdus_in_timed_region"] == len(k1[0])
    assert parts["pdus_matching_sent_payload"] == parts["pdus_in_timed_region"]


def test_eight_rank_rehearsal_independent_streams(tmp_path):
    """BASELINE.json configs[4]'s launch shape -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`, default backend, default
    sharding -- rehearsed on the one GPU of the test box with the small geometry (8 front ends fit beside each other): eight ranks,
    stream seeds 5..12 in rank order, eight per-rank rows, every rank's PDUs match what ITS stream carried, samples summed over the ranks,
    time = the slowest rank's; RCCL is not asked for ranks that share a device (it would hang) and the line says so."""
    import torch
    r, keys = _run_bench(["--workload", "cfg2", "--steps", "26", "--warmup", "0"], 8, 29553, tmp_path, "s8", backend="nccl")
    assert r["n_gpus"] == 8 and r["steps"] == 26 and r["scaling"] == "weak"
    assert r["config"]["stream_seeds"] == [5, 6, 7, 8, 9, 10, 11, 12] and r["config"]["shard"] == "streams"
    pr = r["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and [p["stream_seed"] for p in pr] == list(range(5, 13))
    assert all(p["channels"] == 32 and p["pdus"] > 0 and p["pdus"] == p["pdus_matching_sent_payload"] and p["fold_avg_ms"] > 0 for p in pr)
    assert sum(p["pdus"] for p in pr) == r["pdus_in_timed_region"] == sum(len(k) for k in keys)
    assert max(p["ms_per_step"] for p in pr) == pytest.approx(r["ms_per_step"], rel=1e-3)
    assert abs(r["value"] * 1e6 * r["ms_per_step"] * 26e-3 - 8 * 26 * 917504) < 240               # eight streams' samples over the slowest rank's time
    assert len({tuple(sorted(k)) for k in keys}) == 8                                             # eight different streams
    d = r["distributed"]
    assert d["requested"] == "nccl" and d["world_size"] == 8
    if torch.cuda.device_count() >= 8:
        assert d["backend"] == "nccl" and d["fallback"] is None
    else:
        assert d["backend"] == "gloo" and "share a device" in d["fallback"]
    assert all(p["frontend_create_s"] < 120 and p["input_synthesis_s"] < 120 for p in pr)


def test_eight_rank_rehearsal_at_full_size(tmp_path):
    """The real thing at the real size, on the one GPU of the test box: `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`
    with the DEFAULT workload -- BASELINE.json configs[4]: eight independent 40 Msps x 256-channel streams, seeds 5 .. 12 -- eight front
    ends of ~19 GiB each beside one another in 288 GB of HBM, eight filter-tap designs on cpus / 8 host threads each, eight 0.94 GB
    input syntheses at once.  Eight per-rank rows in rank order, every rank's PDUs are what ITS stream carried, no rank's set-up comes
    near HFDL_BENCH_SETUP_BUDGET_S; the wall time of the whole launch goes to gpurun_out/ for profiles/.  What the first run on an
    8-GPU node then adds is RCCL between devices (exercised at world size 1 above) and eight times the HBM."""
    import time
    import torch
    t0 = time.time()
    r, keys = _run_bench(["--steps", "16", "--warmup", "4"], 8, 29571, tmp_path, "full8", backend="nccl")
    wall = time.time() - t0
    assert r["n_gpus"] == 8 and r["steps"] == 16 and r["scaling"] == "weak"
    assert "configs[2]" in r["config"]["workload"] and r["config"]["channels"] == 256 and r["config"]["fft_size"] == 1 << 23
    assert r["config"]["stream_seeds"] == [5, 6, 7, 8, 9, 10, 11, 12] and r["config"]["shard"] == "streams"
    pr = r["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and [p["stream_seed"] for p in pr] == list(range(5, 13))
    assert all(p["channels"] == 256 and p["pdus"] > 0 and p["pdus"] == p["pdus_matching_sent_payload"] and p["fold_avg_ms"] > 0 for p in pr)
    assert sum(p["pdus"] for p in pr) == r["pdus_in_timed_region"] == sum(len(k) for k in keys)
    assert len({tuple(sorted(k)) for k in keys}) == 8                                             # eight different streams
    assert abs(r["value"] * 1e6 * r["ms_per_step"] * 16e-3 - 8 * 16 * 7340032) < 2000             # eight streams' samples over the slowest rank's time
    budget = float(os.environ.get("HFDL_BENCH_SETUP_BUDGET_S", "600"))
    worst = max(p["frontend_create_s"] + p["input_synthesis_s"] for p in pr)
    assert worst < 0.5 * budget, (worst, pr)
    d = r["distributed"]
    assert d["requested"] == "nccl" and d["world_size"] == 8
    if torch.cuda.device_count() >= 8:
        assert d["backend"] == "nccl" and d["fallback"] is None
    else:
        assert d["backend"] == "gloo" and "share a device" in d["fallback"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(what="torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --steps 16 --warmup 4 (cfg3 per rank) on ONE MI355X",
                   wall_s=round(wall, 1), value_Msamples_s=r["value"], ms_per_step=r["ms_per_step"], worst_rank_setup_s=round(worst, 1),
                   per_rank=[{k: p[k] for k in ("rank", "stream_seed", "frontend_create_s", "input_synthesis_s", "ms_per_step", "pdus")} for p in pr],
                   distributed=d), open(os.path.join(ROOT, "gpurun_out", "r05_eight_rank_cfg3.json"), "w"), indent=1)


def test_eight_rank_rehearsal_one_stream_channel_sharded(tmp_path):
    """SURVEY.md 8(e) at the node's full width: ONE stream, its 32 channels round-robin over eight ranks (4 each), every rank ingests the
    same blocks.  The union of the eight shards' PDUs is the unsharded run's set, no channel is decoded twice, the stream's samples
    count once."""
    args = ["--workload", "cfg2", "--steps", "26", "--warmup", "0", "--shard", "channels"]
    whole, k1 = _run_bench(args, 1, 0, tmp_path, "whole8")
    parts, k8 = _run_bench(args, 8, 29557, tmp_path, "parts8")
    assert parts["n_gpus"] == 8 and parts["scaling"] == "strong" and parts["config"]["stream_seeds"] == [2] * 8
    assert parts["config"]["channels"] == 32 and parts["config"]["channels_rank0"] == 4
    assert [p["rank"] for p in parts["per_rank"]] == list(range(8)) and all(p["channels"] == 4 for p in parts["per_rank"])
    assert abs(parts["value"] * 1e6 * parts["ms_per_step"] * 26e-3 - 26 * 917504) < 30              # ONE stream's samples
    fsets = [{k[0] for k in ks} for ks in k8]
    assert all(not (fsets[i] & fsets[j]) for i in range(8) for j in range(i))                      # channel partition
    assert sorted(k for ks in k8 for k in ks) == sorted(k1[0]) and len(k1[0]) >= 30
    assert parts["pdus_in_timed_region"] == whole["pdus_in_timed_region"] == len(k1[0])
    assert parts["pdus_matching_sent_payload"] == parts["pdus_in_timed_region"]


def test_eight_rank_rehearsal_full_size_one_stream_channel_sharded(tmp_path):
    """SURVEY.md 8(e) at the node's full width AND at the full size (round 6): the default workload -- ONE 40 Msps stream, 256 channels --
    with `--shard channels` over eight ranks on the one GPU of the test box: every rank ingests the same blocks (its own forward FFT of
    2^23 points) and folds / demodulates 32 of the channels.  The union of the eight shards' PDUs is the unsharded 256-channel run's set
    -- same (freq, sample_index, mode, octets) -- no channel is decoded twice, and the stream's samples count once.  With the weak-scaling
    rehearsal above, the first run on an 8-GPU node is then a measurement of both modes, not a debugging session."""
    args = ["--steps", "16", "--warmup", "0", "--shard", "channels"]
    whole, k1 = _run_bench(args, 1, 0, tmp_path, "whole256")
    parts, k8 = _run_bench(args, 8, 29573, tmp_path, "parts256")
    assert "configs[2]" in whole["config"]["workload"] and whole["config"]["channels"] == 256 and whole["config"]["fft_size"] == 1 << 23
    assert parts["n_gpus"] == 8 and parts["scaling"] == "strong" and len(set(parts["config"]["stream_seeds"])) == 1
    assert parts["config"]["channels"] == 256 and parts["config"]["channels_rank0"] == 32 and parts["config"]["fft_size"] == 1 << 23
    assert [p["rank"] for p in parts["per_rank"]] == list(range(8)) and all(p["channels"] == 32 for p in parts["per_rank"])
    assert abs(parts["value"] * 1e6 * parts["ms_per_step"] * 16e-3 - 16 * 7340032) < 250             # ONE stream's samples
    fsets = [{k[0] for k in ks} for ks in k8]
    assert all(not (fsets[i] & fsets[j]) for i in range(8) for j in range(i))                      # channel partition
    assert sorted(k for ks in k8 for k in ks) == sorted(k1[0]) and len(k1[0]) >= 200
    assert parts["pdus_in_timed_region"] == whole["pdus_in_timed_region"] == len(k1[0])
    assert parts["pdus_matching_sent_payload"] == parts["pdus_in_timed_region"]
