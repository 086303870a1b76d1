#!/usr/bin/env python3
"""What separates the device demodulator from the oracle, shown instead of asserted.

Builds (dumphfdl_amd/csrc/build_strict.sh, test-only): `strict_F` = the demodulator as the one-lane serial loop of
tests/hostsim/serial_demod.h on the fixed-sequence elementary functions of tests/hostsim/shared_math.h, with the shipped pipeline's
fast forms switched back on per bit of F: 1 = dot products summed in the DPP scan's order, 2 = AGC on v_log / v_exp / v_rcp,
4 = carrier NCO on v_sin / v_cos, 8 = nearest-point slicer.  The oracle runs the same elementary functions (orc_variant.shared_math).

 Two feeds per build: "wide" = the wideband samples through the whole device path (the device's own channelizer in front of the
 demodulator), "base" = the ORACLE's channelizer output handed to the device's demodulator stage (hfdl_gpu_frontend_push_baseband):
 the two channelizers round differently (different FFT factorisations: ~1e-6 relative), and only "base" takes that out.

 1. strict_0, base, vs oracle: TAP_SYMBOLS of every block and channel compared as bit patterns, PDUs of every SNR bin compared whole.
 2. strict_1 / 2 / 4 / 8, base, vs oracle: frames per bin that differ -- which fast form costs what.
 3. strict_15 vs the shipped library: the serial loop with every fast form on reproduces the three-wave pipeline (PDU sets per bin).
 4. wide vs base for every build: what the channelizers' rounding alone does.

    python profiles/strict_study.py [--bins=-8:2:2] > gpurun_out/strict_study.json       (markdown summary on stderr)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import hfdl_synth as synth          # noqa: E402
import low_snr_parity as L          # noqa: E402

TMP = "/tmp/strict_study"
KEY = lambda p: (p["freq"], p["sample_index"], p["mode"], p["octets"].hex())
VARIANTS = [("shipped", None)] + [("strict_%d" % f, os.path.join(ROOT, "build", "strict", "libhfdl_gpu_strict_%d.so" % f)) for f in (0, 1, 2, 4, 8, 15)]
FORMS = {0: "none (serial loop, shared elementary functions)", 1: "dot products in the DPP scan's order", 2: "AGC on v_log / v_exp / v_rcp",
         4: "carrier NCO on v_sin / v_cos", 8: "nearest-point slicer", 15: "all four"}


def small_traffic():
    fs, cf = 250000, 10_000_000
    freqs = [9_915_000, 9_972_000, 10_026_000, 10_083_000, 10_101_000]
    bursts = synth.plan_traffic(freqs, 9.0, seed=41, dense=True, gap_s=0.12, amp=(0.004, 0.05))
    return fs, cf, freqs, synth.synth_wideband(fs, cf, int(9.0 * fs), bursts, noise_sigma=0.012, seed=41)


def child(out_path, bins):
    """GPU side of one library build (HFDL_GPU_LIB): PDU keys per bin and feed; for strict_0 also the symbol taps against the oracle, bit for bit."""
    import dumphfdl_amd as hf
    from dumphfdl_amd import frontend as F
    res = dict(lib=os.environ.get("HFDL_GPU_LIB", "shipped"), bins={})
    for s in bins:
        x = np.load("%s/bin_%d.npy" % (TMP, s))
        fe = hf.Frontend(L.FS, L.CF, L.FREQS)
        fe.enable_taps(False)
        n, pdus = fe.input_size, []
        for b in range(len(x) // n):
            fe.push_block(x[b * n:(b + 1) * n])
            if b % 8 == 7:
                pdus += fe.poll_pdus(max_in_flight=1)
        pdus += fe.poll_pdus()
        fe.close()
        z = np.load("%s/base_%d.npz" % (TMP, s))
        zo, zc = z["out"], z["cnt"]                  # (an NpzFile re-reads a member on every access)
        fe = hf.Frontend(L.FS, L.CF, L.FREQS)
        fe.enable_taps(False)
        pb = []
        for b in range(len(zc)):
            fe.push_baseband([zo[b, c, :zc[b, c]] for c in range(len(L.FREQS))])
            pb += fe.poll_pdus()
        fe.close()
        res["bins"][str(s)] = dict(wide=sorted(KEY(p) for p in pdus), base=sorted(KEY(p) for p in pb))
    if os.environ.get("STRICT_TAPS"):
        from oracle import pyoracle
        pyoracle.set_variant(shared_math=1)
        fs, cf, freqs, x = small_traffic()
        res["taps"] = {}
        for feed in ("base", "wide"):
            fe = hf.Frontend(fs, cf, freqs)
            ora = pyoracle.Frontend(fs, cf, freqs)
            n, nsym, nbad, pd, worst = fe.input_size, 0, 0, [], 0.0
            for b in range(len(x) // n):
                blk = x[b * n:(b + 1) * n]
                ora.push_block(blk)
                if feed == "wide":
                    fe.push_block(blk)
                else:
                    fe.push_baseband([ora.channel_view(c)["chan_out"] for c in range(len(freqs))])
                pd += fe.poll_pdus()
                for c in range(len(freqs)):
                    a = fe.read_tap(F.TAP_SYMBOLS, c)
                    w = ora.channel_view(c)["symbols"]
                    nsym += len(w)
                    if len(a) != len(w):
                        nbad += len(w)
                    elif len(w):
                        nbad += int(np.count_nonzero(a.view(np.uint32).reshape(-1, 2) != w.view(np.uint32).reshape(-1, 2)) > 0 and
                                    np.count_nonzero(np.any(a.view(np.uint32).reshape(-1, 2) != w.view(np.uint32).reshape(-1, 2), axis=1)))
                        worst = max(worst, float(np.max(np.abs(a - w))))
            res["taps"][feed] = dict(symbols_compared=nsym, symbols_with_other_bits=nbad, max_abs_difference=worst, blocks=len(x) // n, channels=len(freqs),
                                     pdus_gpu=len(pd), pdus_oracle=len(ora.pdus),
                                     pdus_identical=sorted(KEY(p) for p in pd) == sorted(KEY(p) for p in ora.pdus))
            fe.close(); ora.close()
        pyoracle.set_variant()
    json.dump(res, open(out_path, "w"))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", default="-8:2:2")
    ap.add_argument("--child", default=None)
    ap.add_argument("--bursts-per-channel", type=int, default=4)
    a = ap.parse_args()
    lo, hi, st = (int(v) for v in a.bins.split(":"))
    bins = list(range(lo, hi + 1, st))
    if a.child:
        return child(a.child, bins)
    out = run_study(bins, a.bursts_per_channel)
    print(json.dumps(out))
    print(markdown(out), file=sys.stderr)


def run_study(bins, bursts_per_channel=4, builds=None):
    """builds: names out of VARIANTS (default all).  Returns dict(bins, taps, rows, forms)."""
    class A:
        pass
    a = A()
    a.bursts_per_channel = bursts_per_channel
    a.bins = "%d:%d:%d" % (bins[0], bins[-1], (bins[1] - bins[0]) if len(bins) > 1 else 1)
    variants = [v for v in VARIANTS if builds is None or v[0] in builds]
    os.makedirs(TMP, exist_ok=True)
    from multiprocessing import get_context
    with get_context("spawn").Pool(max(1, min(len(bins), (os.cpu_count() or 2) // 2))) as pool:
        made = pool.map(L.synth_bin, [(s, a.bursts_per_channel, L.bin_seed(s)) for s in bins])
    sent = {}
    for s, (bursts, x) in zip(bins, made):
        np.save("%s/bin_%d.npy" % (TMP, s), x)
        sent[s] = bursts
    # the oracle, once per elementary-function set
    from oracle import pyoracle
    threads = max(1, min(os.cpu_count() or 1, 64))
    ora = {}
    for name, sm in (("libm", 0), ("shared_math", 1)):
        pyoracle.set_variant(shared_math=sm)
        ora[name] = {}
        for s, (_, x) in zip(bins, made):
            o = pyoracle.Frontend(L.FS, L.CF, L.FREQS, nthreads=threads)
            n = o.ddc.input_size
            nblk = len(x) // n
            row = o.ddc.post_input_size // o.ddc.post_decimation + 2
            outs, cnts = np.zeros((nblk, len(L.FREQS), row), np.complex64), np.zeros((nblk, len(L.FREQS)), np.int32)
            for b in range(nblk):
                o.push_block(x[b * n:(b + 1) * n], nthreads=threads)
                if sm:          # the channelizer output the device's demodulator stage is fed with (it does not depend on the function set)
                    for c in range(len(L.FREQS)):
                        v = o.channel_view(c)["chan_out"]
                        outs[b, c, :len(v)] = v
                        cnts[b, c] = len(v)
            if sm:
                np.savez("%s/base_%d.npz" % (TMP, s), out=outs, cnt=cnts)
            ora[name][s] = sorted(KEY(p) for p in o.pdus)
            o.close()
    pyoracle.set_variant()
    gpu = {}
    for name, lib in variants:
        env = dict(os.environ)
        if lib:
            if not os.path.exists(lib):
                print("missing %s (run dumphfdl_amd/csrc/build_strict.sh)" % lib, file=sys.stderr)
                continue
            env["HFDL_GPU_LIB"] = lib
        if name == "strict_0":
            env["STRICT_TAPS"] = "1"
        out = "%s/%s.json" % (TMP, name)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, "--bins=" + a.bins], env=env, capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            print("%s failed: %s" % (name, r.stderr[-800:]), file=sys.stderr)
            continue
        gpu[name] = json.load(open(out))

    def recovered(keys, s):
        by = {}
        for b in sent[s]:
            by.setdefault(b["freq"], []).append(b)
        return sum(1 for f, si, m, o in keys if any(bytes.fromhex(o)[:len(b["octets"])] == b["octets"] and m == b["mode"] for b in by.get(f, ())))

    def diff(a, b):
        sa, sb = set(map(tuple, a)), set(map(tuple, b))
        return dict(common=len(sa & sb), only_a=len(sa - sb), only_b=len(sb - sa))

    rows = []

    def row(build, feed, against, s, a, b):
        d = diff(a, b)
        rows.append(dict(build=build, feed=feed, against=against, snr_db=s, pdus=len(a), other_pdus=len(b), common=d["common"], only_here=d["only_a"],
                         only_there=d["only_b"], identical=d["only_a"] == 0 and d["only_b"] == 0, recovered=recovered(a, s), other_recovered=recovered(b, s)))

    for name, _ in variants:
        if name not in gpu:
            continue
        for feed in ("base", "wide"):
            for oname in ("shared_math", "libm"):
                if name != "shipped" and oname == "libm":
                    continue
                for s in bins:
                    row(name, feed, "oracle(%s)" % oname, s, gpu[name]["bins"][str(s)][feed], ora[oname][s])
    if "strict_15" in gpu and "shipped" in gpu:
        for feed in ("base", "wide"):
            for s in bins:
                row("strict_15", feed, "shipped", s, gpu["strict_15"]["bins"][str(s)][feed], gpu["shipped"]["bins"][str(s)][feed])
    for name, _ in variants:
        if name in gpu:
            for s in bins:
                row(name, "wide", "%s, base" % name, s, gpu[name]["bins"][str(s)]["wide"], gpu[name]["bins"][str(s)]["base"])
    for s in bins:
        row("oracle(libm)", "-", "oracle(shared_math)", s, ora["libm"][s], ora["shared_math"][s])
    return dict(bins=bins, bursts_per_bin=len(sent[bins[0]]), taps=gpu.get("strict_0", {}).get("taps"), rows=rows, forms={str(k): v for k, v in FORMS.items()})


def markdown(out):
    bins, rows, lines = out["bins"], out["rows"], []
    lines.append("| build | fast forms on | feed | against | " + " | ".join("%+d dB" % s for s in bins) + " |")
    lines.append("|---|---|---|---|" + "---|" * len(bins))
    seen = []
    for r in rows:
        k = (r["build"], r["feed"], r["against"])
        if k in seen:
            continue
        seen.append(k)
        cells = []
        for s in bins:
            q = [x for x in rows if (x["build"], x["feed"], x["against"]) == k and x["snr_db"] == s][0]
            cells.append("identical (%d)" % q["common"] if q["identical"] else "%d common, %d / %d differ" % (q["common"], q["only_here"], q["only_there"]))
        f = FORMS.get(int(r["build"].split("_")[1]), "") if r["build"].startswith("strict_") else ("the shipped three-wave pipeline" if r["build"] == "shipped" else "")
        lines.append("| %s | %s | %s | %s | %s |" % (r["build"], f, r["feed"], r["against"], " | ".join(cells)))
    if out["taps"]:
        lines.append("")
        lines.append("strict_0 symbol taps vs oracle(shared_math): %s" % json.dumps(out["taps"]))
    return "\n".join(lines)


if __name__ == "__main__":
    main()
