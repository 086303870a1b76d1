#!/usr/bin/env python3
"""Sensitivity of the decoded result to every UNPINNED reading of liquid-dsp / the transmitter (CPU only, oracle only).

Nothing in this image can compile liquid-dsp or src/hfdl.c, so the oracle's liquid objects are restatements ("parity unpinned").  This
script attacks the common-mode risk from the other side: it decodes the SAME traffic under every alternative reading the oracle
knows (oracle/hfdl_oracle.h orc_variant: version-dependent or recollected choices) and under alternative transmit pulses, and tabulates
what moves: the decoded PDU set, the payloads recovered, the preamble counters (A2_found / M1_not_found), the training-bit errors.
A reading whose alternative changes a decoded octet is one a maintainer must check against real liquid-dsp; one that changes nothing
on thousands of bursts down to -6 dB is not where a misreading could hide.

Traffic sets (all seeded):
  cfg3    bench.py's cfg3 traffic (40 Msps, 256 channels, one single-slot burst each, 19..29 dB) through the oracle's channelizer
  cfg4    bench.py's cfg4 traffic (burst-dense, all 8 modes), same route
  bb20    256 bursts at 19..29 dB synthesised directly at the post-channelizer rate (fs / 4096): the baseband twin of cfg3, cheap
          enough to re-synthesise under every transmit-pulse variant
  snr<k>  bins of 2 dB from -6 to +10 dB in-channel SNR, 208 bursts each (26 per mode), baseband

  python profiles/variant_study.py [--jobs N] [--sets cfg3,cfg4,bb20,snr] [--out profiles/r03_variant_sensitivity]
writes <out>.json and <out>.md.  The channelised sets are cached under /tmp (the channelizer does not depend on any variant).
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hfdl_synth as synth          # noqa: E402
import bench                        # noqa: E402

FS, CF, DECIM = 40_000_000, 15_000_000, 4096
RATE = FS / DECIM                                  # 9765.625 Hz: cfg3's post-channelizer rate

# name -> (orc_variant fields, transmit pulse or None, what the alternative stands for)
RX_VARIANTS = [
    ("default", {}, "the restatement every parity test uses and the GPU implements"),
    ("symsync_reset_both", dict(symsync_reset_both=1), "symsync_crcf_reset clears the derivative bank too (default: matched-filter bank only)"),
    ("symsync_bank_floor", dict(symsync_bank_floor=1), "symsync filter-bank index floorf(bf) instead of roundf(bf)"),
    ("resamp_float_64", dict(resamp_kind=1), "arbitrary resampler of liquid <= 1.3.1: float phase, linear interpolation, 64 branches, fc 0.4"),
    ("resamp_float_256", dict(resamp_kind=2), "float phase + interpolation with the 1.3.2 filter (256 branches)"),
    ("resamp_fixed_64", dict(resamp_kind=3), "fixed-point phase, 64 branches"),
    ("kaiser_arg_n_minus_1", dict(kaiser_arg=1), "Kaiser window argument 2t/(N-1) instead of 2t/N in every liquid filter design"),
    ("design_float", dict(design_float=1), "filters designed in single precision with liquid's own series (besseli0f, sincf)"),
    ("soft_dmin_1", dict(soft_dmin_init=1.0), "8-PSK soft de-mapper: 'no neighbour' distance 1.0 instead of 4.0"),
    ("soft_dmin_16", dict(soft_dmin_init=16.0), "the same, 16.0 (any value >= ~1.7 saturates the soft bit of a neighbourless decision)"),
    ("eqlms_norm_exact", dict(eqlms_norm=1), "equaliser step normalised by a freshly summed |x|^2 instead of the running sum"),
    ("eqlms_norm_none", dict(eqlms_norm=2), "equaliser step not normalised (liquid < 1.3)"),
    ("agc_double", dict(agc_double=1), "AGC energy recursion evaluated in double (liquid's 1.0 literal)"),
    ("perr_angle", dict(perr_kind=1), "demodulator phase error = angle(r conj(x_hat)) instead of Im(r conj(x_hat))"),
    ("dot_even_odd", dict(dot_order=1), "dot products summed as even / odd partial sums (SIMD dotprod) instead of sequentially"),
    ("symsync_dmf_x0.5", dict(symsync_dmf_scale=0.5), "symsync derivative filter normalised to 0.03 / max|h dh| instead of 0.06 (timing-error gain halved)"),
    ("symsync_dmf_x2", dict(symsync_dmf_scale=2.0), "the same, 0.12 (gain doubled)"),
    ("symsync_lf_b_0.5", dict(symsync_lf_b=0.5), "symsync loop filter feed-back coefficient 0.5 instead of 0.495"),
    ("soft_gamma_x0.83", dict(soft_gamma_scale=1.0 / 1.2), "8-PSK soft de-mapper gamma = M instead of 1.2 M"),
    ("soft_floor", dict(soft_floor=1), "soft bit = floor(llr * 16 + 127) instead of the C cast's truncation"),
    ("agc_y2_init_0.01", dict(agc_y2_init=0.01), "AGC energy estimate starts at 0.01 instead of 1.0"),
    ("lfsr_old_api", dict(lfsr_kind=1), "scrambler through the pre-1.6 msequence API restated literally (src/hfdl.c:331-333): must equal the default"),
    ("lfsr_right_shift", dict(lfsr_kind=2), "scrambler as a right-shifting register (the other reading of the >= 1.6 API)"),
]
TX_VARIANTS = [
    ("tx_rrc_0.20", ("rrc", 0.2), "default transmitter: root-raised cosine, roll-off 0.2"),
    ("tx_rrc_0.165", ("rrc", 0.165), "roll-off of the RRC that fits the receiver's 19-tap table best"),
    ("tx_rrc_0.35", ("rrc", 0.35), "a wider roll-off"),
    ("tx_mf_table", ("mf_table", 0.0), "the receiver's own matched-filter table as the transmit pulse (band-limited interpolation)"),
]


def snr_to_sigma(amp, snr_db):
    """in-channel SNR = signal power / noise power over the whole channel (complex noise, 2 sigma^2) at the post-channelizer rate"""
    return amp / (10 ** (snr_db / 20.0)) / np.sqrt(2.0)


# ---------------------------------------------------------------- traffic

def plan_baseband(name, seed):
    """streams: list of (stream id, bursts, noise sigma, n samples)"""
    rng = np.random.default_rng(seed)
    streams = []
    if name == "bb20":
        # the bench's traffic parameters (bench.plan_bursts): amplitude 0.01..0.03 against in-channel noise rms 0.05 * sqrt(2 / 4096)
        sigma = 0.05 / np.sqrt(DECIM)
        for i in range(256):
            mode = i % 4
            octets, _ = bench.make_payload(rng, mode)
            b = dict(mode=mode, octets=octets, t0=float(rng.uniform(0.1, 0.6)), amp=float(rng.uniform(0.01, 0.03)), cfo=float(rng.uniform(-15, 15)))
            n = int((b["t0"] + synth.burst_symbols_len(mode) / 1800 + 0.25) * RATE)
            streams.append((i, [b], sigma, n))
        return streams
    snr = float(name[3:])
    for s in range(26):                       # 26 streams x 8 bursts (one per mode, shuffled) = 208 bursts per bin
        order = rng.permutation(8)
        t, bl = float(rng.uniform(0.1, 0.5)), []
        for m in order:
            m = int(m)
            bl.append(dict(mode=m, octets=synth.make_pdu(rng, m), t0=t, amp=0.02, cfo=float(rng.uniform(-20, 20)),
                           snr=snr + float(rng.uniform(-1, 1))))
            t += synth.burst_symbols_len(m) / 1800 + float(rng.uniform(0.25, 0.5))
        streams.append((s, bl, snr_to_sigma(0.02, snr), int((t + 0.2) * RATE)))
    return streams


def synth_stream(args):
    sid, bursts, sigma, n, seed, pulse = args
    synth.set_tx_pulse(*pulse)
    bl = []
    for b in bursts:
        b = dict(b)
        if "snr" in b:                         # per-burst SNR inside the bin: scale the amplitude, the noise is the stream's
            b["amp"] = sigma * np.sqrt(2.0) * 10 ** (b["snr"] / 20.0)
        bl.append(b)
    return sid, synth.synth_channel_baseband(RATE, n, bl, noise_sigma=sigma, seed=seed)


def baseband_set(name, pulse, pool):
    """{stream id: samples}, the sent bursts per stream; cached per (set, pulse)"""
    seed = {"bb20": 20}.get(name, 100 + int(float(name[3:]) if name.startswith("snr") else 0))
    streams = plan_baseband(name, seed)
    cache = "/tmp/hfdl_variant_%s_%s_%s.npz" % (name, pulse[0], pulse[1])
    sent = {sid: bl for sid, bl, _, _ in streams}
    if os.path.exists(cache):
        z = np.load(cache)
        return {int(k): z[k] for k in z.files}, sent
    out = dict(pool.map(synth_stream, [(sid, bl, sig, n, 7919 * seed + sid, pulse) for sid, bl, sig, n in streams]))
    np.savez(cache, **{str(k): v for k, v in out.items()})
    return out, sent


def channelised_set(name, jobs):
    """bench.py's wideband traffic through the oracle's forward FFT + per-channel fold / inverse FFT / NCO: {channel: baseband}."""
    from oracle import pyoracle
    w = bench.WORKLOADS[name]
    freqs = bench.channel_plan(w)
    dec, tbw, ddc = pyoracle.geometry(w["fs"])
    cache = "/tmp/hfdl_variant_%s_chan_out.npz" % name
    nsamp = w["blocks"] * ddc.input_size
    bursts = bench.plan_bursts(w, freqs, nsamp / w["fs"], w["seed"])
    sent = {}
    for b in bursts:
        sent.setdefault(freqs.index(b["freq"]), []).append(b)
    if os.path.exists(cache):
        z = np.load(cache)
        return {int(k): z[k] for k in z.files}, sent
    t0 = time.time()
    x, _ = bench.make_input(w, ddc.input_size, 0, 1)
    fe = pyoracle.Frontend(w["fs"], w["centerfreq"], freqs, nthreads=jobs)
    pieces = {c: [] for c in range(len(freqs))}
    for k in range(w["blocks"]):
        fe.push_block(x[k * ddc.input_size:(k + 1) * ddc.input_size], nthreads=jobs)
        for c in pieces:
            pieces[c].append(fe.channel_view(c)["chan_out"])
        print("  %s: block %d / %d channelised (%.0f s)" % (name, k + 1, w["blocks"], time.time() - t0), file=sys.stderr)
    fe.close()
    out = {c: np.concatenate(v) for c, v in pieces.items()}
    np.savez(cache, **{str(k): v for k, v in out.items()})
    return out, sent


# ---------------------------------------------------------------- decoding under a variant

def decode_streams(args):
    """One process = one variant: the switch is process-global and read when a channel is created."""
    fields, streams = args
    from oracle import pyoracle
    pyoracle.set_variant(**fields)
    pdus, summ = [], dict(a1_found=0, a2_found=0, m1_found=0, m1_not_found=0, train_bits_bad=0, train_bits_total=0)
    per_stream = {}
    for sid, x in streams:
        ch = pyoracle.Channel(FS, CF, CF, want_channelizer=False)
        n = 1792                                     # the channelizer's block size: state is carried across process calls as in the path
        for k in range(0, len(x), n):
            ch.process_baseband(x[k:k + n])
        for p in ch.pdus:
            pdus.append((sid, int(p["sample_index"]), int(p["mode"]), p["octets"].hex(), int(p["train_bits_bad"]), int(p["train_bits_total"])))
        s = ch.summary()
        per_stream[sid] = s
        for k in summ:
            summ[k] += s[k]
        ch.close()
    pyoracle.set_variant()
    return pdus, summ, per_stream


def score(pdus, sent):
    """payloads recovered: PDUs whose octets start with a payload sent on that stream in that mode"""
    ok = 0
    for sid, _, mode, hexo, _, _ in pdus:
        o = bytes.fromhex(hexo)
        if any(o[:len(b["octets"])] == b["octets"] and mode == b["mode"] for b in sent.get(sid, ())):
            ok += 1
    return ok


def run_variant(name, fields, data, sent, jobs, pool):
    items = sorted(data.items())
    chunks = [items[i::jobs] for i in range(jobs)]
    res = pool.map(decode_streams, [(fields, c) for c in chunks if c])
    pdus = sorted(p for r in res for p in r[0])
    summ = {k: sum(r[1][k] for r in res) for k in res[0][1]}
    per_stream = {}
    for r in res:
        per_stream.update(r[2])
    return dict(pdus=pdus, summary=summ, recovered=score(pdus, sent), per_stream=per_stream)


def compare(base, var):
    key = lambda p: p[:4]
    a, b = {key(p) for p in base["pdus"]}, {key(p) for p in var["pdus"]}
    # same stream, same mode, detection within 3 samples, different octets = "a decoded octet changed"
    changed = 0
    bi = {}
    for p in base["pdus"]:
        bi.setdefault((p[0], p[2]), []).append(p)
    for p in var["pdus"]:
        for q in bi.get((p[0], p[2]), ()):
            if abs(q[1] - p[1]) <= 3 and q[3] != p[3]:
                changed += 1
                break
    moved = sum(1 for p in var["pdus"] if key(p) not in a and any(abs(q[1] - p[1]) <= 3 and q[3] == p[3] for q in bi.get((p[0], p[2]), ())))
    return dict(identical=a == b, only_default=len(a - b), only_variant=len(b - a), octets_changed=changed, same_octets_other_position=moved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--sets", default="cfg3,cfg4,bb20,snr")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_variant_sensitivity"))
    args = ap.parse_args()
    sets = []
    for s in args.sets.split(","):
        sets += ["snr%+d" % k for k in range(-6, 11, 2)] if s == "snr" else [s]
    pool = Pool(args.jobs, maxtasksperchild=1)          # fresh processes: a variant never leaks into the next task
    report = dict(rate_hz=RATE, sets={}, rx_variants={n: d for n, _, d in RX_VARIANTS}, tx_variants={n: d for n, _, d in TX_VARIANTS},
                  snr_definition="in-channel SNR: signal power over the noise power of the whole post-channelizer band (fs/4096 = 9765.625 Hz)")
    for sname in sets:
        t0 = time.time()
        if sname in ("cfg3", "cfg4"):
            data, sent = channelised_set(sname, args.jobs)
        else:
            data, sent = baseband_set(sname, TX_VARIANTS[0][1], pool)
        nb = sum(len(v) for v in sent.values())
        rows = {}
        base = run_variant("default", {}, data, sent, args.jobs, pool)
        for name, fields, _ in RX_VARIANTS:
            r = base if name == "default" else run_variant(name, fields, data, sent, args.jobs, pool)
            rows[name] = dict(pdus=len(r["pdus"]), recovered=r["recovered"], **r["summary"], vs_default=compare(base, r))
        if sname not in ("cfg3", "cfg4"):
            for name, pulse, _ in TX_VARIANTS[1:]:
                d2, _ = baseband_set(sname, pulse, pool)
                r = run_variant(name, {}, d2, sent, args.jobs, pool)
                rows[name] = dict(pdus=len(r["pdus"]), recovered=r["recovered"], **r["summary"], vs_default=None)
        # which bursts the default loses with A2 found and M1 not, against the burst's symbol-timing phase at the receiver's 5400 Hz grid
        lost = []
        if sname in ("bb20", "cfg3"):
            for sid, bl in sent.items():
                s = base["per_stream"].get(sid)
                if s and s["m1_not_found"] > 0:
                    lost.append(dict(stream=sid, t0_symbol_phase=round((bl[0]["t0"] * 1800.0) % 1.0, 3), mode=bl[0]["mode"], amp=bl[0].get("amp")))
        report["sets"][sname] = dict(bursts=nb, rows=rows, default_m1_not_found_bursts=lost, seconds=round(time.time() - t0, 1))
        print("%s: %d bursts, %d variants, %.0f s" % (sname, nb, len(rows), time.time() - t0), file=sys.stderr)
    pool.close()
    json.dump(report, open(args.out + ".json", "w"), indent=1)
    write_markdown(report, args.out + ".md")


def write_markdown(rep, path):
    L = []
    L.append("# r03: what the decoded result depends on -- every unpinned reading, one at a time (`profiles/variant_study.py`)")
    L.append("")
    L.append("The oracle's liquid-dsp objects are restatements (liquid is not in this image, `src/hfdl.c` cannot be compiled here).  Each row decodes the SAME "
             "traffic with ONE reading replaced by its alternative (`oracle/hfdl_oracle.h` `orc_variant`); `tx_*` rows change the synthetic transmitter instead.  "
             "`= default` compares the (stream, sample_index, mode, octets) sets; `octets changed` counts frames found at the same place (+-3 samples) whose "
             "octets differ -- the readings that matter.  SNR = %s." % rep["snr_definition"])
    L.append("")
    L.append("| variant | what is replaced |")
    L.append("|---|---|")
    for n, d in list(rep["rx_variants"].items()) + list(rep["tx_variants"].items()):
        L.append("| `%s` | %s |" % (n, d))
    names = list(rep["sets"].keys())
    L.append("")
    L.append("## Summary 1 -- frames whose OCTETS change against the default reading (same place +-3 samples, other octets); `=`: the PDU set is identical, `p`: same octets, only the detection sample moves")
    L.append("")
    L.append("| variant | " + " | ".join(names) + " |")
    L.append("|---|" + "---|" * len(names))
    for n in rep["rx_variants"]:
        if n == "default":
            continue
        cells = []
        for sn in names:
            v = rep["sets"][sn]["rows"].get(n, {}).get("vs_default")
            cells.append("-" if v is None else ("=" if v["identical"] else (str(v["octets_changed"]) if v["octets_changed"] else "p")))
        L.append("| `%s` | %s |" % (n, " | ".join(cells)))
    L.append("")
    L.append("## Summary 2 -- sent payloads recovered (of the bursts in the set)")
    L.append("")
    L.append("| variant | " + " | ".join("%s (%d)" % (sn, rep["sets"][sn]["bursts"]) for sn in names) + " |")
    L.append("|---|" + "---|" * len(names))
    for n in list(rep["rx_variants"]) + list(rep["tx_variants"])[1:]:
        L.append("| `%s` | %s |" % (n, " | ".join(str(rep["sets"][sn]["rows"][n]["recovered"]) if n in rep["sets"][sn]["rows"] else "-" for sn in names)))
    L.append("")
    L.append("## Summary 3 -- `M1_not_found` (A2 found, M1 search failed: the burst is lost) per set")
    L.append("")
    L.append("| variant | " + " | ".join(names) + " |")
    L.append("|---|" + "---|" * len(names))
    for n in list(rep["rx_variants"]) + list(rep["tx_variants"])[1:]:
        L.append("| `%s` | %s |" % (n, " | ".join(str(rep["sets"][sn]["rows"][n]["m1_not_found"]) if n in rep["sets"][sn]["rows"] else "-" for sn in names)))
    for sname, s in rep["sets"].items():
        L.append("")
        L.append("## %s -- %d bursts" % (sname, s["bursts"]))
        L.append("")
        L.append("| variant | PDUs | sent payloads recovered | = default set | only default / only variant | octets changed | same octets, other position | A2_found | M1_found | M1_not_found | training bits bad / total |")
        L.append("|---|---|---|---|---|---|---|---|---|---|---|")
        for name, r in s["rows"].items():
            v = r["vs_default"]
            cmp_ = ("yes" if v["identical"] else "NO", "%d / %d" % (v["only_default"], v["only_variant"]), str(v["octets_changed"]), str(v["same_octets_other_position"])) if v else ("-", "-", "-", "-")
            L.append("| `%s` | %d | %d | %s | %s | %s | %s | %d | %d | %d | %d / %d (%.3f %%) |" % (
                name, r["pdus"], r["recovered"], cmp_[0], cmp_[1], cmp_[2], cmp_[3], r["a2_found"], r["m1_found"], r["m1_not_found"],
                r["train_bits_bad"], r["train_bits_total"], 100.0 * r["train_bits_bad"] / max(1, r["train_bits_total"])))
        if s["default_m1_not_found_bursts"]:
            L.append("")
            L.append("Bursts the default reading loses with `A2_found` + `M1_not_found` (symbol-timing phase of the burst = start time x 1800 mod 1): " +
                     ", ".join("stream %d phase %.3f" % (b["stream"], b["t0_symbol_phase"]) for b in s["default_m1_not_found_bursts"]))
    open(path, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--merge":          # --merge OUT.json PART.json ...: sets of the parts replace those of OUT, the table is re-rendered
        rep = json.load(open(sys.argv[2]))
        for part in sys.argv[3:]:
            rep["sets"].update(json.load(open(part))["sets"])
        order = ["cfg3", "cfg4", "bb20"] + ["snr%+d" % k for k in range(-6, 11, 2)]
        rep["sets"] = {k: rep["sets"][k] for k in order if k in rep["sets"]}
        json.dump(rep, open(sys.argv[2], "w"), indent=1)
        write_markdown(rep, sys.argv[2][:-5] + ".md")
    else:
        main()
