#!/usr/bin/env python3
"""fold_traffic_<workload>.json from the rocprofv3 --pmc databases of profiles/pmc_passes.sh: HBM bytes per fold launch, with the
gfx950 FETCH_SIZE correction calibrated on the stream-read probe kernel of the same run (it reads a known number of bytes).

A fold launch serves NB queued blocks with one pass over the filter taps.  NB is a run-time argument of the kernel:
the shapes come from the launch order of the profiled command, given as the third argument ("8,16,16,4": the warm-up half closed by a
poll, two full halves, the ragged rest), matched with the fold dispatches in time order; the algorithmic bytes are
bench.alg_bytes_per_launch(g, NB).  The record carries a hash of dumphfdl_amd/csrc as it was when
the counters were collected: bench.py compares it with the tree it runs from (roofline.traffic_source.csrc_matches_head), and this
script REFUSES to stamp a commit whose csrc differs from the working tree's."""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl, commit, shapes_arg, dbs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
launch_nbs = [int(v) for v in shapes_arg.split(",")]        # blocks per fold launch of the profiled command, in launch order
csrc_now = bench.csrc_hash()
# the commit being stamped must be the code that ran: profiles/stamp.sh (run where git is, refuses a dirty csrc) left the commit and the
# hash of its device sources; the tree this script runs from must hash the same
try:
    stamp = json.load(open(os.path.join(ROOT, "profiles", "scripts", "stamp.json")))
except Exception:
    sys.exit("fold_traffic.py: no profiles/scripts/stamp.json -- run profiles/stamp.sh on a clean csrc before collecting")
if stamp.get("csrc_sha16") != csrc_now or (commit not in ("", "unknown") and commit != stamp.get("commit")):
    sys.exit("fold_traffic.py: refusing to stamp commit %s: stamped %s with csrc %s, running tree has %s" % (commit, stamp.get("commit"), stamp.get("csrc_sha16"), csrc_now))
commit = stamp["commit"]
FOLD = "fold_mfma"
per_dispatch = {}          # counter -> [(start, kernel, value)] of the fold launches, in launch order
probe = []
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for name, ctr, start, val in cur.execute("select kernel_name, counter_name, start, value from counters_collection where kernel_name like '%hfdl%' order by start"):
        short = name.split("(")[0].replace("void ", "")
        if FOLD in short:
            per_dispatch.setdefault(ctr, []).append((start, short, val))
        elif "stream_read_kernel" in short and ctr == "FETCH_SIZE":
            probe.append(val)
w = bench.WORKLOADS[wl]
import dumphfdl_amd as hf  # noqa: E402
g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
g.channels = w["nch"]
assert per_dispatch.get("FETCH_SIZE"), "no fold kernel in the FETCH_SIZE pass"
for ctr, rows in per_dispatch.items():
    if len(rows) != len(launch_nbs):
        sys.exit("fold_traffic.py: %s pass saw %d fold dispatches, the command makes %d launches (%s): one dispatch per launch expected "
                 "(channel counts that leave no partial workgroup)" % (ctr, len(rows), len(launch_nbs), shapes_arg))
# the kernel's template arguments name the FORM of a launch (<P, W, D, WIN, SMALL, CG>: SMALL = the four-column form, CG = 2 the
# thirty-two-column one): the forms of the dispatches must be those the assumed launch order implies
for ctr, rows in per_dispatch.items():
    for (start, kname, val), nb in zip(rows, launch_nbs):
        m = re.search(r"fold_mfma16_kernel<\d+, \d+, \d+, (?:false|true|0|1), (false|true|0|1), (\d+)>", kname)
        if not m:
            continue
        small, cg = m.group(1) in ("true", "1"), int(m.group(2))
        want = ("4" if nb <= 4 else "32" if nb > 16 else "16")
        have = ("4" if small else "32" if cg == 2 else "16")
        if want != have:
            sys.exit("fold_traffic.py: %s pass: a launch assumed to hold %d blocks ran the %s-column form (%s): the launch order %s is not the command's" % (ctr, nb, have, kname, shapes_arg))
probe_bytes = min(w["nch"] * 8 * g.fft_size // (4 << 20) * (4 << 20), 16 << 30)
corr = probe_bytes / (1024.0 * (sum(probe) / len(probe))) if probe else 2.0


def avg(ctr, nb):
    v = [r[2] for r, n in zip(per_dispatch.get(ctr, []), launch_nbs) if n == nb]
    return (sum(v) / len(v)) if v else 0.0


def shape(nb):
    fname = [r[1] for r, n in zip(per_dispatch["FETCH_SIZE"], launch_nbs) if n == nb][0]
    fetch_kb, write_kb, hit, miss = avg("FETCH_SIZE", nb), avg("WRITE_SIZE", nb), avg("TCC_HIT_sum", nb), avg("TCC_MISS_sum", nb)
    rd, wr = fetch_kb * 1024 * corr, write_kb * 1024
    alg = bench.alg_bytes_per_launch(g, nb)
    return {
        "kernel": "hfdl::" + fname.split("hfdl::")[-1], "blocks_per_launch": nb,
        "FETCH_SIZE_KB_raw_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
        "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None,
        "dispatches": launch_nbs.count(nb),
    }


# every launch shape of the run: the block count of a launch is a run-time argument of the kernel, so the shapes come from the
# command's launch order (warm-up poll, full halves, the ragged rest), not from the kernel name
shapes = {nb: shape(nb) for nb in sorted(set(launch_nbs))}
top = shapes[max(shapes)]            # the full batch: what the timed region of a long run consists of
out = dict(top)
out.update({
    "workload": "%s: %s" % (wl, w["name"]),
    "measured_at_commit": commit,
    "csrc_sha16": csrc_now,
    "command": "profiles/pmc_passes.sh %s (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --workload %s --steps 68 --warmup 8 "
               "--no-cpu-baseline --no-extra-legs: fold launches of %s blocks in that order; second pass --pmc WRITE_SIZE; third --pmc TCC_HIT_sum TCC_MISS_sum)" % (wl, wl, shapes_arg),
    "gfx950_fetch_correction": round(corr, 4),
    "correction_calibration": "same run: stream_read_kernel reads exactly %d bytes and reports FETCH_SIZE = %.1f KB (x %.3f); "
                              "WRITE_SIZE uncorrected (fft passes write 8 N bytes and report that)" % (probe_bytes, sum(probe) / max(len(probe), 1), corr),
    "per_shape": {str(nb): shapes[nb] for nb in sorted(shapes)},
})
print(json.dumps(out, indent=2))
