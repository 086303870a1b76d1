#!/usr/bin/env python3
"""fold_traffic_<workload>.json from the rocprofv3 --pmc databases of profiles/pmc_passes.sh: HBM bytes per fold launch, with the
gfx950 FETCH_SIZE correction calibrated on the stream-read probe kernel of the same run (it reads a known number of bytes).

A fold launch serves NB queued blocks with one pass over the filter taps: NB is read from the kernel's template
arguments, the algorithmic bytes are bench.alg_bytes_per_launch(g, NB).  The record carries a hash of dumphfdl_amd/csrc as it was when
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

wl, commit, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
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
vals = {}
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for name, ctr, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                         "where kernel_name like '%hfdl%' group by kernel_name, counter_name"):
        vals[(name.split("(")[0].replace("void ", ""), ctr)] = (n, avg)
w = bench.WORKLOADS[wl]
import dumphfdl_amd as hf  # noqa: E402
g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
g.channels = w["nch"]


def blocks_of(kernel):
    """NB from the template arguments: fold_kernel<U, R, CS, NC, NB, WV>, fold_kernel_lds<U, R, NC, NB, WPW>"""
    a = [x.strip() for x in re.search(r"<(.*)>", kernel).group(1).split(",")]
    return int(a[3]) if "fold_kernel_lds" in kernel else int(a[4])


fold = [(k, v) for k, v in vals.items() if re.search(r"fold_kernel(_lds)?<", k[0]) and k[1] == "FETCH_SIZE"]
assert fold, "no fold kernel in the FETCH_SIZE pass"
probe_bytes = min(w["nch"] * 8 * g.fft_size // (4 << 20) * (4 << 20), 16 << 30)
probe = [v[1] for k, v in vals.items() if "stream_read_kernel" in k[0] and k[1] == "FETCH_SIZE"]
corr = probe_bytes / (1024.0 * (sum(probe) / len(probe))) if probe else 2.0


def shape(fname):
    nb = blocks_of(fname)
    fetch_kb = vals[(fname, "FETCH_SIZE")][1]
    write_kb = vals.get((fname, "WRITE_SIZE"), (0, 0.0))[1]
    hit = vals.get((fname, "TCC_HIT_sum"), (0, 0.0))[1]
    miss = vals.get((fname, "TCC_MISS_sum"), (0, 0.0))[1]
    rd, wr = fetch_kb * 1024 * corr, write_kb * 1024
    alg = bench.alg_bytes_per_launch(g, nb)
    return nb, {
        "kernel": "hfdl::" + fname.split("hfdl::")[-1], "blocks_per_launch": nb,
        "FETCH_SIZE_KB_raw_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
        "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None,
        "dispatches": vals[(fname, "FETCH_SIZE")][0],
    }


# every launch shape of the run (the pass runs 8 + 4 x 8 + 7 blocks: launches of 8, 4, 2 and 1 blocks); single-channel remainder
# launches (odd channel counts) are not in these workloads
shapes = {}
for (fname, _), _v in fold:
    nb, rec = shape(fname)
    if nb not in shapes or rec["dispatches"] > shapes[nb]["dispatches"]:
        shapes[nb] = rec
top = shapes[max(shapes)]            # the full batch: what the timed region of a long run consists of
out = dict(top)
out.update({
    "workload": "%s: %s" % (wl, w["name"]),
    "measured_at_commit": commit,
    "csrc_sha16": csrc_now,
    "command": "profiles/pmc_passes.sh %s (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --workload %s --steps 39 --warmup 8 "
               "--no-cpu-baseline --no-extra-legs; second pass --pmc WRITE_SIZE; third --pmc TCC_HIT_sum TCC_MISS_sum)" % (wl, wl),
    "gfx950_fetch_correction": round(corr, 4),
    "correction_calibration": "same run: stream_read_kernel reads exactly %d bytes and reports FETCH_SIZE = %.1f KB (x %.3f); "
                              "WRITE_SIZE uncorrected (fft passes write 8 N bytes and report that)" % (probe_bytes, sum(probe) / max(len(probe), 1), corr),
    "per_shape": {str(nb): shapes[nb] for nb in sorted(shapes)},
})
print(json.dumps(out, indent=2))
