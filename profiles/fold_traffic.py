#!/usr/bin/env python3
"""fold_traffic_<workload>.json from the rocprofv3 --pmc databases of profiles/pmc_passes.sh: HBM bytes per fold launch, with the
gfx950 FETCH_SIZE correction calibrated on the stream-read probe kernel of the same run (it reads a known number of bytes)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl, commit, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
vals = {}
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for name, ctr, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                         "where kernel_name like '%hfdl%' group by kernel_name, counter_name"):
        vals[(name.split("(")[0].replace("void ", ""), ctr)] = (n, avg)
w = bench.WORKLOADS[wl]
import dumphfdl_amd as hf  # noqa: E402
g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
nch = w["nch"]
alg = 8 * g.input_size + nch * 8 * g.fft_size + nch * 8 * (g.post_input_size // g.post_decimation)
fold = [k for k in vals if "fold_kernel<" in k[0] and k[1] == "FETCH_SIZE"]
assert fold, "no fold kernel in the FETCH_SIZE pass"
fname = fold[0][0]
probe_bytes = min(nch * 8 * g.fft_size // (4 << 20) * (4 << 20), 16 << 30)
probe = [v[1] for k, v in vals.items() if "stream_read_kernel" in k[0] and k[1] == "FETCH_SIZE"]
corr = probe_bytes / (1024.0 * (sum(probe) / len(probe))) if probe else 2.0
fetch_kb = vals[(fname, "FETCH_SIZE")][1]
write_kb = vals.get((fname, "WRITE_SIZE"), (0, 0.0))[1]
hit = vals.get((fname, "TCC_HIT_sum"), (0, 0.0))[1]
miss = vals.get((fname, "TCC_MISS_sum"), (0, 0.0))[1]
rd, wr = fetch_kb * 1024 * corr, write_kb * 1024
out = {
    "kernel": "hfdl::" + fname.split("hfdl::")[-1],
    "workload": "%s: %s" % (wl, w["name"]),
    "measured_at_commit": commit,
    "command": "profiles/pmc_passes.sh %s (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --workload %s --steps 8 --warmup 2 "
               "--no-cpu-baseline --no-extra-legs; second pass --pmc WRITE_SIZE; third --pmc TCC_HIT_sum TCC_MISS_sum)" % (wl, wl),
    "FETCH_SIZE_KB_raw_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "gfx950_fetch_correction": round(corr, 4),
    "correction_calibration": "same run: stream_read_kernel reads exactly %d bytes and reports FETCH_SIZE = %.1f KB (x %.3f); "
                              "WRITE_SIZE uncorrected (fft passes write 8 N bytes and report that)" % (probe_bytes, sum(probe) / max(len(probe), 1), corr),
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
    "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None,
    "dispatches": vals[(fname, "FETCH_SIZE")][0],
}
print(json.dumps(out, indent=2))
