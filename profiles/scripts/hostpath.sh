# the C host path (hfdl_replay --bench) on cfg2 and cfg3, cf32 and cs16 files in page cache
cd /root/repo
python - <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
for name in sys.argv[1:] or ("cfg2", "cfg3"):
    w = bench.WORKLOADS[name]
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    for fmt in ("CF32", "CS16"):
        r = bench.host_path_leg(w, x, bench.channel_plan(w), fmt)
        print(name, fmt, round(r["value"], 1), r["pdus"], r["lpdu_walk_on_device"], r["file_loops"])
PY
