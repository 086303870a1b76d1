cd /root/repo
python - <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
for name in ("cfg2", "cfg3"):
    w = bench.WORKLOADS[name]
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    for fmt in ("CF32", "CS16"):
        r = bench.host_path_leg(w, x, bench.channel_plan(w), fmt)
        print(name, fmt, r.get("value"), r.get("pdus"), r.get("error"))
PY
timeout 600 python -m pytest tests -m gpu -q -k "host_c_program" 2>&1 | tail -3
