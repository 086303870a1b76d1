cd /root/repo
python - <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
w = bench.WORKLOADS["cfg2"]
g = hf.plan_geometry(1024, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
for rep in range(3):
    for fmt in ("CF32", "CS16"):
        r = bench.host_path_leg(w, x, bench.channel_plan(w), fmt)
        print("cfg2", fmt, r.get("value"), r.get("seconds"), r.get("thread_s"), r.get("error"))
PY
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2', round(r['value'],1), round(r['steady_state_ms_per_step'],4), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], 'host_ram', r['host_ram_input']['value'], 'host_path', r['host_path'].get('value'))"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
