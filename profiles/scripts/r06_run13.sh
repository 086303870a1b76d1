#!/bin/bash
# round 6, call: the C host path after sizing the page-locked input ring from the geometry (14 blocks below 128 channels, 22 from there up)
OUT=/root/repo/gpurun_out/r6q
mkdir -p $OUT
cd /root/repo
(time python -m pytest tests -x -q -m gpu -k "host_c_program or cfg1 or thirty or shard or statsd") 2>&1 | tail -n 6
python - > $OUT/host_path.json 2> $OUT/host_path.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
out = {}
for name in ("cfg3", "cfg2"):
    w = bench.WORKLOADS[name]
    import dumphfdl_amd as hf
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    out[name] = {fmt: [bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for _ in range(2)] for fmt in ("CS16", "CF32")}
print(json.dumps(out))
PY
python - <<PY
import json
hp = json.load(open("$OUT/host_path.json"))
for wl in hp:
    for fmt in hp[wl]:
        print(wl, fmt, [round(r["value"]) for r in hp[wl][fmt]], [r.get("pipeline_drains") for r in hp[wl][fmt]])
PY
grep -v "amdgpu.ids\|UserWarning" $OUT/host_path.err | tail -n 3
