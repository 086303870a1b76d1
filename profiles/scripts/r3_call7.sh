#!/bin/bash
# whole GPU suite + default bench + cfg2 bench + the C host path on cfg2 / cfg3 (cf32, cs16) at the current commit
OUT=/root/repo/gpurun_out/r3g
mkdir -p $OUT
cd /root/repo
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench rc=$?"
timeout 600 python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; echo "bench cfg2 rc=$?"
python - > $OUT/host_path.json 2>> $OUT/bench_cfg2.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
import dumphfdl_amd as hf
out = {}
for name in ("cfg2", "cfg3"):
    w = bench.WORKLOADS[name]
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    out[name] = {fmt: bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for fmt in ("CF32", "CS16")}
print(json.dumps(out))
PY
python - <<'PY'
import json
for f in ("bench_cfg3", "bench_cfg2"):
    r = json.load(open("/root/repo/gpurun_out/r3g/%s.json" % f))
    print(f, "value %.0f host_ram %.0f ms %.4f steady %.4f frac %.4f demod/blk %.4f batch %s pdus %d/%d host_path %.0f parity %s low %s cpu %.1f" % (
        r["value"], r["value_host_ram"], r["ms_per_step"], r["steady_state_ms_per_step"], r["roofline"]["frac"], r["demod_kernel_ms_per_block"], r["demod_blocks_per_launch"],
        r["pdus_matching_sent_payload"], r["pdus_in_timed_region"], r["host_path"]["value"], r["parity"]["pdu_multisets_identical"],
        [(b["snr_db"], b["identical"], b["gpu_only"], b["recovered_sets_identical"]) for b in r["parity"]["low_snr_bins"]], r["cpu_baseline"]["value"]))
    if "cfg2" in r:
        c = r["cfg2"]; print("   cfg2 leg: value %.0f host_ram %.0f steady %.4f demod/blk %.4f pdus %d/%d" % (c["value"], c["value_host_ram"], c["steady_state_ms_per_step"], c["demod_kernel_ms_per_block"], c["pdus_matching_sent_payload"], c["pdus"]))
d = json.load(open("/root/repo/gpurun_out/r3g/host_path.json"))
for k, v in d.items():
    for fmt, r in v.items(): print("host path", k, fmt, r.get("value"), r.get("thread_s"), r.get("pipeline_drains"), r.get("error"))
PY
