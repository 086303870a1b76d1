#!/usr/bin/env python3
"""The key figures of the artefacts under profiles/<round>_final_* on one screen (what DESIGN.md section 6 and README.md quote)."""
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), R + "_final_")


def L(f):
    try:
        return json.load(open(f))
    except Exception:
        return None


for name in ("bench_cfg3", "bench_cfg3_driver_line", "bench_cfg4", "bench_cfg2", "bench_cfg1", "bench_cfg3_host_cs16", "bench_cfg3_host_cf32", "bench_cfg3_fold_batch_1",
             "bench_cfg3_fold_batch_8", "bench_cfg3_rccl_world1", "bench_cfg3_under_rocprof"):
    d = L(P + name + ".json")
    if not d:
        print(name, "MISSING")
        continue
    r = d["roofline"]
    s = d.get("streams", {})
    print(name, "value %.0f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "steady %.4f" % (d.get("steady_state_ms_per_step") or 0), "fill %.2f" % (d.get("fill_drain_ms") or 0), r["bound"],
          "frac %.3f" % (r["frac"] or 0), {k: round(v["avg_ms"], 3) for k, v in r["launch_shapes"].items()}, "traffic", (r["traffic_source"] or {}).get("traffic_over_algorithmic"),
          "hbm %.3f" % (r["hbm"]["frac"] or 0), "A %.2f B %.2f D %.2f" % (s.get("stream_a_ms", 0), s.get("stream_b_ms", 0), s.get("stream_d_ms", 0)), "host_ram", d.get("value_host_ram"),
          "pdus", d["pdus_in_timed_region"], d["pdus_matching_sent_payload"])
d = L(P + "bench_cfg3.json")
print("per_block", d["streams"]["per_block_ms"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("parity", d["parity"]["chan_out_rel_rms"], d["parity"]["pdu_multisets_identical"], d["parity"]["gpu_pdus"])
pf = d["pruned_fold"]
print("pruned", {k: v for k, v in pf.items() if k not in ("what", "streams")})
print("pruned A/B", pf["streams"]["stream_a_ms"], pf["streams"]["stream_b_ms"])
print("host_path", d["host_path"].get("value"), "cfg2", d["cfg2"].get("value"), d["cfg2"].get("value_host_ram"), d["cfg2"].get("demod_kernel_ms_per_block"))
print("hbm", d["roofline"]["hbm"]["stream_read_GBs"], d["roofline"]["hbm"]["achieved"], "TF", d["roofline"]["achieved"])
dl = L(P + "bench_cfg3_driver_line.json")
print("driver pruned", dl["pruned_fold"]["value"], dl["pruned_fold"]["pdus_same_as_full_fold"])
hp = L(P + "host_path.json")
for k, v in hp.items():
    for fmt, runs in v.items():
        print(k, fmt, [round(r["value"]) for r in runs], [r.get("pipeline_drains") for r in runs])
