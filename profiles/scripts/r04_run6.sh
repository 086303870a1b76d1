#!/bin/bash
OUT=/root/repo/gpurun_out/r4m
mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench_driver_line.err; echo "rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_driver_line.json"))
r = d["roofline"]
print("value %.0f ms/step %.4f steady %s fold %.3f ms x %.1f blk frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], r["traffic"]))
print("value_host_ram", d.get("value_host_ram"), "host_path", {k: d["host_path"].get(k) for k in ("msamples_per_s", "error", "pdus")} if "host_path" in d else None)
print("cfg2", {k: d["cfg2"].get(k) for k in ("value", "value_host_ram", "demod_kernel_ms_per_block", "whole_step_frac_of_hbm_peak", "pdus", "pdus_matching_sent_payload", "error")})
print("parity", {k: d["parity"].get(k) for k in ("chan_out_rel_rms", "pdu_multisets_identical", "gpu_pdus", "cpu_pdus")}, [ (b["snr_db"], b["identical"], b["gpu_only"], b["oracle_only"]) for b in d["parity"]["low_snr_bins"]] if isinstance(d["parity"].get("low_snr_bins"), list) else d["parity"].get("low_snr_bins"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("pdus", d["pdus_in_timed_region"], d["pdus_matching_sent_payload"], d["pdus_lpdu_walk_matching_sent"])
print("traffic_source", r["traffic_source"])
PY
tail -5 $OUT/bench_driver_line.err
