#!/bin/bash
# round 5, sixth measurement: pass-3 look-ahead, demodulator wave priority, non-blocking collection in the C host, whole GPU suite
OUT=/root/repo/gpurun_out/r5f
mkdir -p $OUT
cd /root/repo
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s fold_avg %.3f nb %.1f frac %.3f pdus %d/%d demod/blk %s host_ram %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"], d.get("value_host_ram")))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_256.json 2> $OUT/bench_cfg3_256.err
summ $OUT/bench_cfg3_256.json "cfg3 256 steps"
HFDL_GPU_DEMOD_BATCH=1 timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_256_db1.json 2> $OUT/bench_cfg3_256_db1.err
summ $OUT/bench_cfg3_256_db1.json "cfg3 256 steps demod_batch=1"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_20_legs.json 2> $OUT/bench_cfg3_20_legs.err
summ $OUT/bench_cfg3_20_legs.json "cfg3 driver-line 20 steps, extra legs"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_cfg3_20_legs.json"))
    print("host_ram", {k: d["host_ram_input"].get(k) for k in ("value", "pcie_GBs", "steps")}); print("host_path", d.get("host_path")); print("cfg2", {k: d["cfg2"].get(k) for k in ("value", "value_host_ram", "demod_kernel_ms_per_block", "steady_state_ms_per_step", "error")} if "cfg2" in d else None)
    print("stream_read", d["roofline"]["stream_read_GBs"], "fec", d.get("fec", {}).get("viterbi_kernel_trellis_steps_per_s"))
except Exception as e:
    print("legs failed", e)
PY
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
summ $OUT/bench_cfg4.json "cfg4"
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --host-input --sample-format cs16 > $OUT/bench_cfg3_host_cs16.json 2> $OUT/bench_cfg3_host_cs16.err
summ $OUT/bench_cfg3_host_cs16.json "cfg3 host-input cs16 256 steps"
# the C host program on a cs16 file, twice
python - <<PY > $OUT/replay_cs16.txt 2>&1
import sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, bench, dumphfdl_amd as hf
w = bench.WORKLOADS["cfg3"]
g = hf.plan_geometry(4096, 250 / w["fs"])
x, _ = bench.make_input(w, g.input_size, 0, 1)
for i in range(2):
    for fmt in ("CS16", "CF32"):
        r = bench.host_path_leg(w, x, bench.channel_plan(w), fmt)
        print(fmt, i, json.dumps({k: r.get(k) for k in ("value", "seconds", "blocks", "pdus", "thread_s", "pipeline_drains", "error")}))
PY
cat $OUT/replay_cs16.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch\|simple_timer\|generateRocpd\|tool.cpp" $f | tail -n 2; done
