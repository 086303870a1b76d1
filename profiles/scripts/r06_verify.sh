#!/bin/bash
# round 6, verification at the end-of-round tree: smoke, the GPU suite as the driver runs it, the driver's bench command
OUT=/root/repo/gpurun_out/r6v
mkdir -p $OUT
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python -m pytest tests/ -x -q -m gpu) > $OUT/pytest_gpu.log 2>&1; tail -n 5 $OUT/pytest_gpu.log
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -n 4 $OUT/bench_driver.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[0])
r = d["roofline"]
print("driver line: value %.0f ms/step %.4f bound %s frac %.3f traffic %s latency %.2f ms cpu_baseline %.1f parity pdus identical %s" % (d["value"], d["ms_per_step"], r["bound"], r["frac"], r["traffic"], d["block_to_pdus_latency_ms"], d["cpu_baseline"]["value"], d["parity"]["pdu_multisets_identical"]))
PY
