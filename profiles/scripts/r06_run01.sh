#!/bin/bash
# round 6, call 1: baseline at the round-5 tree on this round's box: GPU suite, the driver's line, 256 steps, demodulator phase cycles
OUT=/root/repo/gpurun_out/r6a
mkdir -p $OUT
cd /root/repo
(time timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest_gpu.log 2>&1
tail -n 3 $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench_driver_line.err
timeout 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_256.json 2> $OUT/bench_256.err
python - <<PY
import json
for n in ("bench_driver_line", "bench_256"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        r = d["roofline"]
        print(n, "value %.0f ms/step %.4f steady %s fold_avg %.3f frac %.3f demod/blk %s fill_drain %s" % (d["value"], d["ms_per_step"], d.get("steady_state_ms_per_step"), r["avg_launch_ms"], r["frac"], d.get("demod_kernel_ms_per_block"), d.get("fill_drain_ms")))
        print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python profiles/phase_probe.py cfg3 > $OUT/phase_probe_cfg3.txt 2>&1
tail -n 8 $OUT/phase_probe_cfg3.txt
