#!/bin/bash
# round 6, call 13: the GPU suite with the fold's new slice rule (channels x slices >= 256: one slice at 256 channels) and the 65-entry
# LDS pitch; the live-pipe latency bound; bench lines of cfg3 / cfg2 / cfg4 (product library)
OUT=/root/repo/gpurun_out/r6m
mkdir -p $OUT
cd /root/repo
(time timeout 1800 python -m pytest tests -m gpu -x -q --durations=10) > $OUT/pytest_gpu.log 2>&1
tail -n 20 $OUT/pytest_gpu.log | cut -c1-220
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s bound %s frac %.3f demod/blk %s x%s fill_drain %.2f parity %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["bound"], r["frac"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"], d.get("fill_drain_ms") or 0, json.dumps(d.get("parity"))[:200]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).get("per_block_ms", {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --no-cpu-baseline --no-extra-legs"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/b20_driver.json 2> $OUT/b20_driver.err; summ $OUT/b20_driver.json "driver line"
timeout 400 $B --steps 256 --warmup 32 > $OUT/b256.json 2> $OUT/b256.err; summ $OUT/b256.json "cfg3 256 steps"
for sl in 32 8 2 1; do
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so HFDL_GPU_FOLD_SLICES=$sl timeout 400 $B --workload cfg2 --steps 256 --warmup 32 > $OUT/b256_cfg2_s$sl.json 2> $OUT/b256_cfg2_s$sl.err; summ $OUT/b256_cfg2_s$sl.json "cfg2 256 steps, $sl slices (lab)"
done
timeout 400 $B --workload cfg2 --steps 256 --warmup 32 > $OUT/b256_cfg2.json 2> $OUT/b256_cfg2.err; summ $OUT/b256_cfg2.json "cfg2 256 steps"
timeout 400 $B --workload cfg4 --steps 256 --warmup 32 > $OUT/b256_cfg4.json 2> $OUT/b256_cfg4.err; summ $OUT/b256_cfg4.json "cfg4 256 steps"
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
