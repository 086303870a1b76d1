#!/bin/bash
# round 6, call 3: the GPU suite at the new default (halves of 32 after a first half of 16); bench at 20 / 256 / 1024 steps; A/B: no ramp,
# the two-wave 32-column tiling, one / three blocks per demodulator launch
OUT=/root/repo/gpurun_out/r6c
mkdir -p $OUT
cd /root/repo
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12) > $OUT/pytest_gpu.log 2>&1
tail -n 22 $OUT/pytest_gpu.log | cut -c1-200
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s bound %s frac %.3f (mfma %.3f hbm %.3f) demod/blk %s fill_drain %s pdus %d/%d" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["bound"], r["frac"], r["mfma"]["frac"], r["hbm"]["frac"], d["demod_kernel_ms_per_block"], d.get("fill_drain_ms"), d["pdus_matching_sent_payload"], d["pdus_in_timed_region"]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
    print("   shapes", {k: (v["launches"], round(v["avg_ms"], 3)) for k, v in r["launch_shapes"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --no-cpu-baseline --no-extra-legs"
timeout 400 $B --steps 20 --warmup 5 > $OUT/b20.json 2> $OUT/b20.err; summ $OUT/b20.json "20 steps (driver)"
timeout 400 $B --steps 256 --warmup 32 > $OUT/b256.json 2> $OUT/b256.err; summ $OUT/b256.json "256 steps"
timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_2.json 2> $OUT/b256_2.err; summ $OUT/b256_2.json "256 steps (again)"
timeout 600 $B --steps 1024 --warmup 32 > $OUT/b1024.json 2> $OUT/b1024.err; summ $OUT/b1024.json "1024 steps"
HFDL_GPU_DEMOD_BATCH=1 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_db1.json 2> $OUT/b256_db1.err; summ $OUT/b256_db1.json "256 steps demod_batch=1"
HFDL_GPU_FOLD_BATCH=16 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_fb16.json 2> $OUT/b256_fb16.err; summ $OUT/b256_fb16.json "256 steps fold_batch=16"
export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so
HFDL_GPU_FOLD_RAMP=0 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_noramp.json 2> $OUT/b256_noramp.err; summ $OUT/b256_noramp.json "256 steps lab, no ramp"
HFDL_GPU_FOLD_TILE=5 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_tile5.json 2> $OUT/b256_tile5.err; summ $OUT/b256_tile5.json "256 steps lab, tiling F32(1,8,2)"
HFDL_GPU_FOLD_TILE=4 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_tile4.json 2> $OUT/b256_tile4.err; summ $OUT/b256_tile4.json "256 steps lab, tiling F32(1,4,2)"
unset HFDL_GPU_LIB
timeout 400 $B --workload cfg4 --steps 256 --warmup 32 > $OUT/b256_cfg4.json 2> $OUT/b256_cfg4.err; summ $OUT/b256_cfg4.json "cfg4 256 steps"
timeout 400 $B --workload cfg2 --steps 256 --warmup 32 > $OUT/b256_cfg2.json 2> $OUT/b256_cfg2.err; summ $OUT/b256_cfg2.json "cfg2 256 steps"
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
