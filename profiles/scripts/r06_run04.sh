#!/bin/bash
# round 6, call 4: the timing-recovery output ring (30 instead of 46 bytes of LDS per sample) and three blocks per demodulator launch:
# the GPU suite, then A/B against two blocks per launch and against the build before the ring
OUT=/root/repo/gpurun_out/r6d
mkdir -p $OUT
cd /root/repo
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $OUT/pytest_gpu.log 2>&1
tail -n 6 $OUT/pytest_gpu.log | cut -c1-200
summ() {
python - "$1" "$2" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s value %.0f ms/step %.4f steady %s bound %s frac %.3f demod/blk %s x%s fill_drain %.2f pdus %d/%d" % (sys.argv[2], d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["bound"], r["frac"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"], d.get("fill_drain_ms") or 0, d["pdus_matching_sent_payload"], d["pdus_in_timed_region"]))
    print("   streams", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("streams", {}).items() if k != "note"})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_r$rep.json 2> $OUT/b256.err; summ $OUT/b256_r$rep.json "256 steps ring, 3 blocks / launch"
HFDL_GPU_DEMOD_BATCH=2 timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_db2_r$rep.json 2> $OUT/b256_db2.err; summ $OUT/b256_db2_r$rep.json "256 steps ring, 2 blocks / launch"
HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_noring.so timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_noring_r$rep.json 2> $OUT/b256_noring.err; summ $OUT/b256_noring_r$rep.json "256 steps before the ring (2 blocks)"
done
timeout 600 $B --steps 1024 --warmup 32 > $OUT/b1024.json 2> $OUT/b1024.err; summ $OUT/b1024.json "1024 steps ring, 3 blocks"
HFDL_GPU_DEMOD_BATCH=2 timeout 600 $B --steps 1024 --warmup 32 > $OUT/b1024_db2.json 2> $OUT/b1024_db2.err; summ $OUT/b1024_db2.json "1024 steps ring, 2 blocks"
for rep in 1 2; do
timeout 400 $B --steps 20 --warmup 5 > $OUT/b20_r$rep.json 2> $OUT/b20.err; summ $OUT/b20_r$rep.json "20 steps ring, 3 blocks"
HFDL_GPU_DEMOD_BATCH=2 timeout 400 $B --steps 20 --warmup 5 > $OUT/b20_db2_r$rep.json 2> $OUT/b20_db2.err; summ $OUT/b20_db2_r$rep.json "20 steps ring, 2 blocks"
done
timeout 400 $B --workload cfg2 --steps 256 --warmup 32 > $OUT/b256_cfg2.json 2> $OUT/b256_cfg2.err; summ $OUT/b256_cfg2.json "cfg2 256 steps ring"
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
