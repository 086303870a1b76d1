#!/bin/bash
# round 6: the thirty-two-column tiling with two quads in flight (1, 4, 2: 340 registers) against the product's four (1, 4, 4: 420) at one
# slice of alias rows, alone and in the pipeline
OUT=/root/repo/gpurun_out/r6r
mkdir -p $OUT
cd /root/repo
export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so
export HFDL_GPU_LAB_LIB=$HFDL_GPU_LIB
HFDL_GPU_FOLD_BATCH=32 FOLD_VARIANTS=3,4,5,25 timeout 600 python profiles/fold_variants.py cfg3 3 32 2> $OUT/fv.err | grep "^| " | cut -c1-160
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2 3; do
for tile in -1 4; do
HFDL_GPU_FOLD_TILE=$tile timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_t${tile}_r$rep.json 2> $OUT/b.err
python - <<PY
import json
d = json.load(open("$OUT/b256_t${tile}_r$rep.json"))
print("tile $tile: value %.0f steady %.4f fold %.3f demod %.3f frac %.3f" % (d["value"], d["steady_state_ms_per_step"], d["streams"]["per_block_ms"]["fold"], d["streams"]["per_block_ms"]["demod"], d["roofline"]["frac"]))
PY
done
done
