#!/bin/bash
# round 6, call 6: is the demodulator's slow-down under the fold local to the SIMDs it shares with fold waves?  Laboratory A/B: the
# demodulator's stream on every 2nd CU, the channelizer's on the others (HFDL_GPU_CU_SPLIT=2), one block per demodulator launch so that two
# workgroups fit a CU.  Clocks and cycles from inside the kernels.
OUT=/root/repo/gpurun_out/r6f
mkdir -p $OUT
cd /root/repo
export HFDL_GPU_DEMOD_BATCH=1
echo "=== no split, one block per demodulator launch"
timeout 600 python profiles/clock_probe.py cfg3 160 2> $OUT/cp_nosplit.err | tee $OUT/clock_probe_nosplit.md | grep -A8 "^## pipeline" | cut -c1-250
echo "=== split 2"
HFDL_GPU_CU_SPLIT=2 timeout 600 python profiles/clock_probe.py cfg3 160 2> $OUT/cp_split2.err | tee $OUT/clock_probe_split2.md | grep -A8 "^## pipeline" | cut -c1-250
echo "=== split 4 (two of four demodulator workgroups wait for a CU)"
HFDL_GPU_CU_SPLIT=4 timeout 600 python profiles/clock_probe.py cfg3 160 2> $OUT/cp_split4.err | tee $OUT/clock_probe_split4.md | grep -A8 "^## pipeline" | cut -c1-250
export HFDL_GPU_LIB=/root/repo/dumphfdl_amd/libhfdl_gpu_lab.so
B="python bench.py --no-cpu-baseline --no-extra-legs"
for sp in 0 2; do
	HFDL_GPU_CU_SPLIT=$sp timeout 400 $B --steps 256 --warmup 32 > $OUT/b256_split$sp.json 2> $OUT/b256_split$sp.err
	python - <<PY
import json
try:
    d = json.load(open("$OUT/b256_split$sp.json"))
    print("split $sp: value %.0f ms/step %.4f steady %s demod/blk %s x%s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"]), {k: round(v, 4) for k, v in d["streams"]["per_block_ms"].items()})
except Exception as e:
    print("split $sp failed", e)
PY
done
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
