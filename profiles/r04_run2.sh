#!/bin/bash
OUT=/root/repo/gpurun_out/r4n
mkdir -p $OUT
cd /root/repo
timeout 300 python profiles/fold_variants.py cfg3 3 > $OUT/fold_variants_cfg3.md 2> $OUT/fv.err
grep "^|" $OUT/fold_variants_cfg3.md | awk -F'|' 'NR<3 || ($2+0>=2 && ($7+0==8 || $7+0==4))'
for s in 64 20; do
python bench.py --steps $s --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/b.json 2> $OUT/b.err
python - <<PY
import json
d = json.load(open("$OUT/b.json")); r = d["roofline"]
print("steps $s value %.0f ms/step %.4f steady %.4f fold_avg %.3f (%.1f blk) frac %.3f pdus %d/%d demod/blk %.3f" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], r["avg_launch_ms"], r["blocks_per_launch"], r["frac"], d["pdus_matching_sent_payload"], d["pdus_in_timed_region"], d["demod_kernel_ms_per_block"]))
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "demodulator_stage or fold_batching" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pm1 -- python /root/repo/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
python /root/repo/profiles/pmc_summary.py $(find /tmp/pm1 -name "*.db") | grep -i "fold\|stream_read" | head
