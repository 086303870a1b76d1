#!/bin/bash
# round 5, fifth measurement: stream A no longer waits for a burst decoder, demodulator batch capped at 2 where the fold bounds, fold default P=2 W=4 D=4
OUT=/root/repo/gpurun_out/r5e
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
for cfg in "16 2 96" "16 1 96" "16 2 256" "8 2 96"; do
	set -- $cfg
	HFDL_GPU_FOLD_BATCH=$1 HFDL_GPU_DEMOD_BATCH=$2 timeout 300 python bench.py --steps $3 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_nb$1_db$2_$3.json 2> $OUT/bench_cfg3_nb$1_db$2_$3.err
	summ $OUT/bench_cfg3_nb$1_db$2_$3.json "cfg3 fold_batch=$1 demod_batch=$2 steps=$3"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_20.json 2> $OUT/bench_cfg3_20.err
summ $OUT/bench_cfg3_20.json "cfg3 driver-line 20 steps"
for fmt in cf32 cs16; do
	timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extra-legs --host-input --sample-format $fmt > $OUT/bench_cfg3_host_$fmt.json 2> $OUT/bench_cfg3_host_$fmt.err
	summ $OUT/bench_cfg3_host_$fmt.json "cfg3 host-input $fmt"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg3 -- python /root/repo/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_prof.json 2> $OUT/bench_cfg3_prof.err
cd /root/repo
DB=$(find /tmp/prof_cfg3 -name "*.db" | head -1)
python profiles/summarize_rocpd.py $DB "cfg3 -- rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-extra-legs (r05 work in progress)" > $OUT/kernel_stats_cfg3.md 2>$OUT/kernel_stats.err
head -12 $OUT/kernel_stats_cfg3.md
python profiles/timeline_rocpd.py $DB > $OUT/timeline_cfg3.md 2>> $OUT/kernel_stats.err
grep -v "fft_rpass\|copyBuffer" $OUT/timeline_cfg3.md | head -40
cd /tmp
for set in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
	tag=$(echo $set | tr ' ' '_' | cut -c1-30)
	FOLD_VARIANTS=0,4 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcf_$tag -- python /root/repo/profiles/fold_variants.py cfg3 2 8,16 > $OUT/fv_$tag.md 2> $OUT/fv_$tag.err
done
python /root/repo/profiles/pmc_summary.py $(find /tmp/pmcf_* -name "*.db" | sort) 2>/dev/null | grep "fold_mfma16" > $OUT/fold_pmc.md
cat $OUT/fold_pmc.md; grep "^| 16x16" $OUT/fv_GRBM_GUI_ACTIVE.md
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch\|simple_timer\|generateRocpd\|tool.cpp" $f | tail -n 2; done
