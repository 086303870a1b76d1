#!/bin/bash
# HBM traffic and issue statistics per kernel: SEPARATE rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss, SQ),
# each with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
#   profiles/pmc_passes.sh <workload> <outdir> [commit]
# The profiled command folds 8 blocks (the warm-up, closed by a poll), then -- 68 timed blocks where the fold bounds the block -- 16 (the
# first half after a drain), 32, and the ragged 20 as 16 + 4: the launch shapes of the driver's 20-step line (16 + 4) and of the long
# runs (32).
WL=${1:-cfg3}
OUT=${2:-/root/repo/gpurun_out/pmc_$WL}
COMMIT=${3:-unknown}
# fold launches of the profiled command in launch order: the warm-up's 8 blocks (closed by a poll), then the 68 timed ones: 16 + 32 +
# (16 + 4) (cfg3 / cfg4: the fold bounds the block) or halves of 8 (cfg2: the demodulator does), then the five one-block launches of the
# bench's latency leg (block_to_pdus_latency_ms)
if [ "$WL" = "cfg2" ]; then SHAPES=8,8,8,8,8,8,8,8,8,4,1,1,1,1,1; else SHAPES=8,16,32,16,4,1,1,1,1,1; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_${WL}_*
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
	d=/tmp/pmc_${WL}_$(echo $c | tr ' ' '_' | cut -c1-40)
	rocprofv3 --pmc $c --kernel-trace -d $d -- python /root/repo/bench.py --workload $WL --steps 68 --warmup 8 --no-cpu-baseline --no-extra-legs > $d.log 2>&1
done
DBS=$(find /tmp/pmc_${WL}_* -name "*.db" | sort)
{
	echo "# $WL PMC passes (profiles/pmc_passes.sh $WL: rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --workload $WL --steps 68 --warmup 8 --no-cpu-baseline --no-extra-legs, one pass per counter set; commit $COMMIT)"
	echo
	echo "Per-dispatch averages. FETCH_SIZE / WRITE_SIZE in KB; on gfx950 reads = 2 x FETCH_SIZE for wide coalesced streaming reads (calibrated in the same run on stream_read_kernel, which reads a known byte count). SQ_* count quad-cycles summed over the dispatch's waves."
	echo
	python /root/repo/profiles/pmc_summary.py $DBS
} > $OUT/${WL}_pmc_counters.md
python /root/repo/profiles/fold_traffic.py $WL $COMMIT $SHAPES $DBS > $OUT/fold_traffic_${WL}.json || { echo "fold_traffic refused"; rm -f $OUT/fold_traffic_${WL}.json; }
cat $OUT/fold_traffic_${WL}.json
