#!/bin/bash
# round 5, third measurement: where the matrix-pipe fold's time goes -- SQ counters on three tilings at 16 blocks per launch
OUT=/root/repo/gpurun_out/r5c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
	tag=$(echo $set | tr ' ' '_' | cut -c1-30)
	FOLD_VARIANTS=2,4,7,14 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcf_$tag -- python /root/repo/profiles/fold_variants.py cfg3 2 8,16 > $OUT/fv_$tag.md 2> $OUT/fv_$tag.err
done
python /root/repo/profiles/pmc_summary.py $(find /tmp/pmcf_* -name "*.db" | sort) > $OUT/fold_pmc.md 2> $OUT/fold_pmc.err
cat $OUT/fold_pmc.md | head -60
tail -3 $OUT/fold_pmc.err
grep "^| 2\|^| 4" $OUT/fv_FETCH_SIZE.md
