#!/bin/bash
# round 5, run 32: the clock the board runs the K = 4 fold at (GRBM_GUI_ACTIVE per launch / its duration), product tiling (2, 4, 4) and the
# two-wave tiling (2, 4, 2), alone on the resident taps: profiles/fold_variants.py under two counter passes
OUT=/root/repo/gpurun_out/r5ad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
	tag=$(echo $set | tr ' ' '_' | cut -c1-30)
	rm -rf /tmp/pmck_$tag
	FOLD_VARIANTS=0,1 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmck_$tag -- python /root/repo/profiles/fold_variants.py cfg3 3 16 > $OUT/fv_$tag.md 2> $OUT/fv_$tag.err
done
python /root/repo/profiles/pmc_summary.py $(find /tmp/pmck_* -name "*.db" | sort) 2>/dev/null | grep "fold_mfma16" > $OUT/fold_pmc_k4.md
cat $OUT/fold_pmc_k4.md; grep "^| 16x16" $OUT/fv_GRBM_GUI_ACTIVE.md; tail -2 $OUT/fv_GRBM_GUI_ACTIVE.err
