#!/bin/bash
OUT=/root/repo/gpurun_out/r4o
mkdir -p $OUT
cd /root/repo
HFDL_GPU_FOLD_BATCH=16 timeout 300 python profiles/fold_variants.py cfg3 3 > $OUT/fold_variants_cfg3_nb16.md 2> $OUT/fv.err
grep "^|" $OUT/fold_variants_cfg3_nb16.md | awk -F'|' 'NR<3 || ($7+0>=8)'
tail -2 $OUT/fv.err
