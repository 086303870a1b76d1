#!/bin/bash
OUT=/root/repo/gpurun_out/r4j
mkdir -p $OUT
cd /root/repo
timeout 900 python profiles/fold_variants.py cfg3 3 > $OUT/fold_variants_cfg3.md 2> $OUT/fold_variants_cfg3.err
grep "^|" $OUT/fold_variants_cfg3.md | awk -F'|' '$2+0>=2 || NR<3'
tail -3 $OUT/fold_variants_cfg3.err
