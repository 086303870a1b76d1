#!/bin/bash
# round 5, run 21: what would a K = 4 fold gain?  Timing probe: the sixteen-column loop with 16x16x4 in the place of 16x16x1_4B
mkdir -p gpurun_out/r5t
timeout 600 python profiles/fold_variants.py cfg3 5 16 > gpurun_out/r5t/fold_variants_k4_probe.md 2> gpurun_out/r5t/err.txt
grep -E "PROBE|16x16x1_4B \| 2 \| 4 \| 4|16x16x1_4B \| 1" gpurun_out/r5t/fold_variants_k4_probe.md | head; tail -3 gpurun_out/r5t/err.txt
