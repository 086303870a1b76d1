#!/bin/bash
OUT=/root/repo/gpurun_out/r4k
mkdir -p $OUT
cd /root/repo
timeout 120 python profiles/baseband_smoke.py > $OUT/baseband_smoke.txt 2>&1 || { echo "baseband smoke failed / hung"; tail -5 $OUT/baseband_smoke.txt; exit 1; }
cat $OUT/baseband_smoke.txt
timeout 900 python profiles/strict_study.py --bins=-8:2:2 > $OUT/strict_study.json 2> $OUT/strict_study.md
cat $OUT/strict_study.md
