#!/bin/bash
OUT=/root/repo/gpurun_out/r4g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python /root/repo/bench.py --workload cfg3 --steps 64 --warmup 8 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg3 nb8" > $OUT/kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 2 > $OUT/timeline.md
cat $OUT/kernel_stats.md
cat $OUT/timeline.md | head -80
