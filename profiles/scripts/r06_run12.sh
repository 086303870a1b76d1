#!/bin/bash
# round 6, call 14: kernel-by-kernel timeline of the driver's 20-step line at the current tree
OUT=/root/repo/gpurun_out/r6n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt20
rocprofv3 --kernel-trace --stats -d /tmp/kt20 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_driver_line_under_rocprof.json 2>/dev/null
python /root/repo/profiles/timeline_tail.py $(find /tmp/kt20 -name "*.db" | head -1) -70 > $OUT/cfg3_driver_line_timeline.md
grep -v "fft_rpass" $OUT/cfg3_driver_line_timeline.md | cut -c1-150 | head -80
