#!/bin/bash
# kernel statistics of the other workloads (results under gpurun_out/final/)
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg4; do
	rm -rf /tmp/kt
	rocprofv3 --kernel-trace --stats -d /tmp/kt -- python /root/repo/bench.py --no-cpu-baseline --workload $wl > $OUT/bench_${wl}_under_rocprof.json 2>/dev/null
	DB=$(find /tmp/kt -name "*.db" | head -1)
	python /root/repo/profiles/summarize_rocpd.py $DB "$wl -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --workload $wl (256 timed blocks + 8 warm-up; fft_pass* also run once per channel at create for the filter taps)" > $OUT/${wl}_kernel_stats.md
	python /root/repo/profiles/timeline_rocpd.py $DB 1 > $OUT/${wl}_timeline.md
done
