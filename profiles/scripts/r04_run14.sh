#!/bin/bash
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])")
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	rm -rf /tmp/kt_$wl
	rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --no-extra-legs > $OUT/bench_${wl}_under_rocprof.json 2>/dev/null
	DB=$(find /tmp/kt_$wl -name "*.db" | head -1)
	python /root/repo/profiles/summarize_rocpd.py $DB "$wl -- rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up, 8 blocks per fold launch; fft_pass* also run once per channel at create for the filter taps; commit $C)" > $OUT/${wl}_kernel_stats.md
	python /root/repo/profiles/timeline_rocpd.py $DB 1 > $OUT/${wl}_timeline.md
	python - <<PY
import json
d = json.load(open("$OUT/bench_${wl}_under_rocprof.json")); print("$wl under rocprof: value %.0f fold avg %.4f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
	grep "fold_kernel" $OUT/${wl}_kernel_stats.md | head -2
done
