#!/bin/bash
# Regenerates the measured artefacts kept under profiles/ (run through gpurun; results land in gpurun_out/final/).
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --no-cpu-baseline > $OUT/bench_cfg2.json 2>> $OUT/bench_cfg3.err
python bench.py --workload cfg4 --no-cpu-baseline > $OUT/bench_cfg4.json 2>> $OUT/bench_cfg3.err
python bench.py --host-input --no-cpu-baseline > $OUT/bench_cfg3_host_cf32.json 2>> $OUT/bench_cfg3.err
python bench.py --host-input --sample-format cs16 --no-cpu-baseline > $OUT/bench_cfg3_host_cs16.json 2>> $OUT/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python /root/repo/bench.py --no-cpu-baseline > $OUT/bench_cfg3_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg3 (40 Msps, 256 channels) -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (256 timed blocks + 8 warm-up; fft_pass* also run 256 x at create for the filter taps)" > $OUT/cfg3_kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 2 > $OUT/cfg3_timeline.md
/root/repo/profiles/pmc_passes.sh > $OUT/cfg3_pmc_counters.md 2>&1
