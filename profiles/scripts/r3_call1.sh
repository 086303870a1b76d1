#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the default bench line (now with the cfg2 leg, per-rank rows, value_host_ram),
# a kernel trace of cfg3 and the LDS counters of the forward FFT after the pass-3 tile skew.
OUT=/root/repo/gpurun_out/r3a
mkdir -p $OUT
cd /root/repo
C=$(cat profiles/scripts/commit.txt 2>/dev/null || echo unknown)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt3
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -- python /root/repo/bench.py --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt3 -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg3 -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up; fft_pass* also run once per channel at create for the filter taps; commit $C)" > $OUT/cfg3_kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 2 > $OUT/cfg3_timeline.md
rm -rf /tmp/pmc3
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pmc3 -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > $OUT/pmc3.log 2>&1
python /root/repo/profiles/pmc_summary.py $(find /tmp/pmc3 -name "*.db") > $OUT/cfg3_pmc_lds.md
head -30 $OUT/cfg3_kernel_stats.md
grep fft_pass $OUT/cfg3_pmc_lds.md
