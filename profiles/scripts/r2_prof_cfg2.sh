cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r2p
rm -rf /tmp/kt2
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python /root/repo/bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > /root/repo/gpurun_out/r2p/bench_cfg2_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python /root/repo/profiles/summarize_rocpd.py $DB "cfg2 (8 Msps, 32 channels) -- rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs" > /root/repo/gpurun_out/r2p/cfg2_kernel_stats.md
python /root/repo/profiles/timeline_rocpd.py $DB 3 > /root/repo/gpurun_out/r2p/cfg2_timeline.md
cat /root/repo/gpurun_out/r2p/cfg2_kernel_stats.md /root/repo/gpurun_out/r2p/cfg2_timeline.md
