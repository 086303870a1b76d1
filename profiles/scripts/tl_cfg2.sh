cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -- python /root/repo/bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs --steps 64 --warmup 8 > /tmp/tl.log 2>&1
DB=$(find /tmp/tl -name "*.db" | head -1)
python /root/repo/profiles/timeline_rocpd.py $DB 3
tail -1 /tmp/tl.log | cut -c1-200
