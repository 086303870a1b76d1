cd /root/repo
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pytest_all.log
tail -4 gpurun_out/r2f/pytest_all.log
for wl in cfg3 cfg2 cfg4; do
timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$wl', round(r['value'],1), round(r['ms_per_step'],4), round(r['steady_state_ms_per_step'],4), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], round(r['roofline']['frac'],4))"
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktq && rocprofv3 --kernel-trace --stats -d /tmp/ktq -- python /root/repo/bench.py --no-cpu-baseline --no-extra-legs --steps 64 > /dev/null 2>&1
python /root/repo/profiles/summarize_rocpd.py $(find /tmp/ktq -name "*.db" | head -1) "cfg3 quick" | head -12
