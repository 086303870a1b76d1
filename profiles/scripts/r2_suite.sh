cd /root/repo
mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2s/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2s/pytest_gpu.log
tail -5 gpurun_out/r2s/pytest_gpu.log
for w in cfg2 cfg4 cfg3; do
timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['roofline']['frac'])"
done
