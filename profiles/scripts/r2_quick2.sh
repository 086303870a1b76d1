cd /root/repo
mkdir -p gpurun_out/r2q
for ds in 0 1; do
HFDL_GPU_DECODE_STREAM=$ds timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2 decode_stream=$ds', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'])"
HFDL_GPU_DECODE_STREAM=$ds timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3 decode_stream=$ds', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['roofline']['frac'])"
HFDL_GPU_DECODE_STREAM=$ds timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg4 decode_stream=$ds', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['roofline']['frac'])"
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2q/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2q/pytest_all.log
tail -5 gpurun_out/r2q/pytest_all.log
