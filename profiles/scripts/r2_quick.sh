cd /root/repo
mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or end_to_end or burst_dense or marginal or many_frames or long_idle or collect or ring" > gpurun_out/r2q/pytest_quick.log 2>&1; echo "rc=$?" >> gpurun_out/r2q/pytest_quick.log
tail -5 gpurun_out/r2q/pytest_quick.log
timeout 200 python profiles/phase_probe.py cfg2 2>&1 | grep cycles | tail -4
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3', r['value'], r['ms_per_step'], r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['roofline']['frac'])"
