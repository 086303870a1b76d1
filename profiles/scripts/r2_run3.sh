cd /root/repo
mkdir -p gpurun_out/r2c
timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or end_to_end_small or burst_dense or viterbi" > gpurun_out/r2c/pytest_quick.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pytest_quick.log
tail -30 gpurun_out/r2c/pytest_quick.log
timeout 200 python profiles/phase_probe.py cfg2 > gpurun_out/r2c/phase_cfg2.txt 2>&1
timeout 200 python profiles/phase_probe.py cfg3 > gpurun_out/r2c/phase_cfg3.txt 2>&1
grep cycles gpurun_out/r2c/phase_cfg2.txt gpurun_out/r2c/phase_cfg3.txt
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs > gpurun_out/r2c/bench_cfg2.json 2> gpurun_out/r2c/bench_cfg2.err; echo rc=$?
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r2c/bench_cfg3.json 2> gpurun_out/r2c/bench_cfg3.err; echo rc=$?
python - <<'PY'
import json
for f in ("cfg2","cfg3"):
    try:
        r=json.load(open("gpurun_out/r2c/bench_%s.json"%f)); print(f, r["value"], r["ms_per_step"], r["pdus_in_timed_region"], r["pdus_matching_sent_payload"], r["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -12 gpurun_out/r2c/pytest.log
