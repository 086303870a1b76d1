set -x
cd /root/repo
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python bench.py > gpurun_out/r2a/bench_cfg3.json 2> gpurun_out/r2a/bench_cfg3.err; echo rc=$?
timeout 300 python bench.py --workload cfg2 > gpurun_out/r2a/bench_cfg2.json 2> gpurun_out/r2a/bench_cfg2.err; echo rc=$?
timeout 200 python profiles/phase_probe.py cfg2 > gpurun_out/r2a/phase_cfg2.txt 2>&1
timeout 200 python profiles/phase_probe.py cfg3 > gpurun_out/r2a/phase_cfg3.txt 2>&1
cat gpurun_out/r2a/phase_cfg2.txt gpurun_out/r2a/phase_cfg3.txt
nproc; free -g | head -2
