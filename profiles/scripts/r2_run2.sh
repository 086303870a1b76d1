cd /root/repo
mkdir -p gpurun_out/r2b
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
timeout 600 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r2b/bench_cfg4.json 2> gpurun_out/r2b/bench_cfg4.err; echo rc=$?
timeout 600 python bench.py --no-cpu-baseline --steps 64 2> gpurun_out/r2b/bench_cfg3.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({k:r[k] for k in ('value','host_ram_input','host_path')}))"
