cd /root/repo
for rep in 1 2 3; do
for lib in new old; do
  if [ $lib = new ]; then unset HFDL_GPU_LIB; else export HFDL_GPU_LIB=/root/repo/exp_libs/libold.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 128 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg3 $lib', round(r['value'],1), round(r['steady_state_ms_per_step'],4), 'fold', round(r['roofline']['avg_launch_ms'],4))"
done
done
