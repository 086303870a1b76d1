cd /root/repo
for c in 32 16 24 48 32 16; do
  if [ $c = 32 ]; then unset HFDL_GPU_LIB; else export HFDL_GPU_LIB=/root/repo/exp_libs/libchunk$c.so; fi
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2 chunk=$c', round(r['value'],1), round(r['ms_per_step'],4), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'])"
done
