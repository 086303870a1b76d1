# same-box A/B of two builds of the library on the demodulator-bound workload: wave busy cycles and bench value
cd /root/repo
for lib in "$@"; do
  echo "== $lib"
  HFDL_GPU_LIB=/root/repo/$lib timeout 200 python profiles/phase_probe.py cfg2 2>&1 | grep alone | tail -2 | sed 's/.*per sample/per sample/'
  for i in 1 2; do HFDL_GPU_LIB=/root/repo/$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg2', round(r['value'],1), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'])"; done
done
