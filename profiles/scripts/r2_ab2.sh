cd /tmp && export TMPDIR=/tmp
for lib in new old; do
  if [ $lib = new ]; then unset HFDL_GPU_LIB; else export HFDL_GPU_LIB=/root/repo/exp_libs/libold.so; fi
  rm -rf /tmp/ktq && rocprofv3 --kernel-trace --stats -d /tmp/ktq -- python /root/repo/bench.py --no-cpu-baseline --no-extra-legs --steps 64 > /dev/null 2>&1
  echo "== $lib"; python /root/repo/profiles/summarize_rocpd.py $(find /tmp/ktq -name "*.db" | head -1) "cfg3 quick" | grep "ifft\|nco_table\|fft_pass1"
  python /root/repo/profiles/timeline_rocpd.py $(find /tmp/ktq -name "*.db" | head -1) 1 | grep -v demod
done
