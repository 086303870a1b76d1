cd /root/repo
mkdir -p gpurun_out/r2exp
for e in 0 1 2 3 4 5; do
  if [ $e = 0 ]; then unset HFDL_GPU_LIB; else export HFDL_GPU_LIB=/root/repo/exp_libs/libexp$e.so; fi
  echo "== exp $e" >> gpurun_out/r2exp/phase.txt
  timeout 120 python profiles/phase_probe.py cfg2 2>&1 | grep "alone" | tail -1 >> gpurun_out/r2exp/phase.txt
done
cat gpurun_out/r2exp/phase.txt
