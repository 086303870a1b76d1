cd /root/repo
C=$(cat profiles/scripts/commit.txt)
for wl in cfg2 cfg3 cfg4; do bash profiles/pmc_passes.sh $wl /root/repo/gpurun_out/r2pmc $C > /dev/null 2>&1; done
ls -la gpurun_out/r2pmc; cat gpurun_out/r2pmc/fold_traffic_cfg2.json | head -30
grep "demod_kernel\|burst_decode" gpurun_out/r2pmc/cfg2_pmc_counters.md | head -30
