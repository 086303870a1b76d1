#!/bin/bash
# Regenerates the measured artefacts kept under profiles/ (run through gpurun; results land in gpurun_out/final/).
#   profiles/stamp.sh                      (here, where git is: names the commit + csrc hash the artefacts are stamped with)
#   gpurun -- 'bash profiles/final_artifacts.sh'
#   profiles/collect_final.sh r06          (copies gpurun_out/final/* into profiles/)
C=$(python -c "import json; print(json.load(open('/root/repo/profiles/scripts/stamp.json'))['commit'])" 2>/dev/null || echo unknown)
OUT=/root/repo/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
echo "commit $C" > $OUT/README.txt
# the PMC passes first: the traffic records they leave under profiles/ are what the bench lines below quote (roofline.traffic)
( cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > $OUT/pmc_$wl.log 2>&1
	cp $OUT/fold_traffic_$wl.json /root/repo/profiles/fold_traffic_$wl.json
done )
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
cp gpurun_out/low_snr_sweep.json $OUT/low_snr_sweep.json 2>/dev/null
cp gpurun_out/r05_eight_rank_cfg3.json $OUT/eight_rank_cfg3.json 2>/dev/null      # written by tests/test_gpu_configs.py::test_eight_rank_rehearsal_at_full_size of the suite above
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
python bench.py --workload cfg4 > $OUT/bench_cfg4.json 2>> $OUT/bench.err
python bench.py --workload cfg1 --steps 523 --warmup 0 > $OUT/bench_cfg1.json 2>> $OUT/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg3_driver_line.json 2>> $OUT/bench.err
for nb in 1 16; do
	HFDL_GPU_FOLD_BATCH=$nb python bench.py --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_fold_batch_$nb.json 2>> $OUT/bench.err
done
python bench.py --steps 1024 --warmup 32 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_1024_steps.json 2>> $OUT/bench.err
python bench.py --host-input --sample-format cs16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_host_cs16.json 2>> $OUT/bench.err
python bench.py --host-input --sample-format cf32 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_host_cf32.json 2>> $OUT/bench.err
# the N > 1 launch path through RCCL at world size 1, exactly as the driver starts it
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extra-legs 2>> $OUT/bench.err | grep "^{" > $OUT/bench_cfg3_rccl_world1.json
# the C host path on cf32 and cs16 files (raw samples over PCIe, converted on the device)
python - > $OUT/host_path.json 2>> $OUT/bench.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
out = {}
for name in ("cfg3", "cfg2"):
    w = bench.WORKLOADS[name]
    import dumphfdl_amd as hf
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    out[name] = {fmt: [bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for _ in range(2)] for fmt in ("CS16", "CF32")}
print(json.dumps(out))
PY
HFDL_GPU_FOLD_BATCH=32 timeout 600 python profiles/fold_variants.py cfg3 3 1,4,8,16,24,32 > $OUT/fold_variants_cfg3.md 2>> $OUT/bench.err
GPU_MAX_HW_QUEUES=16 timeout 300 python profiles/neighbour_probe.py cfg3 120 > $OUT/neighbour_probe_cfg3.md 2>> $OUT/bench.err
timeout 300 python profiles/clock_probe.py cfg3 160 > $OUT/clock_probe_cfg3.md 2>> $OUT/bench.err
timeout 300 python profiles/fft_accuracy.py > $OUT/fft_accuracy.txt 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	rm -rf /tmp/kt_$wl
	rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --no-extra-legs > $OUT/bench_${wl}_under_rocprof.json 2>/dev/null
	DB=$(find /tmp/kt_$wl -name "*.db" | head -1)
	python /root/repo/profiles/summarize_rocpd.py $DB "$wl -- rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up; cfg3 / cfg4: fold launches of 8 (warm-up), 16 (first half after a drain) and 32 blocks; the fft passes also run once per channel at create for the filter taps, and once more for the stream-read probe's laboratory front end; commit $C)" > $OUT/${wl}_kernel_stats.md
	python /root/repo/profiles/timeline_rocpd.py $DB 1 > $OUT/${wl}_timeline.md
done
rm -rf /tmp/kt20
rocprofv3 --kernel-trace --stats -d /tmp/kt20 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_driver_line_under_rocprof.json 2>/dev/null
python /root/repo/profiles/timeline_tail.py $(find /tmp/kt20 -name "*.db" | head -1) -56 > $OUT/cfg3_driver_line_timeline.md
ls -la $OUT
