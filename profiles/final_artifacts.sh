#!/bin/bash
# Regenerates the measured artefacts kept under profiles/ (run through gpurun; results land in gpurun_out/final/).
#   bash profiles/final_artifacts.sh [commit]
C=${1:-$(cat /root/repo/profiles/scripts/commit.txt 2>/dev/null || echo unknown)}
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench.err
python bench.py --workload cfg2 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
python bench.py --workload cfg4 > $OUT/bench_cfg4.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_20steps.json 2>> $OUT/bench.err
python bench.py --host-input --sample-format cs16 --no-cpu-baseline --no-extra-legs > $OUT/bench_cfg3_host_cs16.json 2>> $OUT/bench.err
# the C host path on a cs16 file (raw samples over PCIe, converted on the device)
python - > $OUT/host_path_cs16.json 2>> $OUT/bench.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import bench
out = {}
for name in ("cfg3", "cfg2"):
    w = bench.WORKLOADS[name]
    import dumphfdl_amd as hf
    g = hf.plan_geometry(4096 if w["fs"] == 40_000_000 else 1024, 250 / w["fs"])
    x, _ = bench.make_input(w, g.input_size, 0, 1)
    out[name] = {fmt: bench.host_path_leg(w, x, bench.channel_plan(w), fmt) for fmt in ("CS16", "CF32")}
print(json.dumps(out))
PY
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4; do
	rm -rf /tmp/kt_$wl
	rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --no-extra-legs > $OUT/bench_${wl}_under_rocprof.json 2>/dev/null
	DB=$(find /tmp/kt_$wl -name "*.db" | head -1)
	python /root/repo/profiles/summarize_rocpd.py $DB "$wl -- rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-extra-legs (256 timed blocks + 8 warm-up; fft_pass* also run once per channel at create for the filter taps; commit $C)" > $OUT/${wl}_kernel_stats.md
	python /root/repo/profiles/timeline_rocpd.py $DB 2 > $OUT/${wl}_timeline.md
	bash /root/repo/profiles/pmc_passes.sh $wl $OUT $C > /dev/null 2>&1
done
ls -la $OUT
