#!/bin/bash
# round 5, run 19: timeline of the driver's 20-step run (16 + 4 blocks) with the four-column form on the 4-block launch
mkdir -p /root/repo/gpurun_out/r5r
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt20
rocprofv3 --kernel-trace --stats -d /tmp/kt20 -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > /root/repo/gpurun_out/r5r/bench_20_under_rocprof.json 2>/dev/null
DB=$(find /tmp/kt20 -name "*.db" | head -1)
python /root/repo/profiles/timeline_tail.py $DB -56 > /root/repo/gpurun_out/r5r/timeline_20_steps.md
cat /root/repo/gpurun_out/r5r/timeline_20_steps.md
cd /root/repo
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['launch_shapes'])"; done
