#!/bin/bash
# round 6, call 5: clocks measured inside the kernels (lab), board power per kind of work (rocm-smi), even demodulator launches (cfg2)
OUT=/root/repo/gpurun_out/r6e
mkdir -p $OUT
cd /root/repo
timeout 600 python profiles/clock_probe.py cfg3 160 > $OUT/clock_probe_cfg3.md 2> $OUT/clock_probe_cfg3.err
cat $OUT/clock_probe_cfg3.md | cut -c1-260
tail -n 3 $OUT/clock_probe_cfg3.err
for m in idle stream fold4 fold16 fold32 pipeline; do
	timeout 300 python profiles/power_probe.py $m 9 2> $OUT/power_$m.err | tail -n 1 | tee -a $OUT/power_probe.jsonl | cut -c1-400
done
rocm-smi --showclocks --showpower 2>&1 | tail -n 12
B="python bench.py --no-cpu-baseline --no-extra-legs"
timeout 400 $B --workload cfg2 --steps 256 --warmup 32 > $OUT/b256_cfg2.json 2> $OUT/b256_cfg2.err
timeout 400 $B --steps 256 --warmup 32 > $OUT/b256.json 2> $OUT/b256.err
python - <<PY
import json
for n in ("b256_cfg2", "b256"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, "value %.0f ms/step %.4f steady %s demod/blk %s x%s" % (d["value"], d["ms_per_step"], d["steady_state_ms_per_step"], d["demod_kernel_ms_per_block"], d["demod_blocks_per_launch"]), d["streams"]["launches"])
    except Exception as e:
        print(n, "failed", e)
PY
for f in $OUT/*.err; do grep -v "amdgpu.ids\|UserWarning\|dev = torch" $f | tail -n 2; done
