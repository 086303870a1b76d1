cd /root/repo
mkdir -p gpurun_out/r2soak
for spec in "cfg3 6000" "cfg4 6000" "cfg2 30000"; do
set -- $spec
timeout 600 python bench.py --workload $1 --steps $2 --no-cpu-baseline --no-extra-legs 2>/dev/null > gpurun_out/r2soak/soak_$1.json
python -c "import sys,json; r=json.load(open('gpurun_out/r2soak/soak_$1.json')); print('$1', r['steps'], round(r['value'],1), r['pdus_in_timed_region'], r['pdus_matching_sent_payload'], r['pdus_rank0_fcs_good_on_device'], round(r['roofline']['frac'],4))"
done
SOAK_S=75 timeout 600 python profiles/noise_soak.py 2>/dev/null | tail -1
