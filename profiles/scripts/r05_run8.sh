#!/bin/bash
# round 5: the tests added or changed since the last full suite, FFT alone
OUT=/root/repo/gpurun_out/r5h
mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -k "eight_rank_rehearsal_at_full_size or cfg1 or collect_without or end_to_end_small or host_c_program or strict or low_snr or golden" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
cat gpurun_out/r05_eight_rank_cfg3.json
timeout 300 python profiles/fft_accuracy.py > $OUT/fft_accuracy.txt 2>&1; grep "N = " $OUT/fft_accuracy.txt
