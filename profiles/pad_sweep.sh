#!/bin/bash
cd /root/repo
run() { echo "== $*"; env "$@" REPS=3 python profiles/recreate_probe.py 2>&1 | grep -E "^\[" | sed -e 's/"rep": [0-9], //g'; }
run X=base
run HFDL_EXP_SLICES=4
run HFDL_EXP_SLICES=2
run HFDL_EXP_FOLD_CS=4
run HFDL_EXP_FOLD_CS=4 HFDL_EXP_SLICES=4
run HFDL_EXP_FOLD_CS=4 HFDL_EXP_SLICES=2
run HFDL_EXP_FOLD_CS=8 HFDL_EXP_SLICES=2
run HFDL_EXP_FOLD_CS=8 HFDL_EXP_SLICES=4
run HFDL_EXP_FOLD_CS=1 HFDL_EXP_SLICES=8
run HFDL_EXP_FOLD_CS=1 HFDL_EXP_SLICES=16
run X=base
