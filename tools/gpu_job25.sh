#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_cyc1.so timeout -k 10 300 python tools/cycles.py push_box car 4194304 | grep -A16 "== busy"
echo "== car 4M"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh trace default occ1 "SAG_OVERLAP=0 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_occ1.so"
