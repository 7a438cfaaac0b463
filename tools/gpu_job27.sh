#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
echo "== car 4M"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh trace default occ3 "SAG_OVERLAP=0 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_occ3.so"
