#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
L=$PWD/safe_adaptation_gym_amd
echo "== car 4M, kernels serialised"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh trace "SAG_OVERLAP=0" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_cl1.so" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_cl2.so" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_cl3.so"
