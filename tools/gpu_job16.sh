#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh run -r 2 "" "SAG_QUIET_LDS_EXTRA=2048" "SAG_QUIET_LDS_EXTRA=3072" "SAG_QUIET_LDS_EXTRA=4608" "SAG_QUIET_LDS_EXTRA=6144" "SAG_EARLY_FORK=0"
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60 --steps 30" tools/ab.sh run "" "SAG_QUIET_LDS_EXTRA=0" "SAG_QUIET_LDS_EXTRA=2048" "SAG_EARLY_FORK=0" "SAG_OVERLAP=0"
