#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh run -r 2 base default carw1
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh run -r 2 base default
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh trace base default carw1
