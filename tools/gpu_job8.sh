#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh run -r 2 default q3 q3b32 q3b16 q3u
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh trace default q3 q3b32 q3b16 q3u
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" KEY=busy tools/ab.sh pmc q3
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" KEY=quiet tools/ab.sh pmc q3
