#!/bin/bash
# round-2 first GPU call: parity suite, then rocprof passes of the Car and Doggo configs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest1.log
tail -3 gpurun_out/r02_gputest1.log
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" STEPS=20 timeout -k 10 500 tools/prof.sh r02_car_1M > gpurun_out/r02_prof_car.log 2>&1
tail -40 gpurun_out/prof_r02_car_1M/summary.txt
BENCH_ARGS="--robot doggo --task multitask --envs 4096 --burn-in 20" STEPS=10 timeout -k 10 500 tools/prof.sh r02_doggo_4096 > gpurun_out/r02_prof_doggo.log 2>&1
tail -40 gpurun_out/prof_r02_doggo_4096/summary.txt
