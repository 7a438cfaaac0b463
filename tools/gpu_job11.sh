#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r02_gputest11.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_gputest11.log
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh run -r 2 default pq3
