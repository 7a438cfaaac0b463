#!/bin/bash
# steady-state traffic (FETCH_SIZE / WRITE_SIZE passes) + kernel trace per config, each in a clean single-config run
export PYTHONPATH=$PWD:$PWD/tests
STEPS=30 timeout -k 10 500 tools/prof.sh r02_point > gpurun_out/r02_prof_point.log 2>&1
python tools/prof_steady.py gpurun_out/prof_r02_point 30 4194304 point | grep -E "kernel trace|algorithmic|HBM traffic|merged"
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" STEPS=20 timeout -k 10 600 tools/prof.sh r02_car > gpurun_out/r02_prof_car.log 2>&1
python tools/prof_steady.py gpurun_out/prof_r02_car 20 4194304 car | grep -E "kernel trace|algorithmic|HBM traffic|merged"
cp profiles/traffic.json gpurun_out/traffic_r02.json
timeout -k 10 300 python -m pytest tests -m gpu -q -k "free_running_drift" > gpurun_out/r02_drift.log 2>&1; tail -2 gpurun_out/r02_drift.log
