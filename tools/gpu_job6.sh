#!/bin/bash
mkdir -p gpurun_out/diag
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_gputest6.log
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh run -r 2 base default
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60 --steps 30" tools/ab.sh run default
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh run -r 2 base default
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh trace base default
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh trace base default
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench6.json 2> gpurun_out/r02_bench6.err; tail -c 3000 gpurun_out/r02_bench6.json
