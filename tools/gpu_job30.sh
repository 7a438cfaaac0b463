#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gputest30.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest30.log
echo "== car 4M"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh run -r 2 default base
echo "== point 4M"; tools/ab.sh run -r 2 default base
