#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
echo "== car 4M"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh trace default "SAG_OVERLAP=0"
echo "== point 4M"; tools/ab.sh trace default "SAG_OVERLAP=0"
