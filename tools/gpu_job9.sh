#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputest9.log 2>&1; echo "pytest rc=$?"; tail -16 gpurun_out/r02_gputest9.log
