#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gputest29.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest29.log
tools/ab.sh run -r 3 default base
