#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
FULL=1 STEPS=20 timeout -k 10 1000 tools/prof.sh r02_all > gpurun_out/r02_prof_all.log 2>&1
tail -3 gpurun_out/r02_prof_all.log
python tools/prof_steady.py gpurun_out/prof_r02_all 20 4194304 point | tail -14
cp profiles/traffic.json gpurun_out/traffic_r02.json
