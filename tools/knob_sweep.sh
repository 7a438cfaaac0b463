#!/bin/bash
# one-box sweep of the overlap knobs of the split step (Point headline config); run through gpurun
export PYTHONPATH=$PWD:$PWD/tests
tools/ab.sh run -r 2 default "SAG_QUIET_LDS_EXTRA=0" "SAG_QUIET_LDS_EXTRA=2048" "SAG_QUIET_LDS_EXTRA=6144" "SAG_EARLY_FORK=0" "SAG_HOT=0" "SAG_INKERNEL_LIST=1"
