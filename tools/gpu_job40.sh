#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
L=$PWD/safe_adaptation_gym_amd
echo "== point 4M, kernels serialised"; tools/ab.sh trace "SAG_OVERLAP=0" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_abl16.so" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_abl128.so" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_abl4.so"
KEY=busy PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" tools/ab.sh pmc default 2>&1 | grep -E "^==|SQ_"
