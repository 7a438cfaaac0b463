#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
L=$PWD/safe_adaptation_gym_amd
export BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60"
KEY=busy PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD" tools/ab.sh pmc default cl1 cl2 cl4 2>&1 | grep -E "^==|k_step_busy|SQ_"
tools/ab.sh trace "SAG_OVERLAP=0" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_cl4.so"
