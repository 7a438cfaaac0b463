#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_cyc.so timeout -k 10 300 python tools/cycles_doggo.py 4096
