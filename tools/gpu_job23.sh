#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
export SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_cyc.so
timeout -k 10 300 python tools/cycles.py push_box car 4194304
timeout -k 10 300 python tools/cycles.py go_to_goal point 4194304
