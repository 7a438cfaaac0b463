#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 600 python tools/doggo_bench.py 8192 16384 32768 65536 131072
