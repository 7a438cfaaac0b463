#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
tools/ab.sh run -r 3 default old
