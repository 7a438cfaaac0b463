#!/bin/bash
# alternate variants several times (box-to-box and run-to-run noise is +-5 %): tools/ab_run2.sh v1 v2 ...
for rep in 1 2 3; do
for v in "" "$@"; do
  lib=safe_adaptation_gym_amd/libsag${v:+_$v}.so
  SAG_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-c2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('%-8s kernel_ms %.4f' % ('${v:-default}', r['roofline']['kernel_ms']))"
done; done
