#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "doggo" 2>&1 | tail -1
for i in 1 2 3; do printf "doggo 4096: "; timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 --robot doggo --task multitask --envs 4096 --burn-in 20 --steps 30 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"; done
