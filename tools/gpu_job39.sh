#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
for e in 0 16 8 4 2; do
  printf "SAG_EPW=%s: " $e
  SAG_EPW=$e timeout -k 10 200 python bench.py --no-cpu-baseline --no-c2 --envs 4096 --steps 300 --warmup 30 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done
for e in 0 8 4; do
  printf "car SAG_EPW=%s: " $e
  SAG_EPW=$e timeout -k 10 200 python bench.py --no-cpu-baseline --no-c2 --robot car --task push_box --envs 4096 --steps 300 --warmup 30 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done
