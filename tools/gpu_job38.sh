#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lidar" 2>&1 | tail -2
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b38.json 2> gpurun_out/b38.err; python -c "
import json; r=json.load(open('gpurun_out/b38.json'))
for k,v in r['c2_lidar_cost_only'].items():
  if k!='note': print(k, v['ms_per_call'], v['roofline']['frac'], v['roofline']['kernel_ms'])
print('c2_4096', r['c2_4096_envs']['ms_per_step'])"
