#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lidar" 2>&1 | tail -2
SAG_LIDAR_REG=0 timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lidar" 2>&1 | tail -1
for v in 1 0 1 0; do
SAG_LIDAR_REG=$v timeout -k 10 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b41.json 2> gpurun_out/b41.err; python -c "
import json; r=json.load(open('gpurun_out/b41.json'))
for k,v in r['c2_lidar_cost_only'].items():
  if k!='note': print('REG=$v', k, v['ms_per_call'], v['roofline']['frac'], v['roofline']['kernel_ms'])"
done
