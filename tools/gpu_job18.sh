#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "car" > gpurun_out/r02_gputest18.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_gputest18.log
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" STEPS=20 timeout -k 10 600 tools/prof.sh r02_car > gpurun_out/r02_prof_car.log 2>&1
python tools/prof_steady.py gpurun_out/prof_r02_car 20 4194304 car > gpurun_out/r02_car_4M_steady.txt; grep -E "kernel trace|algorithmic|HBM traffic" gpurun_out/r02_car_4M_steady.txt
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "doggo" > gpurun_out/r02_gputest18b.log 2>&1; echo "pytest doggo rc=$?"; tail -2 gpurun_out/r02_gputest18b.log
printf "doggo 4096: "; timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 --robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
