#!/bin/bash
mkdir -p gpurun_out/diag
export PYTHONPATH=$PWD:$PWD/tests
SAG_DOGGO_COOP=0 timeout -k 10 120 python tests/diag_traj.py gpurun_out/diag/gpu_ship2.npz 30 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest4.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest4.log
for cfg in "--robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20" "--robot doggo --task multitask --envs 32768 --burn-in 5 --steps 5"; do
  printf "%-70s " "$cfg"
  timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done
