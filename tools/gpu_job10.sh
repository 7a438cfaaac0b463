#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "doggo or smoke" > gpurun_out/r02_gputest10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_gputest10.log
for cfg in "--robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20" "--robot doggo --task multitask --envs 12288 --burn-in 10 --steps 10"; do
  for v in base ""; do
  printf "%-8s %-70s " "${v:-new}" "$cfg"
  SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag${v:+_$v}.so timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
