#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "doggo" > gpurun_out/r02_gputest12.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest12.log
for cfg in "--robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20" "--robot doggo --task multitask --envs 12288 --burn-in 10 --steps 10"; do
  printf "%-70s " "$cfg"
  timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done
FULL=1 STEPS=20 timeout -k 10 900 tools/prof.sh r02_all > gpurun_out/r02_prof_all.log 2>&1
tail -5 gpurun_out/r02_prof_all.log
python tools/prof_steady.py gpurun_out/prof_r02_all 20 4194304 point | tail -12
