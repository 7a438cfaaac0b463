#!/bin/bash
# A/B: SGPR spills to memory (-mllvm -amdgpu-spill-sgpr-to-vgpr=false) vs the default, all configs
for r in 1 2; do
for v in "" nosv; do
  lib=$PWD/safe_adaptation_gym_amd/libsag${v:+_$v}.so
  for cfg in "--robot point --task go_to_goal --envs 4194304" "--robot car --task push_box --envs 1048576 --burn-in 60" "--robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20" "--robot doggo --task multitask --envs 32768 --burn-in 5 --steps 5"; do
    printf "%-8s %-70s " "${v:-default}" "$cfg"
    SAG_LIB=$lib timeout -k 10 300 python bench.py --warmup 5 --no-cpu-baseline --no-c2 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
done
