#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
L=$PWD/safe_adaptation_gym_amd
mkdir -p gpurun_out/diag
for cfg in "car push_box 2048 150" "car multitask 1024 100"; do
  set -- $cfg
  python tests/diag_traj.py gpurun_out/diag/t_new.npz $1 $2 $3 $4 > /dev/null 2>&1
  SAG_LIB=$L/libsag_ref.so python tests/diag_traj.py gpurun_out/diag/t_ref.npz $1 $2 $3 $4 > /dev/null 2>&1
  printf "%s: " "$cfg"; python tests/diag_traj.py --cmp gpurun_out/diag/t_new.npz gpurun_out/diag/t_ref.npz
done
rm -f gpurun_out/diag/*.npz
echo "== car 4M"; BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh run -r 2 default base
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" tools/ab.sh trace "SAG_OVERLAP=0" "SAG_OVERLAP=0 SAG_LIB=$L/libsag_base.so"
