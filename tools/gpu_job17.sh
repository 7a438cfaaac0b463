#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
mkdir -p gpurun_out/diag
for cfg in "car push_box 512 60" "point push_box 512 60"; do
  for v in vref ""; do SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag${v:+_$v}.so python tests/diag_traj.py gpurun_out/diag/t_${v:-new}.npz $cfg | tail -1; done
  python tests/diag_traj.py --cmp gpurun_out/diag/t_vref.npz gpurun_out/diag/t_new.npz
done
BENCH_ARGS="--robot point --task go_to_goal --envs 4194304" tools/ab.sh run -r 3 q4 default
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60 --steps 30" tools/ab.sh run -r 2 vref default
BENCH_ARGS="--robot car --task push_box --envs 1048576 --burn-in 60" tools/ab.sh run -r 2 vref default
