#!/bin/bash
# one-shot discriminating experiment for DESIGN.md 3.4 (Doggo lane-per-env kernel under a code-shape change)
mkdir -p gpurun_out/diag
export PYTHONPATH=$PWD:$PWD/tests SAG_DOGGO_COOP=0
for v in "" v0 v1 v2 v4; do
  lib=$PWD/safe_adaptation_gym_amd/libsag${v:+_$v}.so
  echo "== ${v:-shipping}"
  SAG_LIB=$lib timeout -k 10 120 python tests/diag_traj.py gpurun_out/diag/gpu_${v:-ship}.npz 30 2>&1 | tail -8
done
unset SAG_DOGGO_COOP
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest2.log
