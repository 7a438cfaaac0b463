#!/bin/bash
# small-batch experiment: split form with few envs per busy wavefront at the BASELINE config-2 batch (4096 envs)
export PYTHONPATH=$PWD:$PWD/tests
for n in 4096 16384 65536; do
echo "== envs $n"
BENCH_ARGS="--robot point --task go_to_goal --envs $n --steps 300" tools/ab.sh run "" "SAG_SPLIT=1" "SAG_SPLIT=1 SAG_OVERLAP=0" "SAG_SPLIT=1 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_be16.so" "SAG_SPLIT=1 SAG_OVERLAP=0 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_be16.so" "SAG_SPLIT=1 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_be8.so" "SAG_SPLIT=1 SAG_OVERLAP=0 SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_be8.so"
done
