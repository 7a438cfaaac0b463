#!/bin/bash
# builds variants of libsag.so with extra -D flags:  tools/ab_build.sh name "-DX=1 -DY=2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-fast-math -ffp-contract=off -Wno-unused-function $2 -pthread -o safe_adaptation_gym_amd/libsag_$1.so safe_adaptation_gym_amd/csrc/sag_api.hip safe_adaptation_gym_amd/csrc/sag_sampler.cpp 2>&1 | grep -E "error" || true
  shift 2
done
ls -la safe_adaptation_gym_amd/libsag_*.so
