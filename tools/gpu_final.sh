#!/bin/bash
# round-end measurement (run on the GPU box from the repo root, one gpurun call each for the two halves if time is short):
#   1. the parity suite;  2. FULL profile of the default bench.py (kernel trace + PMC passes) -> profiles/r04_all_summary.txt
#   + profiles/kernels.json;  3. steady-state traffic passes of the headline and the Car config -> profiles/traffic.json;
#   4. the default bench line, now checked against 2.
# STAGE=1: steps 1-2 and the headline's traffic; STAGE=2: the Car's traffic and step 4 (copy gpurun_out/${R}_kernels.json and
# gpurun_out/traffic_${R}.json of stage 1 into profiles/ first: a gpurun call has 20 minutes and only gpurun_out/ comes back).
export PYTHONPATH=$PWD:$PWD/tests
R=${ROUND:-r04}
if [ "${STAGE:-1}" != 2 ]; then
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${R}_gputest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${R}_gputest_final.log
fi
FULL=1 SUMMARY=profiles/${R}_all_summary.txt STEPS=30 timeout -k 10 1000 tools/prof.sh ${R}_all > gpurun_out/${R}_prof_all.log 2>&1
cp gpurun_out/prof_${R}_all/summary.txt gpurun_out/${R}_all_summary.txt; cp profiles/kernels.json gpurun_out/${R}_kernels.json
grep -E "k_lidar_cost_team<4>|k_doggo_physics|wrote" gpurun_out/${R}_all_summary.txt | head -8
python tools/prof_steady.py gpurun_out/prof_${R}_all 30 4194304 point > gpurun_out/${R}_point_4M_steady.txt; grep -E "kernel trace|algorithmic|HBM traffic" gpurun_out/${R}_point_4M_steady.txt
cp profiles/traffic.json gpurun_out/traffic_${R}.json
fi
if [ "$STAGE" = 1 ]; then exit 0; fi
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60 --no-c4" STEPS=20 timeout -k 10 600 tools/prof.sh ${R}_car > gpurun_out/${R}_prof_car.log 2>&1
python tools/prof_steady.py gpurun_out/prof_${R}_car 20 4194304 car > gpurun_out/${R}_car_4M_steady.txt; grep -E "kernel trace|algorithmic|HBM traffic" gpurun_out/${R}_car_4M_steady.txt
cp profiles/traffic.json gpurun_out/traffic_${R}.json
timeout -k 10 600 python bench.py > gpurun_out/${R}_bench_all_configs.json 2> gpurun_out/${R}_bench.err; python -c "
import json; r=json.load(open('gpurun_out/${R}_bench_all_configs.json'))
print('headline', r['value'], r['ms_per_step'], r['roofline'].get('frac'), r['roofline'].get('traffic'), r['roofline'].get('profile') is not None)
for k in ('c2_4096_envs','c2_lidar_cost_only','c3_car_push_box','c4_doggo_multitask','c5_doggo_haul_box_rgb_4096','cpu_baseline'):
  print(k, json.dumps(r.get(k))[:500])"
