#!/bin/bash
# round-end measurement: parity suite, the default bench line, steady-state traffic passes of the headline and the Car config
export PYTHONPATH=$PWD:$PWD/tests
rm -f gpurun_out/r02_free_running_drift.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gputest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gputest_final.log
STEPS=30 timeout -k 10 500 tools/prof.sh r02_point > gpurun_out/r02_prof_point.log 2>&1
python tools/prof_steady.py gpurun_out/prof_r02_point 30 4194304 point > gpurun_out/r02_point_4M_steady.txt; grep -E "kernel trace|algorithmic|HBM traffic" gpurun_out/r02_point_4M_steady.txt
BENCH_ARGS="--robot car --task push_box --envs 4194304 --burn-in 60" STEPS=20 timeout -k 10 600 tools/prof.sh r02_car > gpurun_out/r02_prof_car.log 2>&1
python tools/prof_steady.py gpurun_out/prof_r02_car 20 4194304 car > gpurun_out/r02_car_4M_steady.txt; grep -E "kernel trace|algorithmic|HBM traffic" gpurun_out/r02_car_4M_steady.txt
cp profiles/traffic.json gpurun_out/traffic_r02.json
# fp64 operation counts of the Doggo physics kernel (flops per env-step for its fp64-vector roofline)
rm -rf /tmp/dgf; TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d /tmp/dgf/pmc5 -- python3 bench.py --no-cpu-baseline --no-c2 --robot doggo --task multitask --envs 4096 --burn-in 20 --steps 20 --warmup 5 > /dev/null 2> /tmp/dgf.err
python3 tools/prof_summary.py /tmp/dgf > gpurun_out/r02_doggo_flops_raw.txt 2>&1; grep -A5 "k_doggo_physics" gpurun_out/r02_doggo_flops_raw.txt | head -6
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_all_configs.json 2> gpurun_out/r02_bench.err; python -c "
import json; r=json.load(open('gpurun_out/r02_bench_all_configs.json'))
print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'])
for k in ('c2_4096_envs','c2_lidar_cost_only','c3_car_push_box','c4_doggo_multitask_4096','c5_doggo_haul_box_rgb_4096','cpu_baseline'):
  print(k, json.dumps(r.get(k))[:600])"
