#!/bin/bash
# PMC passes over the Doggo physics kernel (4096 envs)
export PYTHONPATH=$PWD:$PWD/tests TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-c2 --robot doggo --task multitask --envs 4096 --burn-in 10 --steps 10 --warmup 2"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/dpmc; mkdir -p /tmp/dpmc
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/dpmc/pmc1 -- python3 bench.py $ARGS > /tmp/dpmc.out 2> /tmp/dpmc.err || tail -3 /tmp/dpmc.err
  python3 tools/prof_summary.py /tmp/dpmc 2>&1 | grep -A10 "doggo_physics" | head -11
done
