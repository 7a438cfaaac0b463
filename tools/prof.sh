#!/bin/bash
# usage: [BENCH_ARGS="--robot car --task push_box --envs 1048576"] tools/prof.sh <tag>   (on the GPU box, repo root)
# kernel-trace stats + PMC passes (separate runs, as gpurun requires) for bench.py
set -e
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
# FULL=1: the whole default bench (every config: lidar+cost kernel, Car, Doggo, render) instead of the headline only
ARGS="bench.py --steps ${STEPS:-30} --warmup 5 ${FULL:+--cpu-baseline-seconds 1} $([ -z "$FULL" ] && echo --no-c2 --no-cpu-baseline || true) ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $ARGS > /dev/null 2> $OUT/pmc1.err || true
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc2 -- python3 $ARGS > /dev/null 2> $OUT/pmc2.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > /dev/null 2> $OUT/pmc3.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > /dev/null 2> $OUT/pmc4.err || true
# fp64 operation counts (Doggo: flops per env-step for the fp64-vector roofline)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc5 -- python3 $ARGS > /dev/null 2> $OUT/pmc5.err || true
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc6 -- python3 $ARGS > /dev/null 2> $OUT/pmc6.err || true
# FULL=1 also writes profiles/kernels.json (what bench.py checks its roofline blocks against) for summary file $SUMMARY
python3 tools/prof_summary.py $OUT ${FULL:+--json ${SUMMARY:-profiles/${TAG}_summary.txt}} > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
