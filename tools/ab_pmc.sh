#!/bin/bash
# instruction counters of the step kernels for library variants (own PMC pass per variant)
export TMPDIR=/tmp
for v in "" "$@"; do
  lib=safe_adaptation_gym_amd/libsag${v:+_$v}.so
  rm -rf /tmp/abp; SAG_LIB=$PWD/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/abp/pmc1 -- python3 bench.py --steps 30 --warmup 5 --no-c2 --no-cpu-baseline > /tmp/abp.out 2>/tmp/abp.err
  echo "== ${v:-default}"; python3 tools/prof_summary.py /tmp/abp > /tmp/abp.sum 2>&1
  grep -A9 "k_step_${KEY:-quiet}<0" /tmp/abp.sum | head -10 || { tail -5 /tmp/abp.err; tail -5 /tmp/abp.sum; }
done
