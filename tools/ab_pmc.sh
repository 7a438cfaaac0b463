#!/bin/bash
export TMPDIR=/tmp
for v in "" "$@"; do
  lib=safe_adaptation_gym_amd/libsag${v:+_$v}.so
  rm -rf /tmp/abp; SAG_LIB=$PWD/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/abp -- python3 bench.py --steps 30 --warmup 5 --no-c2 --no-cpu-baseline > /dev/null 2>&1
  echo "== ${v:-default}"; python3 tools/prof_summary.py /tmp/abp 2>/dev/null | grep -A9 "${KEY:-quiet}" | head -10
done
