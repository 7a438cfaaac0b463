#!/bin/bash
# interleaved A/B (the GPU slows by a few % as a call goes on): tools/abab.sh <rounds> <variant> [<variant> ..]
# ("" = libsag.so); prints one line per run
R=$1; shift
for i in $(seq $R); do
  for v in "" "$@"; do
    lib=safe_adaptation_gym_amd/libsag${v:+_$v}.so
    printf "%-10s " "${v:-default}"
    SAG_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c2 ${BENCH_ARGS} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f  kernel_ms %.4f  frac %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))
"
  done
done
