#!/bin/bash
# interleaved comparison of environment switches: tools/abab_env.sh <rounds> "A=1" "A=2 B=3" ...
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    printf "%-34s " "$v"
    env $v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c2 ${BENCH_ARGS} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('ms/step %.4f  kernel_ms %.4f  frac %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))
"
  done
done
