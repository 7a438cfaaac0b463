#!/bin/bash
# on the GPU box: bench the default library under different environment switches, e.g.
#   tools/ab_env.sh "SAG_BUSY_TWO=0" "SAG_BUSY_TWO=1"
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c2 ${BENCH_ARGS} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: r=json.loads(l)
  except Exception: print(l.rstrip()); continue
  print('value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))
"
done
