#!/bin/bash
# A/B measurement of library variants and environment switches (run on the GPU box from the repo root).
#
#   tools/ab.sh build <name> "<-D flags>" [<name> "<flags>" ...]   variants -> safe_adaptation_gym_amd/libsag_<name>.so
#   tools/ab.sh run   [-r ROUNDS] <variant> ...     bench each variant, interleaved ROUNDS times (default 1)
#   tools/ab.sh trace <variant> ...                 steady-state per-kernel times (rocprofv3 kernel trace)
#   tools/ab.sh pmc   <variant> ...                 instruction counters of the step kernel matching $KEY (default quiet)
#
# A <variant> is "" / "default" (libsag.so), the <name> of a built library, or a string of environment
# assignments ("SAG_EARLY_FORK=0 SAG_HOT=1").  Run-to-run spread on one box is a few %, box to box more:
# compare interleaved, in ONE gpurun call.  BENCH_ARGS adds bench.py flags (e.g. "--envs 1048576").
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BASE="--no-cpu-baseline --no-c2 --no-c4 ${BENCH_ARGS}"

variant_env() {  # prints the env assignments that select variant $1
  case "$1" in
    ""|default) ;;
    *=*) echo "$1" ;;
    *) echo "SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_$1.so" ;;
  esac
}
fmt='
import sys, json
for l in sys.stdin:
  try: r = json.loads(l)
  except Exception: print(l.rstrip()); continue
  print("value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]))
'

cmd=$1; shift
case "$cmd" in
  build)
    while [ $# -ge 2 ]; do
      python -m safe_adaptation_gym_amd.build --out libsag_$1.so $2 2>&1 | grep -E "error" || true
      shift 2
    done
    ls -la safe_adaptation_gym_amd/libsag_*.so ;;
  run)
    R=1; if [ "$1" = "-r" ]; then R=$2; shift 2; fi
    for i in $(seq $R); do
      for v in "$@"; do
        printf "%-34s " "${v:-default}"
        env $(variant_env "$v") timeout -k 10 200 python bench.py --steps 100 --warmup 10 $BASE 2>&1 | python -c "$fmt"
      done
    done ;;
  trace)
    for v in "$@"; do
      rm -rf /tmp/abt
      env $(variant_env "$v") rocprofv3 --kernel-trace --output-format csv -d /tmp/abt -- python3 bench.py --steps 40 --warmup 5 $BASE > /dev/null 2>&1
      python3 - "${v:-default}" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/abt/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = []
for key in ('quiet', 'busy', 'phys', 'post', 'compact', 'k_step<'):
  q = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if key in r['Kernel_Name']]
  if q: out.append('%s %.1f us' % (key, sum(q[-20:]) / len(q[-20:]) / 1e3))
print('%-34s' % sys.argv[1], ' | '.join(out))
PY
    done ;;
  pmc)
    for v in "$@"; do
      rm -rf /tmp/abp
      env $(variant_env "$v") rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY} \
        --output-format csv -d /tmp/abp/pmc1 -- python3 bench.py --steps 30 --warmup 5 $BASE > /tmp/abp.out 2>/tmp/abp.err
      echo "== ${v:-default}"; python3 tools/prof_summary.py /tmp/abp > /tmp/abp.sum 2>&1
      grep -A9 "k_step_${KEY:-quiet}<" /tmp/abp.sum | head -10 || { tail -5 /tmp/abp.err; tail -5 /tmp/abp.sum; }
    done ;;
  *) echo "usage: tools/ab.sh build|run|trace|pmc ... (see the header)"; exit 2 ;;
esac
