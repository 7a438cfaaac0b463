#!/bin/bash
# per-kernel steady-state times (rocprofv3 kernel trace) for library variants
export TMPDIR=/tmp
for v in "" "$@"; do
  lib=safe_adaptation_gym_amd/libsag${v:+_$v}.so
  rm -rf /tmp/abt; SAG_LIB=$PWD/$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/abt -- python3 bench.py --steps 40 --warmup 5 --no-c2 --no-cpu-baseline > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/abt/**/*kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
out=[]
for key in ('quiet','busy','phys','post','compact','k_step<'):
    q=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows if key in r['Kernel_Name']]
    if q: out.append('%s %.1f us' % (key, sum(q[-20:])/20/1e3))
print('%-10s' % (sys.argv[1] or 'default'), ' | '.join(out))
PY
done
