"""Counters of the slowest against the fastest context of a tools/regime_probe.py run under `rocprofv3 --pmc` (one
directory per pass).  Dispatches are attributed to contexts by order: every context launches the same number of
k_step_quiet kernels (burn-in 200 + K); the last K of each are averaged (duration from the dispatch timestamps).
  python tools/regime_diff.py <dir with pass*/ subdirectories> [K=60] [steps per context=260]"""
import csv, glob, os, sys
from collections import defaultdict
import numpy as np

root = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
per_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 200 + K
for pdir in sorted(glob.glob(os.path.join(root, 'pass*'))):
  files = glob.glob(os.path.join(pdir, '**', '*counter_collection.csv'), recursive=True)
  if not files:
    continue
  rows = defaultdict(dict)   # (kernel kind, dispatch id) -> {counter: value, 'dur': ns}
  for r in csv.DictReader(open(files[0])):
    kind = 'quiet' if 'k_step_quiet' in r['Kernel_Name'] else ('busy' if 'k_step_busy' in r['Kernel_Name'] else None)
    if kind is None:
      continue
    d = rows[(kind, int(r['Dispatch_Id']))]
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    d['duration_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  print(f'== {os.path.basename(pdir)}')
  for kind in ('quiet', 'busy'):
    ids = sorted(i for k, i in rows if k == kind)
    nctx = len(ids) // per_ctx
    if nctx < 2:
      continue
    ctx = []
    for c in range(nctx):
      sel = ids[c * per_ctx + per_ctx - K:(c + 1) * per_ctx]
      names = sorted(rows[(kind, sel[0])])
      ctx.append({n: float(np.mean([rows[(kind, i)][n] for i in sel])) for n in names})
    order = np.argsort([c['duration_us'] for c in ctx])
    fast, slow = ctx[order[0]], ctx[order[-1]]
    print(f'  k_step_{kind}: {nctx} contexts, kernel duration per context (us): ' + ' '.join(f"{c['duration_us']:.0f}" for c in ctx))
    print(f'    {"counter (mean per dispatch)":44s} {"fastest ctx":>14s} {"slowest ctx":>14s} {"slow / fast":>11s}')
    for n in sorted(fast):
      print(f'    {n:44s} {fast[n]:14.5g} {slow[n]:14.5g} {slow[n] / fast[n] if fast[n] else float("nan"):11.3f}')
