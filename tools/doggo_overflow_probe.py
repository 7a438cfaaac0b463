"""GPU: which Doggo envs of the bench's multitask batch run out of constraint rows (record flag bit 2), and what are they
touching?  Steps the batch like bench.py does, then lists the oracle's contacts of the flagged envs' states.
  python tools/doggo_overflow_probe.py [envs=4096] [steps=30] [task=multitask]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
import oracle_lib as ol  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
task = sys.argv[3] if len(sys.argv) > 3 else 'multitask'
r = bench.DeviceRun(task, n, 0, 0, robot='doggo')
o = ol.Oracle()
seen = {}
for t in range(T):
  r.step(); r.wait()
  rf, ri = r.ctx.get_state()
  for e in np.flatnonzero(ri[:, 13] & 4):
    seen.setdefault(int(e), []).append(t)
print(f'{task} {n} envs x {T} steps: envs that overflowed the row budget at some step: {len(seen)}')
rf, ri = r.ctx.get_state()
for e, steps in list(seen.items())[:6]:
  rows, cc = o.doggo_contacts(o.env(rf[e].astype(np.float64), ri[e]))
  print(f'env {e} task id {ri[e, 0]} overflow at steps {steps[:12]}{"..." if len(steps) > 12 else ""}; now: z {rf[e, 144]:.3f} quat {np.round(rf[e, 145:149], 3)} '
        f'box {np.round(rf[e, 41:44], 3)} robot {np.round(rf[e, 0:3], 3)} | oracle contacts now {len(rows)} (cost rule {cc})')
  for row in rows:
    k = int(row[0])
    what = f'floor {(k - 0x1000) // 4}' if k < 0x10000 else 'obj %d geom %d contact %d robot geom %d' % (
        (k - 0x10000) // 4 // 2048, (k - 0x10000) // 4 // 256 % 8, (k - 0x10000) // 4 // 32 % 8, (k - 0x10000) // 4 % 32)
    print(f'     {what:40s} p {np.round(row[1:4], 3)} n {np.round(row[4:7], 2)} depth {row[7]:.4f} f {row[8]:.4f}')
if seen:
  ids = np.array(list(seen)[:64])
  np.savez(os.path.join(ROOT, 'gpurun_out', 'r04_heavy_envs.npz'), rf=rf[ids], ri=ri[ids], ids=ids)
r.close()
