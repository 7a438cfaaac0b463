"""GPU: the split form WITH the early fork (quiet kernel beside k_compact) against the single full-physics
launch at a batch where the fork is on by default (>= 2 M envs): outputs and final state must be bitwise equal.
  python tools/early_fork_check.py [envs=2500000] [steps=60]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
runs = []
for split in ('0', None):
  if split is None:
    os.environ.pop('SAG_SPLIT', None)
  else:
    os.environ['SAG_SPLIT'] = split
  runs.append(bench.DeviceRun('go_to_goal', n, 0, 0))
for t in range(T):
  for r in runs:
    r.step()
  if t % 10 == 9 or t == T - 1:
    outs = []
    for r in runs:
      r.wait()
      outs.append((r.ctx.dev_download(r.d_obs, (n * 60,), np.float32), r.ctx.dev_download(r.d_rew, (n * 2,), np.float32),
                   r.ctx.dev_download(r.d_cost, (n,), np.uint8), r.ctx.dev_download(r.d_met, (n,), np.uint8)))
    for a, b in zip(*outs):
      assert np.array_equal(a, b), f'outputs differ at step {t}'
    print(f'step {t}: outputs equal; cost rate {outs[0][2].mean():.4f}, busy fraction {runs[1].ctx.busy_count() / n:.4f}', flush=True)
sa, sb = runs[0].ctx.get_state(), runs[1].ctx.get_state()
assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1]), 'final state differs'
print(f'{n} envs x {T} steps: single launch == split form with early fork, bit for bit')
