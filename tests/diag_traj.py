"""Free-running trajectory dump for comparing library builds bit for bit (a refactor that must not change
results; a code-shape change that should not: DESIGN.md 3.4):
  SAG_LIB=... python tests/diag_traj.py out.npz [robot=doggo] [task=go_to_goal] [n=64] [T=30] [--replay base.npz]
--replay: every step starts from the state base.npz had there (one-step comparison of builds whose arithmetic differs
by rounding, e.g. the 64-lane PGS path forced by -DSAG_DC_FAST_ROWS=8: tests/test_hostemu_variants.py).
Doggo: random torques (the scenario of test_doggo_lockstep_vs_oracle); Point / Car: pursuit actions so that goals,
vases and the box are actually hit.  `python tests/diag_traj.py --cmp a.npz b.npz` prints the first difference.
Uses no oracle."""
import sys

import numpy as np

import batch_util as bu
from safe_adaptation_gym_amd import _native as nat

if sys.argv[1] == '--cmp':
  a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
  same = True
  for key in ('states', 'outs'):
    A, B = a[key], b[key]
    eq = (A == B) | (np.isnan(A) & np.isnan(B))
    if eq.all():
      continue
    same = False
    t, e, f = np.argwhere(~eq)[0]
    print(f'{key}: first difference at step {t}, env {e}, field {f}: {A[t, e, f]!r} vs {B[t, e, f]!r}; '
          f'{(~eq).sum()} differing values, max |diff| {np.nanmax(np.abs(A - B)):.3g}')
  print('bit-identical' if same else 'DIFFERENT')
  sys.exit(0 if same else 1)

replay = None
if '--replay' in sys.argv:
  k = sys.argv.index('--replay')
  replay = np.load(sys.argv[k + 1])
  del sys.argv[k:k + 2]
out = sys.argv[1]
robot = sys.argv[2] if len(sys.argv) > 2 else 'doggo'
task = sys.argv[3] if len(sys.argv) > 3 else 'go_to_goal'
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
T = int(sys.argv[5]) if len(sys.argv) > 5 else 30
nu = nat.robot_info(robot)['nu']
rf, ri = bu.sample_records_native(robot, task, n, seed=666)
ctx = nat.Context(robot, n, seed=4321)
ctx.set_layout(rf, ri)
mt, rng = np.random.RandomState(5), np.random.RandomState(6)
states, ints, outs = [], [], []
for t in range(T):
  if replay is not None and t > 0:
    ctx.set_state(replay['states'][t - 1], replay['ints'][t - 1])
  if robot == 'doggo':
    act = mt.uniform(-1, 1, size=(n, nu)).astype(np.float32)
    if t < 3:
      act[:] = 0
  else:
    act = bu.pursuit_actions(ctx.get_state()[0], ri, rng, robot=robot)
  noise = mt.normal(size=(n, nu)).astype(np.float32)
  tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
  o = ctx.step(act, noise, tape)
  s = ctx.get_state()
  states.append(s[0].copy())
  ints.append(s[1].copy())
  outs.append(np.concatenate([o[0], o[1], o[2][:, None], o[3][:, None], o[4][:, None]], 1))
  bad = np.flatnonzero(~np.isfinite(s[0]).all(1))
  if len(bad):
    print(f'step {t}: non-finite envs {bad.tolist()}', flush=True)
np.savez_compressed(out, states=np.stack(states), ints=np.stack(ints), outs=np.stack(outs), rec0=rf, reci0=ri)
print('saved', out, 'cost rate', float(np.mean([o[:, -3].mean() for o in outs])), 'goals met', int(sum(o[:, -1].sum() for o in outs)))
