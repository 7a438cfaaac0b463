"""GPU diagnostic: lockstep device vs oracle (f64 and f32), report the worst mismatches."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import batch_util as bu
from oracle_lib import Oracle
from safe_adaptation_gym_amd import _native as nat

task = sys.argv[1] if len(sys.argv) > 1 else 'go_to_goal_damping'
robot = sys.argv[2] if len(sys.argv) > 2 else 'point'
rid = {'point': 0, 'car': 1}[robot]; od = 60 if rid == 0 else 72
n, T = 192, 160
o64, o32 = Oracle(), Oracle(f32=True)
rf, ri = bu.sample_records(robot, task, n, seed=666)
if task != 'haul_box': rf = bu.goal_beyond_box(rf, ri)
ctx = nat.Context(robot, n, seed=1234)
ctx.set_layout(rf, ri)
rng = np.random.RandomState(7); mt = np.random.RandomState(99)
ctx.observe()
worst = []
for t in range(T):
  rf, ri = ctx.get_state()
  a64, a32 = o64.make_batch(rf, ri), o32.make_batch(rf, ri)
  act = bu.pursuit_actions(rf, ri, rng, robot=robot)
  noise = mt.normal(size=(n, 2)).astype(np.float32)
  tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
  d = ctx.step(act, noise, tape)
  r64 = o64.step_batch_full(a64, rid, act, noise, tape, obs_dim=od)
  r32 = o32.step_batch_full(a32, rid, act, noise, tape, obs_dim=od)
  d_rf, _ = ctx.get_state()
  f64, _ = o64.batch_records(a64)
  f32, _ = o32.batch_records(a32)
  e64 = np.abs(d_rf - f64); e32 = np.abs(d_rf - f32); e6432 = np.abs(f64 - f32)
  rel64 = e64 / (1e-4 + 1e-4 * np.abs(f64)); rel32 = e32 / (2e-5 + 2e-5 * np.abs(f32))
  i, k = np.unravel_index(np.argmax(rel64), e64.shape)
  nb64 = int((rel64.max(1) > 1).sum()); nb32 = int((rel32.max(1) > 1).sum())
  tot64 = globals().get('tot64', 0) + nb64; tot32 = globals().get('tot32', 0) + nb32
  i2, k2 = np.unravel_index(np.argmax(rel32), e32.shape)
  if rel32[i2, k2] > 1 and len(worst) < 25:
    worst.append(('f32', t, int(i2), int(k2), float(d_rf[i2, k2]), float(f64[i2, k2]), float(f32[i2, k2]), nb64, nb32))
  if rel64[i, k] > 1:
    worst.append((t, int(i), int(k), float(d_rf[i, k]), float(f64[i, k]), float(f32[i, k]), nb64, nb32, int(d[2][i]), int(r64[2][i]), int(r32[2][i])))
print('step env field device oracle64 oracle32 nbad64 nbad32 cost_dev cost64 cost32')
for w in worst[:40]:
  print(w)
print(len(worst), 'steps with violations; total rows out of tol: fp64', tot64, 'fp32', tot32)
