"""GPU: where do the split (quiet + busy) and the single-launch form differ?  (must be nowhere)"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import batch_util as bu
from safe_adaptation_gym_amd import _native as nat
robot, task, n = 'point', sys.argv[1] if len(sys.argv) > 1 else 'go_to_goal', 1500
rf, ri = bu.sample_records_native(robot, task, n, seed=4000)
ctxs = []
for flag in ('0', '1'):
  os.environ['SAG_SPLIT'] = flag
  c = nat.Context(robot, n, seed=77); c.set_layout(rf, ri); ctxs.append(c)
rng = np.random.RandomState(3)
for t in range(4):
  s_rf, s_ri = ctxs[0].get_state()
  act = bu.pursuit_actions(s_rf, s_ri, rng, robot=robot)
  outs = [c.step(act) for c in ctxs]
  d = outs[0][0] != outs[1][0]
  envs = np.argwhere(d.any(1)).ravel()
  print(f'step {t}: {d.sum()} obs elements differ in {len(envs)} envs; envs {envs[:10]}; columns {sorted(set(np.argwhere(d)[:, 1]))[:30]}')
  if len(envs):
    e = envs[0]; cols = np.argwhere(d[e]).ravel()
    print('   env', e, 'cols', cols[:12], 'single', outs[0][0][e, cols[:6]], 'split', outs[1][0][e, cols[:6]])
  sa, sb = ctxs[0].get_state(), ctxs[1].get_state()
  print('   state differs in', int((sa[0] != sb[0]).any(1).sum()), 'envs')
