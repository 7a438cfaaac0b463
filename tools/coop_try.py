"""GPU: cooperative vs lane-per-env Doggo physics from the same state, field by field."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import batch_util as bu
from safe_adaptation_gym_amd import _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
task = sys.argv[2] if len(sys.argv) > 2 else 'go_to_goal'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
if task == 'multitask':
  from safe_adaptation_gym_amd import benchmark
  task = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=666).train_tasks]
rf, ri = bu.sample_records_native('doggo', task, n, seed=666)
ctxs = []
for flag in ('0', '1'):
  os.environ['SAG_DOGGO_COOP'] = flag
  c = nat.Context('doggo', n, seed=1)
  c.set_layout(rf, ri)
  ctxs.append(c)
rng = np.random.RandomState(0)
for t in range(steps):
  s_rf, s_ri = ctxs[0].get_state()
  ctxs[1].set_state(s_rf, s_ri)
  act = rng.uniform(-1, 1, (n, 12)).astype(np.float32) if t >= 2 else np.zeros((n, 12), np.float32)
  noise = rng.normal(size=(n, 12)).astype(np.float32)
  oa = ctxs[0].step(act, noise)
  ob = ctxs[1].step(act, noise)
  a, _ = ctxs[0].get_state(); b, _ = ctxs[1].get_state()
  d = np.abs(a - b)
  bad = np.argwhere(d > 1e-5 + 1e-4 * np.abs(a))
  print(f'step {t}: max state diff {d.max():.3e}; obs diff {np.abs(oa[0] - ob[0]).max():.3e}; rew diff {np.abs(oa[1] - ob[1]).max():.3e};',
        'bad envs', sorted(set(bad[:, 0]))[:16], 'fields', sorted(set(bad[:, 1]))[:24], flush=True)
