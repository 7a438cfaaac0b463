"""GPU: front-end cycle at a batch large enough for the split launch form."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import safe_adaptation_gym_amd as sag
n = 300000
t0 = time.time()
env = sag.make('point', 'go_to_goal', seed=1, n_envs=n)
obs = env.reset()
print('make+reset %.2fs' % (time.time() - t0), obs.shape, flush=True)
rng = np.random.RandomState(0)
tot_cost = 0
for t in range(30):
  obs, rew, done, info = env.step(rng.uniform(-1, 1, (n, 2)).astype(np.float32))
  tot_cost += info['cost'].sum()
assert np.isfinite(obs).all() and not done.any()
print('30 steps ok; cost events', int(tot_cost), 'goal_met', int(info['goal_met'].sum()), flush=True)
obs = env.reset()
obs, rew, done, info = env.step(np.zeros((n, 2), np.float32))
assert np.isfinite(obs).all()
print('reset + step ok', flush=True)
env.close()
