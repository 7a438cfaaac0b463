"""Doggo: how many constraint rows does a forward evaluation have?  (CPU, oracle only.)  The device kernel gives every
row a lane when an env has at most 32 (two envs per wavefront); this is the histogram that decides how often that holds.
  python tests/study_rows.py [envs=256] [steps=60]"""
import ctypes as C
import sys

import numpy as np

import batch_util as bu
from oracle_lib import Oracle
from safe_adaptation_gym_amd import benchmark

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
o = Oracle()
o.lib.sago_set_threads(8)
NR = 99
hist = (C.c_long * NR)()
for task in ('go_to_goal', 'haul_box', 'multitask'):
  names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=666).train_tasks] if task == 'multitask' else task
  rf, ri = bu.sample_records_native('doggo', names, n, seed=666)
  arr = o.make_batch(rf, ri)
  mt = np.random.RandomState(5)
  o.lib.sago_doggo_row_hist(hist, 1)
  for t in range(T):
    act = mt.uniform(-1, 1, size=(n, 12)).astype(np.float32)
    noise = mt.normal(size=(n, 12)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    o.step_batch_full(arr, 2, act, noise, tape, obs_dim=104)
    if t in (9, T - 1):
      o.lib.sago_doggo_row_hist(hist, 1)
      h = np.array(list(hist), float)
      c = np.cumsum(h) / h.sum()
      print(f'{task:12s} steps..{t + 1:3d}: rows median {int(np.searchsorted(c, .5))} p90 {int(np.searchsorted(c, .9))} p99 {int(np.searchsorted(c, .99))} max {int(np.flatnonzero(h).max())}'
            f' | > 32 rows: {h[33:].sum() / h.sum():.4f} | > 50: {h[51:].sum() / h.sum():.4f} | pair of envs with one > 32 (if independent): {1 - (1 - h[33:].sum() / h.sum()) ** 2:.4f}')
