"""Diagnosis of the Doggo lane-per-env kernel under a code-shape change: runs the scenario of
test_doggo_lockstep_vs_oracle[go_to_goal-0] (64 envs, seed 666, lane-per-env form) free-running for T steps with the
library named by SAG_LIB and stores the state after every step:
  SAG_LIB=... SAG_DOGGO_COOP=0 python tests/diag_doggo_variant.py out.npz [T]
Comparing the files of two libraries (or of the host sanitizer build, tests/hostemu) shows the first step / env /
field at which they part.  Uses no oracle."""
import sys

import numpy as np

import batch_util as bu
from safe_adaptation_gym_amd import _native as nat

out, T = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 64
rf, ri = bu.sample_records_native('doggo', 'go_to_goal', n, seed=666)
ctx = nat.Context('doggo', n, seed=4321)
ctx.set_layout(rf, ri)
mt = np.random.RandomState(5)
states, outs = [], []
for t in range(T):
  act = mt.uniform(-1, 1, size=(n, 12)).astype(np.float32)
  if t < 3:
    act[:] = 0
  noise = mt.normal(size=(n, 12)).astype(np.float32)
  tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
  o = ctx.step(act, noise, tape)
  s = ctx.get_state()
  states.append(s[0].copy())
  outs.append(np.concatenate([o[0], o[1], o[2][:, None], o[3][:, None]], 1))
  bad = np.flatnonzero(~np.isfinite(s[0]).all(1))
  if len(bad):
    print(f'step {t}: non-finite envs {bad.tolist()}', flush=True)
np.savez_compressed(out, states=np.stack(states), outs=np.stack(outs), rec0=rf, reci0=ri)
print('saved', out)
