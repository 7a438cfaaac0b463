"""How far is the specification's ONE Gauss-Seidel sweep per forward evaluation from the converged solution of the
same constraint set?    CPU only, oracle only (this is a study of the
specification, not a device test):

  python tests/study_sweeps.py [envs=512] [steps=200]  ->  table on stdout (+ profiles/r02_sweep_convergence.txt)

For each bench workload the 1-sweep oracle free-runs (pursuit actions, so that vases, the box and goals are hit).
At every step the SAME pre-step state is also advanced with N sweeps (N = 2, 4, 16, 64; 64 is converged: the
16 -> 64 difference is reported as the yardstick) and the post-step states / flags are compared:
  * robot qpos / qvel distance after one env-step (all env-steps, and the env-steps with any constraint active),
  * cost-flag and goal-met disagreement rate per env-step.
Doggo: its robot rows already run 4 sweeps (DG_PGS_ITERS); N replaces that 4."""
import os
import sys

import numpy as np

import batch_util as bu
from oracle_lib import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
o = Oracle()
import ctypes as C
o.lib.sago_set_sweeps.argtypes = [C.c_int]
lines = []


def say(s):
  print(s, flush=True)
  lines.append(s)


say(f'# one sweep vs N sweeps of the same constraint set, one env-step from identical state; {T} steps free-running on the 1-sweep model')
for robot, task, nn in (('point', 'go_to_goal', n), ('car', 'push_box', n), ('doggo', 'go_to_goal', max(n // 8, 32))):
  rid = {'point': 0, 'car': 1, 'doggo': 2}[robot]
  nu, od = {'point': 2, 'car': 2, 'doggo': 12}[robot], {'point': 60, 'car': 72, 'doggo': 104}[robot]
  rf, ri = bu.sample_records_native(robot, task, nn, seed=777)
  arr = o.make_batch(rf, ri)
  rng, mt = np.random.RandomState(1), np.random.RandomState(2)
  NS = [2, 4, 16, 64] if robot != 'doggo' else [1, 16, 64]
  stat = {N: dict(dq=[], dv=[], cost=0, met=0) for N in NS}
  active_steps = total = 0
  for t in range(T):
    rf, ri = o.batch_records(arr)
    act = mt.uniform(-1, 1, size=(nn, nu)).astype(np.float32) if robot == 'doggo' else bu.pursuit_actions(rf, ri, rng, robot=robot)
    noise = mt.normal(size=(nn, nu)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(nn, 64), dtype=np.uint32)
    outs = {}
    for N in NS:
      o.lib.sago_set_sweeps(N)
      a2 = o.make_batch(rf, ri)
      r = o.step_batch_full(a2, rid, act, noise, tape, obs_dim=od)
      outs[N] = (o.batch_records(a2)[0], r[2].copy(), r[4].copy())
    o.lib.sago_set_sweeps(1)
    r1 = o.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    s1 = o.batch_records(arr)[0]
    E = 144
    qcols = [0, 1, 2] + ([E] + list(range(E + 1, E + 5)) + list(range(E + 9, E + 22)) if robot == 'doggo' else [])
    vcols = [3, 4, 5] + ([E + 5, E + 6, E + 7, E + 8] + list(range(E + 22, E + 35)) if robot == 'doggo' else ([E, E + 1] if robot == 'car' else []))
    total += nn
    for N in NS:
      sN, cN, mN = outs[N]
      stat[N]['dq'].append(np.abs(sN[:, qcols] - s1[:, qcols]).max(1))
      stat[N]['dv'].append(np.abs(sN[:, vcols] - s1[:, vcols]).max(1))
      stat[N]['cost'] += int((cN != r1[2]).sum())
      stat[N]['met'] += int((mN != r1[4]).sum())
  say(f'## {robot}/{task}: {nn} envs x {T} steps = {total} env-steps; 1-sweep cost rate {float(np.mean(r1[2])):.3f} (last step)')
  for N in NS:
    dq, dv = np.concatenate(stat[N]['dq']), np.concatenate(stat[N]['dv'])
    act_ = dq > 0
    say(f'  N={N:3d}: env-steps that differ at all {act_.mean():.3f} | qpos diff median/p99/max over those '
        f'{np.median(dq[act_]) if act_.any() else 0:.2e}/{np.quantile(dq[act_], .99) if act_.any() else 0:.2e}/{dq.max():.2e} | '
        f'qvel diff median/p99/max {np.median(dv[act_]) if act_.any() else 0:.2e}/{np.quantile(dv[act_], .99) if act_.any() else 0:.2e}/{dv.max():.2e} | '
        f'cost-flag disagreement {stat[N]["cost"] / total:.2e} | goal-met disagreement {stat[N]["met"] / total:.2e}')
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'r02_sweep_convergence.txt')
open(out, 'w').write('\n'.join(lines) + '\n')
