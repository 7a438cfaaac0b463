"""How far is the specification's constraint solve from the CONVERGED solution of the same constraint set?
  CPU only, oracle only: a study of the specification, not a device test.

  python tests/study_sweeps.py [envs=512] [steps=120]  ->  table on stdout + profiles/r04_sweep_convergence.txt

For each bench workload the SPECIFICATION free-runs (pursuit actions for Point / Car so that vases, the box and goals
are hit; random torques for the Doggo).  At every step the SAME pre-step state is also advanced with N cold
projected-Gauss-Seidel sweeps over every constraint (`sago_set_sweeps(N)`: accumulated forces; N = 256 is the
converged reference, 64 -> 256 is reported as the yardstick) and post-step states / flags are compared:
  * robot qpos / qvel distance after one env-step (max over the coordinates), over ALL env-steps,
  * cost-flag and goal-met disagreement rate per env-step.
Specification (round 3): Point - one sweep; Car - floor friction solved to convergence (direct solve + 4 sweeps over
5 merged elements, before and after the contacts), contacts one sweep; Doggo - warm-started PGS, 24 sweeps (48 cold).
Also listed: the round-2 specification of the Doggo (4 cold sweeps) and what more Doggo sweeps would buy."""
import ctypes as C
import os
import sys

import numpy as np

import batch_util as bu
from oracle_lib import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
o = Oracle()
L = o.lib
L.sago_set_sweeps.argtypes = [C.c_int]
L.sago_set_doggo_solver.argtypes = [C.c_int] * 3
L.sago_set_threads(8)
lines = []


def say(s):
  print(s, flush=True)
  lines.append(s)


def cfg(sweeps=1, dg=(24, 48, 1)):
  def f():
    L.sago_set_sweeps(sweeps)
    L.sago_set_doggo_solver(*dg)
  return f


SPEC = cfg()
say(f'# specification vs N cold sweeps over the same constraint set, one env-step from identical state; {T} steps free-running on the specification')
for robot, task, nn in (('point', 'go_to_goal', n), ('car', 'push_box', n), ('doggo', 'go_to_goal', max(n // 4, 32))):
  rid = {'point': 0, 'car': 1, 'doggo': 2}[robot]
  nu, od = {'point': 2, 'car': 2, 'doggo': 12}[robot], {'point': 60, 'car': 72, 'doggo': 104}[robot]
  rf, ri = bu.sample_records_native(robot, task, nn, seed=777)
  arr = o.make_batch(rf, ri)
  rng, mt = np.random.RandomState(1), np.random.RandomState(2)
  cands = {f'{N:3d} cold sweeps': cfg(sweeps=N) for N in (4, 16, 64, 256)}
  if robot == 'doggo':
    cands['round-2 spec (4 cold)'] = cfg(dg=(4, 4, 0))
    for a in (8, 16, 32):
      cands[f'warm {a} / cold {2 * a}'] = cfg(dg=(a, 2 * a, 1))
  stat = {k: dict(dq=[], dv=[], cost=0, met=0) for k in cands}
  total = 0
  E = 144
  qcols = [0, 1, 2] + ([E] + list(range(E + 1, E + 5)) + list(range(E + 9, E + 22)) if robot == 'doggo' else [])
  vcols = [3, 4, 5] + ([E + 5, E + 6, E + 7, E + 8] + list(range(E + 22, E + 35)) if robot == 'doggo' else ([E, E + 1] if robot == 'car' else []))
  for t in range(T):
    rf, ri = o.batch_records(arr)
    act = mt.uniform(-1, 1, size=(nn, nu)).astype(np.float32) if robot == 'doggo' else bu.pursuit_actions(rf, ri, rng, robot=robot)
    noise = mt.normal(size=(nn, nu)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(nn, 64), dtype=np.uint32)
    outs = {}
    for k, f in cands.items():
      f()
      a2 = o.make_batch(rf, ri)
      r = o.step_batch_full(a2, rid, act, noise, tape, obs_dim=od)
      outs[k] = (o.batch_records(a2)[0], r[2].copy(), r[4].copy())
    SPEC()
    r1 = o.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    s1 = o.batch_records(arr)[0]
    total += nn
    ref = outs['256 cold sweeps']
    for k in cands:
      # the sweep rows are compared with the SPECIFICATION's step, the other rows (variants) with the converged step
      sN, cN, mN = outs[k]
      base_s, base_c, base_m = (s1, r1[2], r1[4]) if k.endswith('cold sweeps') else ref
      stat[k]['dq'].append(np.abs(sN[:, qcols] - base_s[:, qcols]).max(1))
      stat[k]['dv'].append(np.abs(sN[:, vcols] - base_s[:, vcols]).max(1))
      stat[k]['cost'] += int((cN != base_c).sum())
      stat[k]['met'] += int((mN != base_m).sum())
  say(f'## {robot}/{task}: {nn} envs x {T} steps = {total} env-steps; specification cost rate {float(np.mean(r1[2])):.3f} (last step)')
  for k in cands:
    dq, dv = np.concatenate(stat[k]['dq']), np.concatenate(stat[k]['dv'])
    what = 'specification vs' if k.endswith('cold sweeps') else 'vs converged (256):'
    say(f'  {what} {k:22s}: qpos diff median/p90/p99/max {np.median(dq):.2e}/{np.quantile(dq, .9):.2e}/{np.quantile(dq, .99):.2e}/{dq.max():.2e} | '
        f'qvel median/p99 {np.median(dv):.2e}/{np.quantile(dv, .99):.2e} | cost-flag disagreement {stat[k]["cost"] / total:.2e} | goal-met {stat[k]["met"] / total:.2e}')
# Car: straight-line heading drift under equal full throttle (<= 0.01 rad / 250 steps)
import test_oracle_physics as tp
SPEC()
rf1, ri1 = tp.empty_world()
rf1[tp.F_ROBOT + 2] = 0.3
e, _ = tp.car_run(o, rf1, ri1, [1.0, 1.0], 250)
f = o.record(e)[0]
yaw = f[tp.F_ROBOT + 2]
say(f'## car straight line, 250 steps at full throttle: heading drift {yaw - 0.3:+.4f} rad, speed {float(np.dot(f[tp.F_ROBOT + 3:tp.F_ROBOT + 5], [np.sin(yaw), -np.cos(yaw)])):.4f} m/s '
    '(round 2, one sweep of six elements: 0.06 rad, 0.77 m/s; 64 sweeps of them: 0.009 rad, 0.82 m/s)')
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'r04_sweep_convergence.txt')
open(out, 'w').write('\n'.join(lines) + '\n')
