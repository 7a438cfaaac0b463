"""GPU experiment (library built with -DSAG_WAVE_TIMES): what would busy wavefronts of ONE kind of contact each take?  The envs of a
running batch are re-installed sorted by what their robot touched in the last step (task object / static circle / vases only / nothing),
so that the busy list - which keeps env order - comes out sorted by kind; busy wavefront durations and the step time before and after.
  SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_wt.so python tools/busy_sort_probe.py [task] [robot] [envs]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
task = sys.argv[1] if len(sys.argv) > 1 else 'push_box'
robot = sys.argv[2] if len(sys.argv) > 2 else 'car'
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
run = bench.DeviceRun(task, envs, 0, 0, robot=robot)
run.burn_in(200)
ctx = run.ctx
WT = 16384


def times(reset=False):
  n = (2 + 32 + 64 + 2) * WT
  out = np.zeros(n, np.uint64)
  ctx._check(ctx.lib.sag_debug_cycles(ctx.h, int(reset), out.ctypes.data_as(C.POINTER(C.c_uint64)), n), 'sag_debug_cycles')
  t = out[:2 * WT].reshape(WT, 2).astype(np.int64)
  env = out[2 * WT:34 * WT].view(np.int32).reshape(WT, 64)
  work = out[34 * WT:98 * WT].view(np.uint8).reshape(WT, 64, 8)
  trips = out[98 * WT:].view(np.uint16).reshape(WT, 8)
  keep = t[:, 1] > 0
  return t[keep], env[keep], work[keep], trips[keep]


def report(tag, steps=5):
  for rep in range(steps):
    times(reset=True)
    run.timing(True); run.run(1); run.wait()
    ms = run.kernel_time_ms()[0]; run.timing(False)
    t, env, work, trips = times()
    dur = (t[:, 1] - t[:, 0]) / 100.0
    span = (t[:, 1].max() - t[:, 0].min()) / 100.0
    print(f'{tag} step {rep}: {ms:.3f} ms; {len(dur)} busy wavefronts, span {span:.0f} us, duration mean {dur.mean():.0f} p10 {np.quantile(dur, .1):.0f} p50 {np.median(dur):.0f} '
          f'p90 {np.quantile(dur, .9):.0f} max {dur.max():.0f}; passes (static / vase / object): {trips[:, 0].mean():.1f} / {trips[:, 1].mean():.1f} / {trips[:, 2].mean():.1f}', flush=True)
  return t, env, work, trips


t, env, work, trips = report('as sampled')
kind = np.full(envs, -1, np.int8)
live = env >= 0
k = (work[:, :, 2] > 0).astype(np.int8) + 2 * (work[:, :, 0] > 0).astype(np.int8)
if os.environ.get('PROBE_FINE'):   # a third bit: the robot touched a vase or a free body moved
  k = k + 4 * ((work[:, :, 1] > 0) | (work[:, :, 5] > 0)).astype(np.int8)
kind[env[live]] = k[live]
print('busy envs by kind (bit 0 object, bit 1 static, bit 2 vases / moving bodies):', np.bincount(kind[kind >= 0], minlength=4), flush=True)
order = np.argsort(-kind.astype(np.int32), kind='stable').astype(np.int32)
CH = 1 << 18
for a in range(0, envs, CH):          # re-install in sorted order, a chunk at a time (host memory)
  src = order[a:a + CH]
  rf, ri = ctx.get_state(src)
  if a == 0:
    recs = []
  recs.append((rf, ri))
pos = 0
for rf, ri in recs:
  ids = np.arange(pos, pos + len(rf), dtype=np.int32)
  ctx.set_state(rf, ri, ids)
  pos += len(rf)
run.run(2); run.wait()      # the step after an install takes every env as busy
report('sorted by kind', steps=6)
