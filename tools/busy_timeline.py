"""GPU: start / end of every wavefront of the BUSY kernel in one step (library built with -DSAG_WAVE_TIMES: two clock reads and one
store per wavefront, nothing else changes), and what the launch would take if its wavefronts were handed out longest first.
  python -m safe_adaptation_gym_amd.build --out libsag_wt.so -DSAG_WAVE_TIMES
  SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_wt.so python tools/busy_timeline.py [task] [robot] [envs]"""
import ctypes as C
import heapq
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


def times(reset=False, full=False):
  n = (2 + 32 + 64 + 2) * WT if full else 2 * WT
  out = np.zeros(n, np.uint64)
  ctx._check(ctx.lib.sag_debug_cycles(ctx.h, int(reset), out.ctypes.data_as(C.POINTER(C.c_uint64)), n), 'sag_debug_cycles')
  t = out[:2 * WT].reshape(WT, 2).astype(np.int64)
  if not full:
    return t
  env = out[2 * WT:34 * WT].view(np.int32).reshape(WT, 64)
  work = out[34 * WT:98 * WT].view(np.uint8).reshape(WT, 64, 8)
  trips = out[98 * WT:].view(np.uint16).reshape(WT, 8)
  return t, env, work, trips


def packed(dur, slots, order):
  """greedy list schedule: the span when `slots` servers take the wavefronts in `order`."""
  free = [0.0] * slots
  heapq.heapify(free)
  end = 0.0
  for k in order:
    t = heapq.heappop(free) + dur[k]
    end = max(end, t)
    heapq.heappush(free, t)
  return end


for rep in range(5):
  times(reset=True)
  run.run(1); run.wait()
  t, env, work, trips = times(full=True)
  keep = t[:, 1] > 0
  if rep == 4:
    out = os.path.join(ROOT, 'gpurun_out', f'busy_timeline_{robot}_{task}_{envs}.npz')
    np.savez_compressed(out, t=t[keep], env=env[keep], work=work[keep], trips=trips[keep])
  t = t[keep]
  n = len(t)
  t0 = t[:, 0].min()
  s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # us (100-MHz clock)
  dur = e - s
  span = e.max()
  # resident wavefronts: how many are in flight at the median start
  slots = int(((s <= np.median(s)) & (e > np.median(s))).sum())
  peak = max(int(((s <= x) & (e > x)).sum()) for x in np.quantile(s, [0.1, 0.3, 0.5, 0.7]))
  last = np.argsort(-e)[:5]
  print(f'step {rep}: {n} busy wavefronts, span {span:.0f} us; duration mean {dur.mean():.0f} / p50 {np.median(dur):.0f} / p99 {np.quantile(dur, .99):.0f} / max {dur.max():.0f} us; '
        f'resident at once ~{peak}; sum / resident = {dur.sum() / peak:.0f} us')
  print('   the five that end last: ' + '  '.join(f'[start {s[k]:.0f} dur {dur[k]:.0f}]' for k in last))
  print(f'   list schedule on {peak} slots: launch order {packed(dur, peak, range(n)):.0f} us, longest first {packed(dur, peak, np.argsort(-dur)):.0f} us')
