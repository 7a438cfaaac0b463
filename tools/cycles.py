"""GPU: per-section wavefront clock profile of the step kernels (library built with -DSAG_CYCLES).
  SAG_LIB=$PWD/safe_adaptation_gym_amd/libsag_cyc.so python tools/cycles.py [task] [robot] [envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
NAMES = ['load', 'robot', 'robot-static', 'robot-free', 'free-static', 'free-free broad', 'free-free narrow',
         'integrate', 'writeback+classify', 'reward', 'resample', 'cost', 'lidar+sensors', 'obs store', 'tail']
task = sys.argv[1] if len(sys.argv) > 1 else 'go_to_goal'
robot = sys.argv[2] if len(sys.argv) > 2 else 'point'
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
run = bench.DeviceRun(task, envs, 0, 0, robot=robot)
run.burn_in(200)
run.ctx.debug_cycles(reset=True)
K = 20
if os.environ.get('SAG_CYC_ONE_STEP'):   # the slowest wavefront of ONE step, and the state its envs had BEFORE that step
  K = 1
  pre = run.ctx.get_state()
run.run(K); run.wait()
c = run.ctx.debug_cycles().astype(np.float64)
w, wblock = run.ctx.debug_cycles(worst=True)
w = w.astype(np.float64)
print(f'{task} {robot} {envs} envs, {K} steps, busy fraction {run.ctx.busy_count() / envs:.4f}')
for mode, name in enumerate(['single', 'quiet', 'busy']):
  waves = c[mode, 15]
  if not waves: continue
  tot = c[mode, :15].sum()
  print(f'== {name}: {waves / K:.0f} wavefronts/step, {tot / waves:.0f} ticks/wavefront')
  print(f'   {"":22s} {"mean":>10s}  {"":7s} {"slowest wavefront":>18s} (total {w[mode, 15]:.0f} ticks)')
  for k, n in enumerate(NAMES):
    print(f'   {n:22s} {c[mode, k] / waves:10.0f}  {100 * c[mode, k] / tot:5.1f} % {w[mode, k]:18.0f}')
  hist = run.ctx.cycles_hist[mode].astype(np.float64)
  cum = np.cumsum(hist[::-1])[::-1]
  print('   wavefronts by total ticks (quarter octaves): ' + '  '.join(f'{2 ** ((b + 40) / 4) / 1e3:.0f}k:{int(hist[b])}' for b in range(64) if hist[b]))
  mid = np.array([2 ** ((b + 40.5) / 4) for b in range(64)])
  share = np.cumsum((hist * mid)[::-1])[::-1] / max((hist * mid).sum(), 1)
  for frac in (0.01, 0.001):
    b = int(np.argmax(cum <= frac * cum[0])) if (cum <= frac * cum[0]).any() else 63
    print(f'   the slowest {100 * frac:.1f} % of the wavefronts (>= {2 ** ((b + 40) / 4) / 1e3:.0f}k ticks) hold {100 * share[b]:.1f} % of the ticks')

if os.environ.get('SAG_CYC_ONE_STEP'):
  mode = 0 if c[0, 15] else 2
  n_waves = int(c[mode, 15])
  epw = -(-envs // n_waves) if mode == 0 else 64
  b = int(wblock[mode])
  ids = np.arange(b * epw, min(envs, (b + 1) * epw))
  out = os.path.join(ROOT, 'gpurun_out', f'cyc_worst_{robot}_{task}_{envs}.npz')
  np.savez_compressed(out, rf=pre[0], ri=pre[1], ids=ids, block=b, epw=epw, sections=w[mode])
  print(f'slowest wavefront: block {b} (envs {ids[0]}..{ids[-1]}), records before the step -> {out}')
