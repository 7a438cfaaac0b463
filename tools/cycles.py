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
run.run(K); run.wait()
c = run.ctx.debug_cycles().astype(np.float64)
print(f'{task} {robot} {envs} envs, {K} steps, busy fraction {run.ctx.busy_count() / envs:.4f}')
for mode, name in enumerate(['single', 'quiet', 'busy']):
  waves = c[mode, 15]
  if not waves: continue
  tot = c[mode, :15].sum()
  print(f'== {name}: {waves / K:.0f} wavefronts/step, {tot / waves:.0f} ticks/wavefront')
  for k, n in enumerate(NAMES):
    print(f'   {n:22s} {c[mode, k] / waves:10.0f}  {100 * c[mode, k] / tot:5.1f} %')
