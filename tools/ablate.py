"""GPU: time ONE step kernel from a state snapshot (250 burn-in steps) for several library variants
(timing-only ablation builds, see SAG_ABLATE in sag_device.hpp).
  python tools/ablate.py snapshot          -> gpurun_out/snap.npz (burn-in with the default lib)
  SAG_LIB=... python tools/ablate.py time  -> mean kernel ms over repeats from the snapshot"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
from safe_adaptation_gym_amd import _native as nat
N = 1 << 20
snap = '/tmp/sag_snap.npz'  # large: keep it out of gpurun_out (64 MiB merge limit)
if sys.argv[1] == 'snapshot':
  run = bench.DeviceRun('go_to_goal', N, 0, 0)
  run.burn_in(250)
  rf, ri = run.ctx.get_state()
  np.savez(snap, rf=rf, ri=ri)
  print('snapshot saved', (np.abs(rf[:, 81:141].reshape(-1, 10, 6)[:, :, 3:]).max(-1) > 0).mean() * 10, 'moving vases/env')
else:
  z = np.load(snap)
  ctx = nat.Context('point', N, device=0, seed=666, max_buttons=0, has_box=False)
  ctx.set_layout(z['rf'], z['ri'])
  d_act = ctx.dev_alloc(N * 8); ctx.dev_fill_actions(d_act, 3)
  d_obs = ctx.dev_alloc(N * 240); d_rew = ctx.dev_alloc(N * 8)
  d_c, d_d, d_m = ctx.dev_alloc(N), ctx.dev_alloc(N), ctx.dev_alloc(N)
  for r in range(12):
    ctx.reset()
    # install flags every env busy: one untimed step classifies, the second one is timed
    ctx.enable_timing(False)
    ctx.step_device(d_act, None, -1, d_obs, d_rew, d_c, d_d, d_m)
    ctx.enable_timing(True)
    if r == 2: ctx.kernel_time_ms(reset=True)
    ctx.step_device(d_act, None, -1, d_obs, d_rew, d_c, d_d, d_m)
  ms, k = ctx.kernel_time_ms()
  print(f'{os.path.basename(os.environ.get("SAG_LIB", "default")):28s} kernel {ms:.4f} ms  ({k} launches, {N} envs)')
