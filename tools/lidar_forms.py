"""GPU: the lidar + cost entry at n poses x 21 points under each kernel form (SAG_LIDAR_TEAM = 0 / 4 / 16, SAG_LIDAR_REG = 0), kernel-only time.
  python tools/lidar_forms.py [n]"""
import os, subprocess, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import numpy as np
  from safe_adaptation_gym_amd import _native as nat
  n, nK = int(sys.argv[2]), 21
  rs = np.random.RandomState(0)
  rob = np.concatenate([rs.uniform(-2, 2, (n, 2)), rs.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
  pts = rs.uniform(-2.5, 2.5, (n, nK, 2)).astype(np.float32)
  grp = np.tile(np.array([1 + 128] * 8 + [1] * 12 + [2], np.uint8), (n, 1))
  cx = nat.Context('point', 64, device=0)
  d = [cx.dev_alloc(x.nbytes) for x in (rob, pts, grp)]
  for dp, x in zip(d, (rob, pts, grp)):
    cx.dev_upload(dp, x)
  d_lid, d_cost = cx.dev_alloc(n * 48 * 4), cx.dev_alloc(n)
  for _ in range(5):
    cx.lidar_cost_device(n, nK, d[0], d[1], d[2], d_lid, None, d_cost)
  cx.wait(); cx.enable_timing(True); cx.kernel_time_ms(reset=True)
  for _ in range(30):
    cx.lidar_cost_device(n, nK, d[0], d[1], d[2], d_lid, None, d_cost)
  cx.wait()
  ms, cnt = cx.kernel_time_ms(reset=True)
  print(json.dumps({'ms': ms, 'frac': 376 * n / (ms * 1e-3) / 8e12}))
else:
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
  for name, extra in (('default', {}), ('team 2', {'SAG_LIDAR_TEAM': '2'}), ('team 4', {'SAG_LIDAR_TEAM': '4'}), ('team 8', {'SAG_LIDAR_TEAM': '8'}), ('team 16', {'SAG_LIDAR_TEAM': '16'}), ('lane per pose, LDS-staged', {'SAG_LIDAR_TEAM': '0', 'SAG_LIDAR_REG': '0'})):
    out = subprocess.run([sys.executable, __file__, 'one', str(n)], env=dict(os.environ, **extra), capture_output=True, text=True)
    print(f'{n} poses, {name:28s} ' + (out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:]), flush=True)
