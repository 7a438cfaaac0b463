import os, sys, subprocess, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  import bench
  out = []
  for robot, task in (('doggo', 'haul_box'), ('point', 'push_box'), ('car', 'press_buttons')):
    r = bench.DeviceRun(task, 512, 0, 0, robot=robot)
    r.burn_in(30)
    out.append(r.ctx.render_rgb())
    for cam, (w, h) in ((1, (96, 72)), (2, (130, 50))):
      try:
        out.append(r.ctx.render(camera=cam, width=w, height=h, overlays=True)[:64])
      except Exception as e:
        print('render() variant skipped:', e)
  np.savez(sys.argv[2], *out)
else:
  for name, lib in (('new', None), ('old', os.path.join(ROOT, 'safe_adaptation_gym_amd', 'libsag_rold.so'))):
    env = dict(os.environ)
    if lib: env['SAG_LIB'] = lib
    subprocess.check_call([sys.executable, __file__, 'one', f'/tmp/img_{name}.npz'], env=env)
  a, b = np.load('/tmp/img_new.npz'), np.load('/tmp/img_old.npz')
  for k in a.files:
    print(k, a[k].shape, 'identical' if np.array_equal(a[k], b[k]) else f'DIFFER in {(a[k] != b[k]).sum()} bytes')
