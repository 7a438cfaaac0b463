"""GPU: step time of the split form against the envs a busy wavefront takes (SAG_BUSY_E fixed, or balanced over SAG_BUSY_SLOTS).
  python tools/busy_e_sweep.py [envs]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import bench
  task, robot, n = sys.argv[2], sys.argv[3], int(sys.argv[4])
  run = bench.DeviceRun(task, n, 0, 0, robot=robot)
  run.burn_in(200); run.timing(True); run.run(40); run.wait()
  print(json.dumps({'ms': run.kernel_time_ms()[0], 'busy': run.ctx.busy_count()}))
else:
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
  cases = [('E=64', {'SAG_BUSY_E': '64'}), ('E=56', {'SAG_BUSY_E': '56'}), ('E=48', {'SAG_BUSY_E': '48'}), ('E=40', {'SAG_BUSY_E': '40'}),
           ('slots=2048', {'SAG_BUSY_SLOTS': '2048'}), ('slots=1920', {'SAG_BUSY_SLOTS': '1920'}), ('slots=1792', {'SAG_BUSY_SLOTS': '1792'}),
           ('slots=1536', {'SAG_BUSY_SLOTS': '1536'}), ('slots=1100', {'SAG_BUSY_SLOTS': '1100'}), ('slots=1024', {'SAG_BUSY_SLOTS': '1024'})]
  for task, robot in (('push_box', 'car'), ('go_to_goal', 'point')):
    for name, extra in cases:
      env = dict(os.environ, **extra)
      out = subprocess.run([sys.executable, __file__, 'one', task, robot, str(n)], env=env, capture_output=True, text=True)
      r = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else {'ms': float('nan'), 'busy': -1}
      print(f'{robot}/{task} N={n} {name:12s} {r["ms"]:.4f} ms  (busy envs {r["busy"]})', flush=True)
