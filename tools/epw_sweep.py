"""GPU: step time of the single-launch form against envs per wavefront (small batches)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import bench
  task, robot, n = sys.argv[2], sys.argv[3], int(sys.argv[4])
  run = bench.DeviceRun(task, n, 0, 0, robot=robot)
  run.burn_in(150); run.timing(True); run.run(100); run.wait()
  print(json.dumps({'ms': run.kernel_time_ms()[0]}))
else:
  for task, robot in (('go_to_goal', 'point'), ('push_box', 'car')):
    for n in (4096, 16384):
      row = []
      for epw in ('64', '32', '16', '8', '4'):
        env = dict(os.environ, SAG_EPW=epw, SAG_SPLIT='0')
        out = subprocess.run([sys.executable, __file__, 'one', task, robot, str(n)], env=env, capture_output=True, text=True)
        row.append(json.loads(out.stdout.strip().splitlines()[-1])['ms'] if out.returncode == 0 else float('nan'))
      print(f'{robot}/{task} N={n:6d}  epw 64/32/16/8/4: ' + '  '.join(f'{v:.4f}' for v in row) + ' ms', flush=True)
