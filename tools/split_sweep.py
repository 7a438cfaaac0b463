"""GPU: step time of the single-kernel and the split (quiet + busy) launch forms vs batch size."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import bench
  task, robot, n = sys.argv[2], sys.argv[3], int(sys.argv[4])
  run = bench.DeviceRun(task, n, 0, 0, robot=robot)
  run.burn_in(200); run.timing(True); run.run(100); run.wait()
  print(json.dumps({'ms': run.kernel_time_ms()[0]}))
else:
  for task, robot in (('go_to_goal', 'point'), ('push_box', 'car')):
    for n in (4096, 32768, 131072, 262144, 524288):
      row = []
      for split in ('0', '1'):
        env = dict(os.environ, SAG_SPLIT=split)
        out = subprocess.run([sys.executable, __file__, 'one', task, robot, str(n)], env=env, capture_output=True, text=True)
        row.append(json.loads(out.stdout.strip().splitlines()[-1])['ms'] if out.returncode == 0 else float('nan'))
      print(f'{robot}/{task} N={n:7d}  single {row[0]:.4f} ms   split {row[1]:.4f} ms', flush=True)
