"""GPU: steady-state step time and throughput of Point/GoToGoal against the batch size (split form).
  python tools/batch_sweep.py 524288 1048576 ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import bench
  n = int(sys.argv[2])
  run = bench.DeviceRun('go_to_goal', n, 0, 0, robot='point')
  run.burn_in(200); run.timing(True); run.run(100); run.wait()
  print(json.dumps({'ms': run.kernel_time_ms()[0], 'busy': run.ctx.busy_count() / n}))
else:
  for n in [int(a) for a in sys.argv[1:]]:
    out = subprocess.run([sys.executable, __file__, 'one', str(n)], capture_output=True, text=True)
    if out.returncode:
      print(n, 'failed', out.stderr[-300:], flush=True)
      continue
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(f'N={n:8d}  {r["ms"]:.4f} ms/step  {n / r["ms"] / 1e6:.3f}e9 env-steps/s  frac {892 * n / r["ms"] / 1e6 / 8000:.4f}  busy {r["busy"]:.4f}  busy wavefronts/SIMD {r["busy"] * n / 64 / 1024:.2f}', flush=True)
