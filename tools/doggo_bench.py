"""GPU: Doggo step time per task and per step index (kernel time by HIP events).
  python tools/doggo_bench.py [envs=4096] [steps=40]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for task in ('go_to_goal', 'haul_box', 'multitask'):
  r = bench.DeviceRun(task, n, 0, 0, robot='doggo')
  r.timing(True)
  ms = []
  for t in range(T):
    r.step(); r.wait()
    ms.append(r.kernel_time_ms()[0])
  rf, ri = r.ctx.get_state()
  print(f'doggo {task:12s} N={n}: kernel ms per step, steps 0.. : ' + ' '.join(f'{m:.2f}' for m in ms[:12]) + f' ... mean of last 10: {np.mean(ms[-10:]):.3f} | row-overflow flags: {int((ri[:, 13] & 4 != 0).sum())}', flush=True)
  r.close()
