"""GPU: kernel time and number of moving vases as the simulation ages."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
run = bench.DeviceRun('go_to_goal', n, 0, 0)
run.ctx.enable_timing(True)
for blk in range(12):
  run.ctx.kernel_time_ms(reset=True)
  t0 = time.perf_counter(); run.run(25); run.ctx.wait(); dt = time.perf_counter() - t0
  ms, k = run.ctx.kernel_time_ms(reset=True)
  rf, ri = run.ctx.get_state(np.arange(4096, dtype=np.int32))
  v = rf[:, 81:141].reshape(-1, 10, 6)[:, :, 3:]
  moving = (np.abs(v).max(-1) > 0).mean() * 10
  tiny = ((np.abs(v).max(-1) > 0) & (np.abs(v).max(-1) < 1e-6)).mean() * 10
  print(f'steps {blk*25:4d}-{blk*25+24:4d}: event kernel {ms:.3f} ms, wall/step {dt/25*1e3:.3f} ms, moving vases/env {moving:.2f} (|v|<1e-6: {tiny:.2f})', flush=True)
