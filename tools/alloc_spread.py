"""GPU: does the headline's run-to-run spread come with the ALLOCATION?  One process, the same layouts, the context (all device
buffers) created afresh R times: burn-in 200, then 100 timed steps each.  Also times a second pass on the SAME context.
  python tools/alloc_spread.py [R=5] [envs=4194304]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_adaptation_gym_amd import _native as nat

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
rf, ri = bench.build_records('go_to_goal', envs, 0)
orig = bench.build_records
bench.build_records = lambda *a, **k: (rf, ri)   # the same layouts for every context
for r in range(R):
  run = bench.DeviceRun('go_to_goal', envs, 0, 0)
  run.burn_in(200)
  out = []
  for rep in range(2):
    run.wait(); t0 = time.perf_counter(); run.run(100); run.wait()
    out.append((time.perf_counter() - t0) * 10)
  print(f'context {r}: ms/step {out[0]:.4f} then {out[1]:.4f} (same context, steps 200-300 / 300-400)', flush=True)
  run.close()
