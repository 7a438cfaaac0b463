"""GPU: the two regimes of the headline (0.83 / 0.95 ms per step) under hardware counters.  ONE process
creates R contexts one after the other on the same layouts (as tools/alloc_spread.py: the regime comes with the
allocation), burn-in 200 steps + K steps each, and prints each context's ms per step; run it directly after `rocprofv3
--pmc ... --` and tools/regime_diff.py lines the slowest context's counters up against the fastest one's.
  python tools/regime_probe.py [R=6] [envs=4194304] [K=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
K = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rf, ri = bench.build_records('go_to_goal', envs, 0)
bench.build_records = lambda *a, **k: (rf, ri)   # the same layouts for every context
for r in range(R):
  run = bench.DeviceRun('go_to_goal', envs, 0, 0)
  run.burn_in(200)
  run.wait(); t0 = time.perf_counter(); run.run(K); run.wait()
  print(f'context {r}: {(time.perf_counter() - t0) / K * 1e3:.4f} ms/step over {K} steps after 200', flush=True)
  run.close()
