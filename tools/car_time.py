"""GPU: Car / push_box step time (BASELINE config 3) at a few batch sizes.  SAG_LIB selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
for n in [int(a) for a in sys.argv[1:]] or [4096, 262144]:
  r = bench.DeviceRun('push_box', n, 0, 0, robot='car')
  r.burn_in(60); r.timing(True); r.run(100); r.wait()
  ms = r.kernel_time_ms()[0]
  print(f'{os.path.basename(os.environ.get("SAG_LIB", "libsag.so")):16s} car/push_box N={n:7d}  {ms:.4f} ms/step  {n / ms / 1e3:.3f}e6 env-steps/s', flush=True)
  r.close()
