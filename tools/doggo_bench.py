"""GPU: Doggo multitask step time, lane-per-env vs wave-cooperative physics."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
for n in [int(a) for a in sys.argv[1:]] or (4096, 32768):
  for flag in ('0', '1'):
    os.environ['SAG_DOGGO_COOP'] = flag
    r = bench.DeviceRun('multitask', n, 0, 0, robot='doggo')
    r.burn_in(15)
    r.timing(True)
    t = bench.timed(r, 20, 3, lambda: None)
    ms, _ = r.kernel_time_ms()
    print(f'doggo multitask N={n} coop={flag}: {t / 20 * 1e3:.3f} ms/step (kernels {ms:.3f} ms) = {n * 20 / t:.3e} env-steps/s', flush=True)
    r.close()
