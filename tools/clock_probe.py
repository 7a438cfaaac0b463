"""GPU: is the headline's step time tied to the clocks?  Steps the headline config continuously for ~S seconds in blocks of 250 steps
(population frozen by restoring the age-200 state before every block, so the load is the same), printing ms/step per block beside
rocm-smi's sclk / mclk / fclk / power / temperature sampled in a side thread.   python tools/clock_probe.py [seconds=20]"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

S = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
run = bench.DeviceRun('go_to_goal', 1 << 22, 0, 0)
run.burn_in(200)
rf, ri = run.ctx.get_state()
samples, stop = [], False


def smi():
  while not stop:
    try:
      out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=10).stdout
      g = lambda pat: (re.search(pat, out) or [None, '?'])[1]
      samples.append((time.perf_counter(), g(r'sclk clock level: \S+ \((\d+)Mhz\)'), g(r'mclk clock level: \S+ \((\d+)Mhz\)'), g(r'fclk clock level: \S+ \((\d+)Mhz\)'),
                      g(r'Graphics Package Power \(W\): ([\d.]+)'), g(r'Sensor junction\) \(C\): ([\d.]+)'), g(r'Sensor memory\) \(C\): ([\d.]+)')))
    except Exception as e:   # noqa: BLE001
      samples.append((time.perf_counter(), 'err', str(e)[:40]))
    time.sleep(0.5)


th = threading.Thread(target=smi, daemon=True)
th.start()
t_end = time.perf_counter() + S
blk = 0
while time.perf_counter() < t_end:
  run.ctx.set_state(rf, ri)      # same population every block
  run.run(5); run.wait()
  t0 = time.perf_counter(); run.run(250); run.wait(); t1 = time.perf_counter()
  near = [s for s in samples if t0 <= s[0] <= t1] or samples[-1:]
  print(f'block {blk}: {(t1 - t0) / 250 * 1e3:.4f} ms/step | sclk/mclk/fclk MHz, W, Tj, Tmem: {[s[1:] for s in near]}', flush=True)
  blk += 1
stop = True
