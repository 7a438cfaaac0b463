"""GPU: what does a wavefront with a HEAVY Doggo env cost?  Fills a context with copies of one recorded heavy state
(tools/doggo_overflow_probe.py -> gpurun_out/r04_heavy_envs.npz, copied to tools/data/; zero actions and noise, so the copies stay identical),
and prints the kernel time per step and - with a -DSAG_CYCLES library - the section profile.
  [SAG_LIB=...libsag_cyc.so] python tools/doggo_heavy_profile.py [envs=4096] [steps=5] [which=0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from safe_adaptation_gym_amd import _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = np.load(os.path.join(ROOT, 'tools', 'data', 'r04_heavy_envs.npz'))
rf = np.repeat(d['rf'][which:which + 1], n, 0).astype(np.float32); ri = np.repeat(d['ri'][which:which + 1], n, 0)
ctx = nat.Context('doggo', n, seed=1)
ctx.set_layout(rf, ri)
ctx.set_state(rf, ri)
ctx.enable_timing(True)
zero = np.zeros((n, 12), np.float32)
tape = np.zeros((n, 64), np.uint32)
try:
  ctx.debug_cycles(reset=True)
  cyc = True
except Exception:
  cyc = False
for t in range(T):
  ctx.step(zero, zero, tape)
  ms = ctx.kernel_time_ms(reset=True)[0]
  print(f'step {t}: kernels {ms:.3f} ms', flush=True)
NAMES = ['load', 'kinematics', 'composite + RNEA bodies', 'tau', 'cholesky', 'M^-1 + qacc0', 'rows: limits + floor', 'rows: world objects',
         'rows finish (W, A)', 'PGS', 'after PGS (qacc, touch)', 'planar world + integrate', 'store', 'CRBA forces + RNEA bias', 'CRBA rows']
if cyc:
  c = ctx.debug_cycles().astype(np.float64)
  if c[1, 15] > 0:
    tot = c[1, :len(NAMES)].sum()
    print(f'{tot / c[1, 15]:.0f} ticks per wavefront-step')
    for k, nm in enumerate(NAMES):
      print(f'   {nm:28s} {c[1, k] / c[1, 15]:12.0f}  {100 * c[1, k] / tot:5.1f} %')
    print('rows per evaluation, buckets of 4: ' + ' '.join(str(int(v)) for v in c[0]))
ctx.close()
