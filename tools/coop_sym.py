"""GPU: identical envs in both halves of the cooperative Doggo kernel must give identical results."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import batch_util as bu
from safe_adaptation_gym_amd import _native as nat
n = 4
rf, ri = bu.sample_records_native('doggo', 'go_to_goal', 1, seed=666)
rf = np.repeat(rf, n, 0); ri = np.repeat(ri, n, 0)
os.environ['SAG_DOGGO_COOP'] = '1'
c = nat.Context('doggo', n, seed=1)
c.set_layout(rf, ri)
x0 = c.get_state()[0][:, :6].copy()
print('initial robot6', x0[0], flush=True)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
  out = c.step(np.zeros((n, 12), np.float32), np.zeros((n, 12), np.float32))
  a, _ = c.get_state()
  d = np.abs(a[0] - a[1])
  print('   robot6 even', a[0, :6], 'odd', a[1, :6], flush=True)
  print(f'step {t}: even-odd max diff {d.max():.3e} at fields {np.argwhere(d > 1e-9).ravel()[:12]}; even0-even2 {np.abs(a[0]-a[2]).max():.2e}; z {a[0,144]:.4f} {a[1,144]:.4f}', flush=True)
