"""GPU: census of the busy envs after burn-in (which condition makes them busy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_adaptation_gym_amd import _native as nat
N = 1 << 18
run = bench.DeviceRun('go_to_goal', N, 0, 0)
run.burn_in(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
rf, ri = run.ctx.get_state()
print('busy fraction', run.ctx.busy_count() / N)
rob = rf[:, nat.F_ROBOT:nat.F_ROBOT + 6]
V = rf[:, nat.F_VASES:nat.F_VASES + 60].reshape(N, 10, 6)
nV = ri[:, nat.I_NV]
valid = np.arange(10)[None, :] < nV[:, None]
moving = (np.abs(V[:, :, 3:6]).max(-1) > 0) & valid
nawake = moving.sum(1)
T = 0.01
speed = np.hypot(rob[:, 3], rob[:, 4])
reach = speed * T + 1.05 * 0.3 * 0.05 / 0.0046 * T * T + 0.005
d = np.hypot(V[:, :, 0] - rob[:, None, 0], V[:, :, 1] - rob[:, None, 1])
vr = 0.1 * 2 ** 0.5
near_v = ((d <= (0.15 + vr + reach[:, None])) & valid).any(1)
P = rf[:, nat.F_PILLARS:nat.F_PILLARS + 4].reshape(N, 2, 2)
nP = ri[:, nat.I_NP]
dp = np.hypot(P[:, :, 0] - rob[:, None, 0], P[:, :, 1] - rob[:, None, 1])
near_p = ((dp <= 0.15 + 0.2 + reach[:, None]) & (np.arange(2)[None, :] < nP[:, None])).any(1)
near = near_v | near_p
print('awake>0: %.4f  robot near: %.4f  both: %.4f  either: %.4f' % ((nawake > 0).mean(), near.mean(), ((nawake > 0) & near).mean(), ((nawake > 0) | near).mean()))
for k in range(5):
  print('  awake == %d: %.4f   and robot not near: %.4f' % (k, (nawake == k).mean(), ((nawake == k) & ~near).mean()))
# isolated awake vase: no other vase / pillar / robot within bounds + travel
vs = np.hypot(V[:, :, 3], V[:, :, 4])
trav = vs * T + 0.002
dvv = np.hypot(V[:, :, None, 0] - V[:, None, :, 0], V[:, :, None, 1] - V[:, None, :, 1])
pairvalid = valid[:, :, None] & valid[:, None, :] & ~np.eye(10, dtype=bool)[None]
close_vv = ((dvv <= 2 * vr + trav[:, :, None] + trav[:, None, :]) & pairvalid).any(2)
dvp = np.hypot(V[:, :, None, 0] - P[:, None, :, 0], V[:, :, None, 1] - P[:, None, :, 1])
close_vp = ((dvp <= vr + 0.2 + trav[:, :, None]) & (np.arange(2)[None, None, :] < nP[:, None, None])).any(2)
close_vr = d <= 0.15 + vr + reach[:, None] + trav
iso = moving & ~close_vv & ~close_vp & ~close_vr
all_iso = (moving == iso).all(1) & (nawake > 0)
print('envs whose awake vases are all isolated: %.4f; with awake==1: %.4f; and robot not near: %.4f' % (all_iso.mean(), (all_iso & (nawake == 1)).mean(), (all_iso & (nawake == 1) & ~near).mean()))
print('vase speed quantiles (moving):', np.quantile(vs[moving], [.1, .5, .9, .99]))
w = np.abs(V[:, :, 5])[moving]
print('vase |w| quantiles:', np.quantile(w, [.1, .5, .9, .99]))
