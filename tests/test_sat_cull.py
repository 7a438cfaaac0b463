"""The separating-axis cull in front of the box-box vertex tests (csrc/sag_device.hpp: boxes_separated, and the pair
mask of collide_list_nb, which evaluates the same four inequalities in the two body frames) must be CONSERVATIVE: a
pair it rejects has no vertex of either box strictly inside the other, which is all bb_contact looks for.  This is a
NumPy restatement of the inequalities in float32 (the kernels' precision, fused or not) against the vertex criterion
in float64, on random and on near-touching configurations.  The device code itself is pinned by the bit-identical
trajectory comparison against the plain pair loops (tests/diag_traj.py --cmp)."""
import numpy as np

SAT_EPS = np.float32(1e-5)


def separated_f32(ax, ay, tha, ahx, ahy, bx, by, thb, bhx, bhy):
  f = np.float32
  ax, ay, ahx, ahy, bx, by, bhx, bhy = (np.asarray(v, f) for v in (ax, ay, ahx, ahy, bx, by, bhx, bhy))
  ca, sa, cb, sb = np.cos(tha).astype(f), np.sin(tha).astype(f), np.cos(thb).astype(f), np.sin(thb).astype(f)
  c, s = np.abs(ca * cb + sa * sb), np.abs(sa * cb - ca * sb)
  tx, ty = bx - ax, by - ay
  ta_x, ta_y = ca * tx + sa * ty, ca * ty - sa * tx
  tb_x, tb_y = cb * tx + sb * ty, cb * ty - sb * tx
  return ((np.abs(ta_x) > ahx + bhx * c + bhy * s + SAT_EPS) | (np.abs(ta_y) > ahy + bhx * s + bhy * c + SAT_EPS) |
          (np.abs(tb_x) > bhx + ahx * c + ahy * s + SAT_EPS) | (np.abs(tb_y) > bhy + ahx * s + ahy * c + SAT_EPS))


def any_vertex_inside(px, py, thp, phx, phy, qx, qy, thq, qhx, qhy):
  """bb_contact's criterion (verts_inside_mask): a vertex of P strictly inside Q, float64."""
  cp, sp, cq, sq = np.cos(thp), np.sin(thp), np.cos(thq), np.sin(thq)
  hit = np.zeros(len(px), bool)
  for sx, sy in ((1, 1), (-1, 1), (-1, -1), (1, -1)):
    vx, vy = px + cp * sx * phx - sp * sy * phy, py + sp * sx * phx + cp * sy * phy
    wx, wy = vx - qx, vy - qy
    lx, ly = cq * wx + sq * wy, -sq * wx + cq * wy
    hit |= (qhx - np.abs(lx) > 0) & (qhy - np.abs(ly) > 0)
  return hit


def _check(ax, ay, tha, ahx, ahy, bx, by, thb, bhx, bhy):
  sep = separated_f32(ax, ay, tha, ahx, ahy, bx, by, thb, bhx, bhy)
  touch = any_vertex_inside(ax, ay, tha, ahx, ahy, bx, by, thb, bhx, bhy) | \
          any_vertex_inside(bx, by, thb, bhx, bhy, ax, ay, tha, ahx, ahy)
  assert not (sep & touch).any(), 'the cull rejected a pair with a vertex inside'
  return sep, touch


def test_sat_cull_is_conservative_random():
  rs = np.random.RandomState(0)
  n = 400_000
  # sizes of the reference's geoms: car bumpers (.1 x .01), wheels, vases (.1), box parts (.1, .2), rod (.08 x .3)
  ah = rs.choice([0.01, 0.025, 0.03, 0.05, 0.1], (n, 2))
  bh = rs.choice([0.08, 0.1, 0.15, 0.2, 0.3], (n, 2))
  ax, ay = rs.uniform(-2, 2, (2, n))
  r = rs.uniform(0, 0.6, n); phi = rs.uniform(0, 2 * np.pi, n)
  bx, by = ax + r * np.cos(phi), ay + r * np.sin(phi)
  tha, thb = rs.uniform(-np.pi, np.pi, (2, n))
  sep, touch = _check(ax, ay, tha, ah[:, 0], ah[:, 1], bx, by, thb, bh[:, 0], bh[:, 1])
  assert 0.2 < sep.mean() < 0.95 and touch.mean() > 0.05   # both outcomes are exercised
  # and it is tight: pairs that pass the cull but have no vertex inside (edge-edge crossings, near misses) are few
  assert ((~sep) & (~touch)).mean() < 0.25


def test_sat_cull_at_grazing_contact():
  """A thin bumper pressed flat against a box face by a hair (|depth| around the cull's margin and far below it)."""
  rs = np.random.RandomState(1)
  n = 200_000
  thb = rs.uniform(-np.pi, np.pi, n)
  bhx = bhy = np.full(n, 0.2)
  ahx, ahy = np.full(n, 0.1), np.full(n, 0.01)
  depth = rs.choice([1e-7, 1e-6, 5e-6, 2e-5, 1e-4], n) * rs.choice([-1, 1], n)   # + = penetrating
  slide = rs.uniform(-0.25, 0.25, n)
  bx, by = rs.uniform(-2, 2, (2, n))
  # A sits on B's +x face, long side along the face, its inner side `depth` inside
  cx, cy = 0.2 + 0.01 - depth, slide
  ax, ay = bx + np.cos(thb) * cx - np.sin(thb) * cy, by + np.sin(thb) * cx + np.cos(thb) * cy
  tha = thb + np.pi / 2 + rs.uniform(-1e-3, 1e-3, n)
  _check(ax, ay, tha, ahx, ahy, bx, by, thb, bhx, bhy)
