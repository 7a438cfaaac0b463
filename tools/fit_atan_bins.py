"""CPU: coefficients of angle_bins() in csrc/sag_device.hpp - odd minimax polynomial for
atan(q) * 8/pi on [0, 1] (Lawson iteration) and its error when evaluated in fp32 as the kernel does."""
import numpy as np


def fit(nterms):
  q = np.sort(np.cos(np.linspace(0, np.pi / 2, 40001)))
  q = q[q > 0]
  z = q * q
  A = np.stack([z**k for k in range(nterms)], 1)
  y = np.arctan(q) / q * (8 / np.pi)
  w = np.ones_like(q)
  for _ in range(400):
    c = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
    e = np.abs(A @ c - y) * q
    w = w * (e / e.max() + 1e-3)
    w /= w.max()
  return c, e.max()


def fma32(a, b, c):
  """fp32 fused multiply-add: the product of two fp32 is exact in fp64; one rounding to fp32 (double rounding through
  fp64 is negligible for an error bound).  The kernels evaluate both polynomials with explicit fmaf since round 2."""
  return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32) if np.isscalar(b) or np.ndim(b) == 0 \
      else (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def angle_bins_f32(ex, ey, c):
  f = np.float32
  c = c.astype(f)
  ax, ay = np.abs(ex), np.abs(ey)
  mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
  q = (mn * (f(1) / mx).astype(f)).astype(f)
  z = (q * q).astype(f)
  t = np.full(len(ex), c[-1], f)
  for k in range(len(c) - 2, -1, -1):
    t = fma32(t, z, c[k])
  t = (t * q).astype(f)
  t = np.where(ay > ax, (f(4) - t).astype(f), t)
  t = np.where(ex < 0, (f(8) - t).astype(f), t)
  return np.where(ey < 0, (f(16) - t).astype(f), t)


def check_fast():
  c, e = fit(8)
  print('polynomial error (bins):', e)
  print(', '.join('%.9ef' % x for x in c))
  rng = np.random.RandomState(1)
  n = 4_000_000
  ex = rng.uniform(-4, 4, n).astype(np.float32)
  ey = rng.uniform(-4, 4, n).astype(np.float32)
  ex[:100000] *= np.float32(1e-4)
  ey[100000:200000] *= np.float32(1e-4)
  a = np.arctan2(ey.astype(np.float64), ex.astype(np.float64))
  a = np.where(a < 0, a + 2 * np.pi, a)
  print('fp32 evaluation, max error (bins):', np.abs(angle_bins_f32(ex, ey, c) - a * 16 / (2 * np.pi)).max())


def angle_bin_alias_f32(ex, ey, c):
  """(bin, alias) as lidar_bin_alias() in csrc/sag_device.hpp computes them: the polynomial gives the angle inside
  the octant in [0, 2] bins; whole bins and fraction are split THERE and carried through the three reflections as
  (C - 1 - bin, 1 - fraction), so the fraction never sees the 1e-6 ulp of a number near 16."""
  f = np.float32
  c = c.astype(f)
  ax, ay = np.abs(ex), np.abs(ey)
  mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
  q = (mn * (f(1) / mx).astype(f)).astype(f)   # v_rcp_f32 (1 ulp) + multiply
  z = (q * q).astype(f)
  t = np.full(len(ex), c[-1], f)
  for k in range(len(c) - 2, -1, -1):
    t = fma32(t, z, c[k])
  t = (t * q).astype(f)
  b = np.minimum(np.floor(t), 1).astype(np.int32)      # t in [0, 2]: bin 0 or 1 (t == 2 -> bin 1, fraction 1)
  fr = (t - b.astype(f)).astype(f)
  for cond, C in ((ay > ax, 4), (ex < 0, 8), (ey < 0, 16)):
    nb = np.where(fr > 0, C - 1 - b, C - b)
    nf = np.where(fr > 0, (f(1) - fr).astype(f), f(0))
    b, fr = np.where(cond, nb, b), np.where(cond, nf, fr)
  return b, fr


def check_hi(nterms=11):
  c, e = fit(nterms)
  print(f'{nterms} terms: polynomial error (bins):', e)
  print(', '.join('%.9ef' % x for x in c))
  rng = np.random.RandomState(2)
  n = 4_000_000
  ex = rng.uniform(-4, 4, n).astype(np.float32)
  ey = rng.uniform(-4, 4, n).astype(np.float32)
  ex[:100000] *= np.float32(1e-4)
  ey[100000:200000] *= np.float32(1e-4)
  a = np.arctan2(ey.astype(np.float64), ex.astype(np.float64))
  a = np.where(a < 0, a + 2 * np.pi, a) * 16 / (2 * np.pi)
  b, fr = angle_bin_alias_f32(ex, ey, c)
  err = np.abs((b + fr.astype(np.float64)) - a)
  err = np.minimum(err, 16 - err)
  print('fp32 (bin, alias) evaluation, max error (bins):', err.max())


if __name__ == '__main__':
  check_fast()    # angle_bins(): the fused step's estimate (8 terms)
  check_hi(11)    # lidar_bin_alias(): k_lidar_cost's (11 terms, octant-local fraction)
