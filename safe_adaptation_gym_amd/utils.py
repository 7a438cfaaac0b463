"""Host-side placement helpers (reference utils.py:6-70,118-119 semantics)."""
import numpy as np


class ResamplingError(AssertionError):
  """Raised when no valid layout / goal can be sampled (reference utils.py:6)."""


def random_rot(rs):
  return rs.uniform(0, 2 * np.pi)


def shrink(rect, keepout):
  xmin, ymin, xmax, ymax = rect
  return xmin + keepout, ymin + keepout, xmax - keepout, ymax - keepout


def grow(rect, scale=1.01):
  return tuple(np.asarray(rect) * scale)


def draw_placement(rs, placements, extents, keepout):
  """One (x, y) draw.  `placements` None -> the task extents; else a list of rectangles,
  area-weighted when more than one survives the keepout shrink."""
  if placements is None:
    rect = shrink(extents, keepout)
  else:
    ok = []
    for r in placements:
      x0, y0, x1, y1 = shrink(r, keepout)
      if x0 > x1 or y0 > y1:
        continue
      ok.append((x0, y0, x1, y1))
    assert len(ok), 'Failed to find any placements with satisfy keepout'
    if len(ok) == 1:
      rect = ok[0]
    else:
      areas = np.array([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in ok])
      rect = ok[rs.choice(len(ok), p=areas / np.sum(areas))]
  x0, y0, x1, y1 = rect
  return np.array([rs.uniform(x0, x1), rs.uniform(y0, y1)])
