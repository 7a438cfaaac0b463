"""Host-side errors (reference utils.py:6).  Placement sampling itself is native (csrc/sag_sampler.cpp)."""


class ResamplingError(AssertionError):
  """Raised when no valid layout / goal can be sampled (reference utils.py:6)."""
