"""Builds libsag.so (HIP, gfx950) in-tree.  `python -m safe_adaptation_gym_amd.build`."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsag.so')
SOURCES = ['sag_api.hip', 'sag_sampler.cpp']
# every header under csrc/ (sag_device.hpp includes the Doggo, cooperative-Doggo and render headers) + the ABI
HEADERS = sorted(os.path.basename(h) for h in glob.glob(os.path.join(CSRC, '*.hpp'))) + [
    os.path.join('..', '..', 'include', 'sag.h')]


def hipcc():
  for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found: the HIP extension cannot be built')


def up_to_date():
  if not os.path.exists(LIB):
    return False
  t = os.path.getmtime(LIB)
  deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
  return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, extra=()):
  if not force and up_to_date():
    return LIB
  cmd = [
      hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
      '-fno-fast-math', '-ffp-contract=off', '-Wall', '-Wno-unused-function', '-pthread', *extra, '-o', LIB
  ] + [os.path.join(CSRC, s) for s in SOURCES]
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd)
  return LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv, verbose=True,
        extra=['-Rpass-analysis=kernel-resource-usage'] if '--usage' in sys.argv else ())
  print(LIB)
