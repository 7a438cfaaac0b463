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
EXTRA = {}   # per-source extra flags
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


def build(force=False, verbose=False, extra=(), out=None):
  """out: path of a VARIANT library (tools/ab.sh, profiling builds with -D switches) - always rebuilt, own object dir."""
  if out is None and not force and up_to_date():
    return LIB
  lib = out or LIB
  base = [hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-fast-math', '-ffp-contract=off', '-Wall',
          '-Wno-unused-function', '-pthread', *extra]
  objdir = os.path.join(HERE, 'build', os.path.basename(lib) if out else '')
  os.makedirs(objdir, exist_ok=True)
  jobs = []
  for src in SOURCES:   # the translation units compile side by side
    obj = os.path.join(objdir, src + '.o')
    cmd = base + EXTRA.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
    if verbose:
      print(' '.join(cmd))
    jobs.append((cmd, obj, subprocess.Popen(cmd)))
  for cmd, obj, proc in jobs:
    if proc.wait() != 0:
      raise subprocess.CalledProcessError(proc.returncode, cmd)
  link = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', lib] + [j[1] for j in jobs]
  if verbose:
    print(' '.join(link))
  subprocess.check_call(link)
  return lib


if __name__ == '__main__':
  # python -m safe_adaptation_gym_amd.build [--force] [--usage] [--out libsag_<name>.so] [-DSWITCH ...]
  argv = sys.argv[1:]
  out = os.path.join(HERE, argv[argv.index('--out') + 1]) if '--out' in argv else None
  extra = [a for a in argv if a.startswith('-D') or a.startswith('-mllvm') or a.startswith('-amdgpu')]
  if '--usage' in argv:
    extra.append('-Rpass-analysis=kernel-resource-usage')
  print(build(force='--force' in argv, verbose=True, extra=extra, out=out))
