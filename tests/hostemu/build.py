"""Builds tests/hostemu/_build/libsag_hostemu*.so: the library's OWN sources (csrc/sag_api.hip with every
kernel it includes, sag_sampler.cpp) compiled for x86-64 against tests/hostemu/hip/hip_runtime.h, with
sanitizers.  TEST INFRASTRUCTURE: a checker of the device source (indexing, uninitialised reads, undefined
behaviour), never a product path - see the header of hip/hip_runtime.h.

  python tests/hostemu/build.py [--cc gcc|clang] [--pattern] [--no-san] [--tag NAME] [-D...]
    --pattern   clang only: -ftrivial-auto-var-init=pattern (uninitialised locals become NaN / 0xAA..)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'safe_adaptation_gym_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def lib_path(cc='gcc', pattern=False, san=True, tag=''):
  return os.path.join(OUT, 'libsag_hostemu_%s%s%s%s.so' % (cc, '_pattern' if pattern else '', '' if san else '_nosan',
                                                           '_' + tag if tag else ''))


def build(cc='gcc', pattern=False, san=True, extra=(), force=False, verbose=False, tag=''):
  os.makedirs(OUT, exist_ok=True)
  lib = lib_path(cc, pattern, san, tag)
  deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, 'hip', 'hip_runtime.h'),
                                                              os.path.join(ROOT, 'include', 'sag.h'), __file__]
  if not force and not extra and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
    return lib
  cmd = [CLANG if cc == 'clang' else 'g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-pthread', '-ffp-contract=off',
         '-fno-omit-frame-pointer', '-DSAG_HOSTEMU=1', '-I', HERE, '-Wno-attributes', '-Wno-unknown-attributes',
         '-Wno-unused-function', '-Wno-unused-value']
  if san:
    cmd += ['-fsanitize=address,undefined', '-fsanitize=float-cast-overflow']
    if cc == 'clang':
      cmd += ['-shared-libsan']
  if pattern:
    assert cc == 'clang', '--pattern needs clang'
    cmd += ['-ftrivial-auto-var-init=pattern']
  cmd += list(extra) + ['-x', 'c++', os.path.join(CSRC, 'sag_api.hip'), os.path.join(CSRC, 'sag_sampler.cpp'), '-o', lib]
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd)
  return lib


def preload(cc='gcc'):
  """The sanitizer runtime that must be LD_PRELOADed into python before the library is opened."""
  if cc == 'clang':
    return subprocess.check_output([CLANG, '-print-file-name=libclang_rt.asan-x86_64.so'], text=True).strip()
  return subprocess.check_output(['g++', '-print-file-name=libasan.so'], text=True).strip()


if __name__ == '__main__':
  a = sys.argv[1:]
  cc = a[a.index('--cc') + 1] if '--cc' in a else 'gcc'
  tag = a[a.index('--tag') + 1] if '--tag' in a else ''
  print(build(cc, '--pattern' in a, '--no-san' not in a, [x for x in a if x.startswith('-D')], force=True, verbose=True, tag=tag))
