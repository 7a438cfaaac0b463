"""Per-kernel resource table of libsag.so: `python tools/usage.py [-DSWITCH ...]` builds a throw-away variant with
-Rpass-analysis=kernel-resource-usage and prints name, SGPRs (+ spilled), VGPRs (+ spilled), AGPRs, scratch bytes per
lane, occupancy (wavefronts per SIMD) and LDS bytes per workgroup.  Warns when a kernel spills SGPRs to VGPR lanes (the
path the round-1 Doggo miscompile sat on: DESIGN.md 3.4)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safe_adaptation_gym_amd import build as B  # noqa: E402

extra = [a for a in sys.argv[1:] if a.startswith('-D')]
cmd = [B.hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-fast-math', '-ffp-contract=off', '-Wno-unused-function',
       '-Rpass-analysis=kernel-resource-usage', *extra, '-c', os.path.join(B.CSRC, 'sag_api.hip'), '-o', '/tmp/sag_usage.o']
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
  m = re.search(r'remark: .*?Function Name: (\S+)', line)
  if m:
    cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
    rows.append(cur)
    continue
  m = re.search(r'remark:\s+(.+?): (\d+) \[', line)
  if m and cur is not None:
    cur[m.group(1).strip()] = int(m.group(2))
print(f'{"kernel":70s} {"SGPR":>5s} {"spill":>5s} {"VGPR":>5s} {"spill":>5s} {"AGPR":>5s} {"scratch":>8s} {"occ":>4s} {"LDS":>7s}')
for r in rows:
  nm = re.sub(r'\(.*', '', r['name']).replace('sag::', '')
  print(f'{nm[:70]:70s} {r.get("TotalSGPRs", r.get("SGPRs", 0)):5d} {r.get("SGPRs Spill", 0):5d} {r.get("VGPRs", 0):5d} {r.get("VGPRs Spill", 0):5d} '
        f'{r.get("AGPRs", 0):5d} {r.get("ScratchSize [bytes/lane]", 0):8d} {r.get("Occupancy [waves/SIMD]", 0):4d} {r.get("LDS Size [bytes/block]", 0):7d}')
bad = [r['name'] for r in rows if r.get('SGPRs Spill', 0) > 0]
if bad:
  print(f'WARNING: {len(bad)} kernels spill SGPRs (to VGPR lanes): ' + ', '.join(re.sub(r"\(.*", "", b) for b in bad[:8]))
