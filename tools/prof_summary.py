"""Summarise rocprofv3 csv output (kernel stats + PMC) into a few lines.
  python tools/prof_summary.py <prof dir> [--json profiles/<summary file name>]
--json: also write profiles/kernels.json = {src_sha16, bench_py_sha16, summary, kernels: {name: {calls, avg_us}},
doggo_flops_per_env_step: {fp64, fp32}} - the machine-readable index of the summary that bench.py checks every
roofline block against (a block whose kernel is not in it is refused)."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
out = sys.argv[1]
json_for = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
kernels, flops = {}, {}
def find(pat):
  return sorted(glob.glob(os.path.join(out, pat), recursive=True))
for f in find('trace/**/*kernel_stats.csv'):
  print('== kernel stats', os.path.relpath(f, out))
  for r in csv.DictReader(open(f)):
    print({k: r[k] for k in r if k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')})
    kernels[r['Name']] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3}
for d in ('pmc1', 'pmc2', 'pmc3', 'pmc4', 'pmc5', 'pmc6'):
  for f in find(f'{d}/**/*counter_collection.csv'):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    vg = {}
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name'][:60]
      acc[k][r['Counter_Name']] += float(r['Counter_Value'])
      cnt[(k, r['Counter_Name'])] += 1
      vg[k] = (r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'), r.get('Scratch_Size'), r.get('Grid_Size'), r.get('Workgroup_Size'))
    print('== pmc', d)
    for k in acc:
      if not any(t in k for t in ('step', 'lidar', 'compact', 'doggo', 'render')):
        continue
      print(k, 'vgpr/sgpr/lds/scratch/grid/wg', vg[k])
      for c, v in acc[k].items():
        print(f'   {c}: mean per dispatch {v / cnt[(k, c)]:.4g}  (n={cnt[(k, c)]})')
      if 'k_doggo_physics' in k and d in ('pmc5', 'pmc6'):
        # issued lane-operations per env-step: (ADD + MUL + 2 FMA + TRANS) x 64 lanes / envs of the dispatch (grid = envs / 2 workgroups of 64)
        m = {c.rsplit('_', 2)[-2]: v / cnt[(k, c)] for c, v in acc[k].items()}
        envs = int(vg[k][4]) // 32 if vg[k][4] else 4096
        flops['fp64' if d == 'pmc5' else 'fp32'] = (m.get('ADD', 0) + m.get('MUL', 0) + 2 * m.get('FMA', 0) + m.get('TRANS', 0)) * 64 / envs
        print(f'   -> {flops["fp64" if d == "pmc5" else "fp32"]:.4g} issued {"fp64" if d == "pmc5" else "fp32"} lane-operations per env-step ({envs} envs per dispatch)')

if json_for:
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  doc = {'src_sha16': bench.source_sha16(), 'bench_py_sha16': hashlib.sha256(open(os.path.join(root, 'bench.py'), 'rb').read()).hexdigest()[:16],
         'summary': json_for, 'command': 'FULL=1 tools/prof.sh (rocprofv3 --kernel-trace --stats + PMC passes over the default bench.py)',
         'kernels': kernels, 'doggo_flops_per_env_step': flops if len(flops) == 2 else None}
  json.dump(doc, open(os.path.join(root, 'profiles', 'kernels.json'), 'w'), indent=1)
  print('== wrote profiles/kernels.json for sources', doc['src_sha16'], 'with', len(kernels), 'kernels')
