"""Summarise rocprofv3 csv output (kernel stats + PMC) into a few lines."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def find(pat):
  return sorted(glob.glob(os.path.join(out, pat), recursive=True))
for f in find('trace/**/*kernel_stats.csv'):
  print('== kernel stats', os.path.relpath(f, out))
  for r in csv.DictReader(open(f)):
    print({k: r[k] for k in r if k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')})
for d in ('pmc1', 'pmc2', 'pmc3', 'pmc4', 'pmc5'):
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
