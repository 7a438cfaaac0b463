"""Steady-state figures from a tools/prof.sh output directory: only the LAST `n` steps of the run
(bench.py ages the population for 200 steps first; young steps are cheaper and would bias a mean
over all dispatches).
  python tools/prof_steady.py gpurun_out/prof_<tag> [n=30] [envs] [key=point] -> text; the traffic figure is merged
into profiles/traffic.json under configs[key], keyed to the hash of the device sources (bench.source_sha16): bench.py
quotes it as roofline.traffic only while the sources are the ones it was measured on."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
ckey = sys.argv[4] if len(sys.argv) > 4 else 'point'
ALG = {'point': 892, 'car': 804, 'lidar_cost': 376}.get(ckey, 892)
RID = {'point': 0, 'car': 1}.get(ckey, 0)
KEYS = ('k_lidar_cost',) if ckey == 'lidar_cost' else ('k_compact', 'k_step_quiet', 'k_step_busy')


def key_of(name):
  """The step kernels of THIS robot only (a FULL profile holds the Point headline and the Car config); k_compact is shared."""
  for k in KEYS:
    if k in name and (k in ('k_compact', 'k_lidar_cost') or f'{k}<{RID},' in name):
      return k
  return None


def steps_of(seq):
  """seq: (key, value) in dispatch order -> one dict per step of this robot: the compaction launch is the last one
  before the step's quiet kernel (other configs' compactions in between are dropped)."""
  out, last_c, busy = [], None, []
  for k, v in seq:
    if k == 'k_compact':
      last_c = v
    elif k == 'k_step_quiet':
      out.append({'k_compact': last_c, 'k_step_quiet': v})
    elif k == 'k_step_busy':
      busy.append(v)
    else:
      out.append({k: v})
  for st, b in zip(out, busy):
    st['k_step_busy'] = b
  return [st for st in out if all(k in st and st[k] is not None for k in KEYS)]


# kernel trace: per step = the three launches; span = first start .. last end (the two step kernels overlap)
tr = glob.glob(os.path.join(out, 'trace/**/*kernel_trace.csv'), recursive=True)
if tr:
  seq = sorted(((int(r['Start_Timestamp']), key_of(r['Kernel_Name']), int(r['End_Timestamp'])) for r in csv.DictReader(open(tr[0]))
                if key_of(r['Kernel_Name'])), key=lambda x: x[0])
  st = steps_of([(k, (a, b)) for a, k, b in seq])
  steps = len(st)
  span, dur = [], defaultdict(list)
  for s in range(steps - n, steps):
    t0 = min(st[s][k][0] for k in KEYS)
    t1 = max(st[s][k][1] for k in KEYS)
    span.append(t1 - t0)
    for k in KEYS:
      dur[k].append(st[s][k][1] - st[s][k][0])
  print(f'kernel trace, last {n} of {steps} steps: step span (compact start .. last kernel end) '
        f'{sum(span) / n / 1e3:.1f} us; ' + '; '.join(f'{k} {sum(dur[k]) / n / 1e3:.1f} us' for k in KEYS))
  print(f'  algorithmic {ALG} B x {envs} envs / span = {ALG * envs / (sum(span) / n) :.0f} GB/s = '
        f'{ALG * envs / (sum(span) / n) / 8000:.4f} of 8 TB/s')

res = {}
for d, cname in (('pmc3', 'FETCH_SIZE'), ('pmc4', 'WRITE_SIZE')):
  f = glob.glob(os.path.join(out, f'{d}/**/*counter_collection.csv'), recursive=True)
  if not f:
    continue
  seq = sorted(((int(r['Dispatch_Id']), key_of(r['Kernel_Name']), float(r['Counter_Value'])) for r in csv.DictReader(open(f[0]))
                if key_of(r['Kernel_Name']) and r['Counter_Name'] == cname), key=lambda x: x[0])
  st = steps_of([(k, v) for _, k, v in seq])
  res[cname] = {k: sum(x[k] for x in st[-n:]) / n for k in KEYS}
  print(f'{cname} (KB per launch, last {n} launches):', {k: round(v, 1) for k, v in res[cname].items()})
if len(res) == 2:
  fetch = 2 * sum(res['FETCH_SIZE'].values()) * 1024   # gfx950: 128-B requests counted as 64 B
  write = sum(res['WRITE_SIZE'].values()) * 1024
  print(f'HBM traffic per step: fetch {fetch / 1e6:.1f} MB (x2 corrected) + write {write / 1e6:.1f} MB = '
        f'{(fetch + write) / envs:.1f} B per env-step (algorithmic {ALG})')
  entry = {'envs': envs, 'last_launches': n, 'FETCH_SIZE_KB_per_step': res['FETCH_SIZE'], 'WRITE_SIZE_KB_per_step': res['WRITE_SIZE'],
           'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact',
           'bytes_per_env_step': (fetch + write) / envs, 'source': os.path.basename(os.path.normpath(out))}
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  sha = bench.source_sha16()
  path = os.path.join(root, 'profiles', 'traffic.json')
  try:
    doc = json.load(open(path))
    if doc.get('src_sha16') != sha:
      doc = {}
  except (OSError, ValueError):
    doc = {}
  doc.setdefault('configs', {})[ckey] = entry
  doc['src_sha16'] = sha
  json.dump(doc, open(path, 'w'), indent=1)
  print(json.dumps(entry, indent=1))
  print('merged into', path, 'for sources', sha)
