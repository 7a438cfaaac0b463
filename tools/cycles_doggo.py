"""GPU: per-section clock profile of doggo_physics (library built with -DSAG_CYCLES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
NAMES = ['load', 'kinematics', 'composite + RNEA bodies', 'tau', 'cholesky', 'M^-1 + qacc0', 'rows: limits + floor',
         'rows: world objects', 'rows finish (W, A)', 'PGS', 'after PGS (qacc, touch)', 'planar world + integrate', 'store', 'CRBA forces + RNEA bias', 'CRBA rows']
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
task = sys.argv[2] if len(sys.argv) > 2 else 'multitask'
run = bench.DeviceRun(task, envs, 0, 0, robot='doggo')
run.burn_in(int(sys.argv[3]) if len(sys.argv) > 3 else 10)
run.ctx.debug_cycles(reset=True)
run.timing(True)
K = 10
run.run(K); run.wait()
ms, _ = run.kernel_time_ms()
c = run.ctx.debug_cycles().astype(np.float64)
waves = c[1, 15]
tot = c[1, :len(NAMES)].sum()
print(f'doggo {task} {envs} envs: {ms:.3f} ms/step, {tot / waves:.0f} ticks/wavefront in doggo_physics')
for k, n in enumerate(NAMES):
  print(f'   {n:28s} {c[1, k] / waves:12.0f}  {100 * c[1, k] / tot:5.1f} %')
print('rows per evaluation (max of the wavefront\'s two envs), buckets of 4 rows from 0: ' + ' '.join(f'{int(v)}' for v in c[0]))
print('wavefront totals, buckets of 2^19 ticks from 0:                                  ' + ' '.join(f'{int(v)}' for v in c[2]))
