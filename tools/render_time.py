"""GPU: time of sag_render_rgb_device for 4096 Doggo / haul_box envs (BASELINE config 5), ms per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
robot = sys.argv[1] if len(sys.argv) > 1 else 'doggo'
task = sys.argv[2] if len(sys.argv) > 2 else 'haul_box'
r5 = bench.DeviceRun(task, 4096, 0, 0, robot=robot)
r5.burn_in(20)
d_img = r5.ctx.dev_alloc(4096 * 64 * 64 * 3)
r5.ctx.render_rgb_device(d_img); r5.wait()
t0 = time.perf_counter()
for _ in range(20):
  r5.ctx.render_rgb_device(d_img)
r5.wait()
print(f'{robot}/{task}: render {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per 4096 images')
