// rgb_observation (safe_adaptation_gym.py:122-126): the 64 x 64 x 3 uint8 image of
// the robot's `vision` camera, ray-cast on the device.  The reference renders with MuJoCo's OpenGL
// rasteriser ("parity unpinned"); the specification of this image - camera from the robot XML
// (point.xml:14, car.xml:14, doggo.xml:13), scene of mujoco_bridge.py:88-123, geoms and colours of
// consts.py / primitive_objects.py / the task files, one blend layer, Lambert + headlight shade,
// own body not drawn - is written out in DESIGN.md "rgb_observation".  fp64: a pixel is a hard
// decision (which surface, which checker square, rounding to 8 bits), like a lidar bin.
// One workgroup per env: lane 0 builds the scene (<= 40 geoms) in LDS, 256 threads x 16 pixels.
#pragma once

namespace sag {

struct RCam { double o[3], X[3], Y[3], Z[3], tanh_; };
struct RObj { int kind; double c[3], a, b, h, cs, sn, rgb[3], alpha; };  // 0 box, 1 cylinder, 2 sphere
constexpr int R_MAXOBJ = 40, R_W = 64, R_H = 64;

__device__ inline void r_norm(double* v) {
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n; v[1] /= n; v[2] /= n;
}

__device__ inline void r_camera(int robot, const double* R, const double* p, RCam& cam) {
  const double CPOS[3][3] = {{0, 0, .15}, {0, .1, .2}, {.125, 0, .2}};
  const double CX[3][3] = {{0, -1, 0}, {-1, 0, 0}, {0, -1, 0}};
  const double CY[3][3] = {{.4, 0, 1}, {0, -.4, 1}, {.4, 0, 1}};
  const double FOVY[3] = {90, 45, 100};
  double x[3], y[3], z[3];
  for (int k = 0; k < 3; k++) { x[k] = CX[robot][k]; y[k] = CY[robot][k]; }
  r_norm(x);
  const double d = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
  for (int k = 0; k < 3; k++) y[k] -= d * x[k];
  r_norm(y);
  z[0] = x[1] * y[2] - x[2] * y[1]; z[1] = x[2] * y[0] - x[0] * y[2]; z[2] = x[0] * y[1] - x[1] * y[0];
  for (int k = 0; k < 3; k++) {
    cam.o[k] = p[k] + R[3 * k] * CPOS[robot][0] + R[3 * k + 1] * CPOS[robot][1] + R[3 * k + 2] * CPOS[robot][2];
    cam.X[k] = R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2];
    cam.Y[k] = R[3 * k] * y[0] + R[3 * k + 1] * y[1] + R[3 * k + 2] * y[2];
    cam.Z[k] = R[3 * k] * z[0] + R[3 * k + 1] * z[1] + R[3 * k + 2] * z[2];
  }
  cam.tanh_ = tan(0.5 * FOVY[robot] * PI_D / 180);
}

// nearest intersection t > 1e-6 of the ray o + t d with the geom; n = surface normal there
__device__ inline bool r_hit(const RObj& ob, const double* o, const double* d, double& tout, double* n) {
  if (ob.kind == 2) {
    const double oc[3] = {o[0] - ob.c[0], o[1] - ob.c[1], o[2] - ob.c[2]};
    const double b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
    const double c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - ob.a * ob.a;
    const double disc = b * b - c;
    if (disc <= 0) return false;
    const double t = -b - sqrt(disc);
    if (t <= 1e-6) return false;
    tout = t;
    for (int k = 0; k < 3; k++) n[k] = (oc[k] + t * d[k]) / ob.a;
    return true;
  }
  const double ox = o[0] - ob.c[0], oy = o[1] - ob.c[1], oz = o[2] - ob.c[2];
  const double lo[3] = {ob.cs * ox + ob.sn * oy, -ob.sn * ox + ob.cs * oy, oz};
  const double ld[3] = {ob.cs * d[0] + ob.sn * d[1], -ob.sn * d[0] + ob.cs * d[1], d[2]};
  double ln[3] = {0, 0, 0}, best = 1e30;
  bool found = false;
  if (ob.kind == 0) {
    const double hh[3] = {ob.a, ob.b, ob.h};
    double t0 = -1e30, t1 = 1e30, sg0 = 0;
    int ax0 = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (ld[k] == 0) { if (fabs(lo[k]) > hh[k]) return false; continue; }
      double ta = (-hh[k] - lo[k]) / ld[k], tb = (hh[k] - lo[k]) / ld[k], s = -1;
      if (ta > tb) { const double t = ta; ta = tb; tb = t; s = 1; }
      if (ta > t0) { t0 = ta; ax0 = k; sg0 = s; }
      if (tb < t1) t1 = tb;
    }
    if (t0 >= t1 || t0 <= 1e-6 || ax0 < 0) return false;
    best = t0; found = true;
    ln[0] = ax0 == 0 ? sg0 : 0; ln[1] = ax0 == 1 ? sg0 : 0; ln[2] = ax0 == 2 ? sg0 : 0;
  } else {
    const double a = ld[0] * ld[0] + ld[1] * ld[1], b = lo[0] * ld[0] + lo[1] * ld[1];
    const double c = lo[0] * lo[0] + lo[1] * lo[1] - ob.a * ob.a;
    if (a > 0) {
      const double disc = b * b - a * c;
      if (disc > 0) {
        const double t = (-b - sqrt(disc)) / a, z = lo[2] + t * ld[2];
        if (t > 1e-6 && fabs(z) <= ob.h) {
          best = t; found = true;
          ln[0] = (lo[0] + t * ld[0]) / ob.a; ln[1] = (lo[1] + t * ld[1]) / ob.a; ln[2] = 0;
        }
      }
    }
    if (ld[2] != 0) {
      const double s = ld[2] < 0 ? 1 : -1, t = (s * ob.h - lo[2]) / ld[2];
      const double x = lo[0] + t * ld[0], y = lo[1] + t * ld[1];
      if (t > 1e-6 && t < best && x * x + y * y <= ob.a * ob.a) { best = t; found = true; ln[0] = 0; ln[1] = 0; ln[2] = s; }
    }
  }
  if (!found) return false;
  tout = best;
  n[0] = ob.cs * ln[0] - ob.sn * ln[1]; n[1] = ob.sn * ln[0] + ob.cs * ln[1]; n[2] = ln[2];
  return true;
}

__device__ inline void r_shade(const double* rgb, const double* n, const double* d, double* out) {
  const double nz = n[2] > 0 ? n[2] : 0;
  double hd = -(n[0] * d[0] + n[1] * d[1] + n[2] * d[2]);
  if (hd < 0) hd = 0;
  double s = 0.4 + 0.6 * nz + 0.2 * hd;
  if (s > 1) s = 1;
  for (int k = 0; k < 3; k++) out[k] = rgb[k] * s;
}

__device__ inline void r_obj(RObj* ob, int& n, int kind, double x, double y, double z, double a, double b, double h,
                             double yaw, double cr, double cg, double cb, double al) {
  RObj& q = ob[n++];
  q.kind = kind; q.c[0] = x; q.c[1] = y; q.c[2] = z; q.a = a; q.b = b; q.h = h;
  q.cs = cos(yaw); q.sn = sin(yaw); q.rgb[0] = cr; q.rgb[1] = cg; q.rgb[2] = cb; q.alpha = al;
}

// out: [N][64][64][3] uint8
__global__ __launch_bounds__(256) void k_render_rgb(const float* __restrict__ S, const int32_t* __restrict__ I, int N,
                                                    int robot, uint8_t* __restrict__ out) {
  __shared__ RObj ob[R_MAXOBJ];
  __shared__ RCam cam;
  __shared__ int nob_s;
  const size_t i = blockIdx.x;
  if (threadIdx.x == 0) {
    auto F = [&](int k) { return (double)S[saddr(k, (size_t)N, i)]; };
    const uint32_t meta = (uint32_t)I[iaddr(DI_META, (size_t)N, i)], tstate = (uint32_t)I[iaddr(DI_TSTATE, (size_t)N, i)];
    const int task = meta & 15, nH = meta >> 4 & 15, nV = meta >> 8 & 15, nP = meta >> 12 & 3, nB = meta >> 14 & 7,
              box_kind = meta >> 17 & 3;
    double R[9], p[3];
    if (robot == SAG_ROBOT_DOGGO) {
      DgState D;
      dg_load(D, S, (size_t)N, i);
      dg_quat2mat(D.quat, R);
      p[0] = D.pos[0]; p[1] = D.pos[1]; p[2] = D.pos[2];
    } else {
      const double yaw = F(SAG_F_ROBOT + 2), c = cos(yaw), s = sin(yaw);
      R[0] = c; R[1] = -s; R[2] = 0; R[3] = s; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
      p[0] = F(SAG_F_ROBOT); p[1] = F(SAG_F_ROBOT + 1); p[2] = PT_Z;
    }
    r_camera(robot, R, p, cam);
    int n = 0;
    const double vs = F(SAG_F_VASE_SIZE);
    for (int k = 0; k < nH; k++)
      r_obj(ob, n, 1, F(SAG_F_HAZARDS + 2 * k), F(SAG_F_HAZARDS + 2 * k + 1), 0.02, F(SAG_F_HAZARD_SIZE), 0, 0.01, 0.0, 0, 0, 1, 0.25);
    for (int k = 0; k < nV; k++)
      r_obj(ob, n, 0, F(SAG_F_VASES + 6 * k), F(SAG_F_VASES + 6 * k + 1), vs - 4e-5, vs, vs, vs, F(SAG_F_VASES + 6 * k + 2), 0, 1, 1, 1.0);
    for (int k = 0; k < nP; k++)
      r_obj(ob, n, 1, F(SAG_F_PILLARS + 2 * k), F(SAG_F_PILLARS + 2 * k + 1), 0.5, F(SAG_F_PILLAR_SIZE), 0, 0.5, 0.0, .5, .5, 1, 1.0);
    const bool goal_body = !(task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE || task == SAG_TASK_COLLECT);
    if (goal_body)
      r_obj(ob, n, 1, F(SAG_F_GOAL), F(SAG_F_GOAL + 1), GOAL_Z, GOAL_SIZE, 0, GOAL_SIZE / 2, 0.0, 0, 1, 0,
            task == SAG_TASK_UNSUPERVISED ? 0.1 : 0.25);
    const int gb = tstate & 7, bstate = tstate >> 3 & 1;
    const uint32_t act = tstate >> 11 & 63;
    for (int b = 0; b < nB; b++) {
      int g;
      if (task == SAG_TASK_COLLECT) g = (act >> b & 1) ? 2 : 0;
      else g = bstate == 0 ? 0 : (b == gb ? 2 : 3);
      if (g == 2) r_obj(ob, n, 2, F(SAG_F_BUTTONS + 2 * b), F(SAG_F_BUTTONS + 2 * b + 1), (double)BUTTON_R, (double)BUTTON_R, 0, 0, 0.0, 0, 1, 0, 1.0);
      else r_obj(ob, n, 2, F(SAG_F_BUTTONS + 2 * b), F(SAG_F_BUTTONS + 2 * b + 1), (double)BUTTON_R, (double)BUTTON_R, 0, 0, 0.0, 1, 105.0 / 255, 180.0 / 255, 1.0);
    }
    if (box_kind == SAG_BOX_BOX) r_obj(ob, n, 0, F(SAG_F_BOX), F(SAG_F_BOX + 1), 0.2, 0.2, 0.2, 0.2, F(SAG_F_BOX + 2), 1, 1, 0, 0.25);
    else if (box_kind == SAG_BOX_ROD) r_obj(ob, n, 0, F(SAG_F_BOX), F(SAG_F_BOX + 1), 0.08, 0.08, 0.3, 0.08, F(SAG_F_BOX + 2), 1, 1, 1, 1.0);
    else if (box_kind == SAG_BOX_BALL) r_obj(ob, n, 2, F(SAG_F_BOX), F(SAG_F_BOX + 1), 0.14, 0.14, 0, 0, 0.0, 1, 1, 1, 1.0);
    nob_s = n;
  }
  __syncthreads();
  const int nob = nob_s;
  uint8_t* img = out + i * (size_t)(R_W * R_H * 3);
#pragma unroll 1
  for (int px = threadIdx.x; px < R_W * R_H; px += 256) {
    const int r = px >> 6, c = px & 63;
    const double u = ((c + 0.5) / 32.0 - 1.0) * cam.tanh_, v = (1.0 - (r + 0.5) / 32.0) * cam.tanh_;
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = u * cam.X[k] + v * cam.Y[k] - cam.Z[k];
    r_norm(d);
    double best = 1e30, col[3] = {0, 0, 0}, t;
    bool hit = false;
#pragma unroll 1
    for (int k = 0; k < nob; k++) {
      double n[3];
      if (ob[k].alpha >= 1.0 && r_hit(ob[k], cam.o, d, t, n) && t < best) { best = t; hit = true; r_shade(ob[k].rgb, n, d, col); }
    }
    if (d[2] < 0) {
      const double tf = -cam.o[2] / d[2], fx = cam.o[0] + tf * d[0], fy = cam.o[1] + tf * d[1];
      if (tf > 1e-6 && tf < best && fabs(fx) <= 3.5 && fabs(fy) <= 3.5) {
        const int ix = (int)floor((fx + 3.5) / 0.35), iy = (int)floor((fy + 3.5) / 0.35);
        const double g = ((ix + iy) & 1) ? 0.8 : 0.7, up[3] = {0, 0, 1}, rgb[3] = {g, g, g};
        best = tf; hit = true; r_shade(rgb, up, d, col);
      }
    }
    if (!hit) {
      const double w = 0.5 * (d[2] + 1.0);
      col[0] = 0.1 + (0.527 - 0.1) * w; col[1] = 0.1 + (0.582 - 0.1) * w; col[2] = 0.35 + (0.906 - 0.35) * w;
    }
    double tb = best, nb[3] = {0, 0, 1};
    int kb = -1;
#pragma unroll 1
    for (int k = 0; k < nob; k++) {
      double n[3];
      if (ob[k].alpha < 1.0 && r_hit(ob[k], cam.o, d, t, n) && t < tb) { tb = t; kb = k; nb[0] = n[0]; nb[1] = n[1]; nb[2] = n[2]; }
    }
    if (kb >= 0) {
      double sc[3];
      r_shade(ob[kb].rgb, nb, d, sc);
      for (int k = 0; k < 3; k++) col[k] = ob[kb].alpha * sc[k] + (1 - ob[kb].alpha) * col[k];
    }
    for (int k = 0; k < 3; k++) {
      const double x = col[k] < 0 ? 0 : (col[k] > 1 ? 1 : col[k]);
      img[px * 3 + k] = (uint8_t)(x * 255.0 + 0.5);
    }
  }
}

}  // namespace sag
