// rgb_observation (safe_adaptation_gym.py:122-126): the 64 x 64 x 3 uint8 image of
// the robot's `vision` camera, ray-cast on the device.  The reference renders with MuJoCo's OpenGL
// rasteriser ("parity unpinned"); the specification of this image - camera from the robot XML
// (point.xml:14, car.xml:14, doggo.xml:13), scene of mujoco_bridge.py:88-123, geoms and colours of
// consts.py / primitive_objects.py / the task files and the robot XMLs (the robot's own geoms included),
// translucent layers composited back to front, Lambert + headlight shade - is written out in DESIGN.md
// "rgb_observation".  The same ray caster serves the human view (render.py,
// safe_adaptation_gym.py:109-111,239-257): fixed / tracking cameras, any image size, lidar rings and cost indicator.  fp64: a pixel is a hard
// decision (which surface, which checker square, rounding to 8 bits), like a lidar bin.
// One workgroup per env: lane 0 builds the scene (<= 112 geoms) in LDS, 256 threads share the pixels.
// Round 4: every geom also gets a bounding sphere relative to the camera (fp32, a margin of 2 % + 2 cm over the geom's
// own circumscribed radius - far more than fp32 rounding of the test can amount to inside the 7 x 7 m scene), and a ray
// only runs the fp64 intersection of the geoms whose sphere it passes: a cull, conservative by construction, so the
// image is the same to the last bit.
#pragma once

namespace sag {

// Cameras (mujoco_bridge.py:126-153 + the robot XMLs): the robot's own `vision` camera, the two fixed world cameras
// and the tracking camera of the human view (render.py, safe_adaptation_gym.py:109-111 render_options['camera_id']).
enum { SAG_CAM_VISION = 0, SAG_CAM_FIXEDNEAR = 1, SAG_CAM_FIXEDFAR = 2, SAG_CAM_TRACK = 3 };
enum { SAG_RENDER_OVERLAYS = 1 };   // lidar rings + cost indicator (render.py)

struct RCam { double o[3], X[3], Y[3], Z[3], tanh_; };
// kind 0 box (yaw-rotated; half extents a, b, h), 1 vertical cylinder (a = r, h = half height), 2 sphere (a = r),
// 3 rod: cylinder (b = 0) or capsule (b = 1) of radius a from c to e (any orientation: wheels, the Doggo's limbs)
struct RObj { int kind; double c[3], a, b, h, cs, sn, rgb[3], alpha, e[3]; };
constexpr int R_MAXOBJ = 112, R_W = 64, R_H = 64, R_MAXLAYERS = 12;

__device__ inline void r_norm(double* v) {
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n; v[1] /= n; v[2] /= n;
}

// R: body -> world (row-major), p: body origin
__device__ inline void r_camera(int robot, int cam_id, const double* R, const double* p, RCam& cam) {
  if (cam_id != SAG_CAM_VISION) {
    // <camera name="fixednear" pos="0 -2 2" zaxis="0 -1 1"/>, "fixedfar" pos="0 -5 5"; the tracking camera has the
    // same world orientation and keeps the offset (0, -2, 2) from the robot body (mode="track": position only).
    // zaxis = the direction the camera looks AWAY from; x = world x; default fovy 45
    const double off = cam_id == SAG_CAM_FIXEDFAR ? 5.0 : 2.0, q = 0.70710678118654752440;
    cam.o[0] = cam_id == SAG_CAM_TRACK ? p[0] : 0.0;
    cam.o[1] = (cam_id == SAG_CAM_TRACK ? p[1] : 0.0) - off;
    cam.o[2] = (cam_id == SAG_CAM_TRACK ? p[2] : 0.0) + off;
    cam.X[0] = 1; cam.X[1] = 0; cam.X[2] = 0;
    cam.Y[0] = 0; cam.Y[1] = q; cam.Y[2] = q;
    cam.Z[0] = 0; cam.Z[1] = -q; cam.Z[2] = q;
    cam.tanh_ = tan(0.5 * 45.0 * PI_D / 180);
    return;
  }
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

// entry of the ray o + t d into the sphere (centre c, radius r), t > 1e-6
__device__ inline bool r_sphere(const double* c, double r, const double* o, const double* d, double& tout, double* n) {
  const double oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
  const double b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
  const double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - r * r;
  const double disc = b * b - cc;
  if (disc <= 0) return false;
  const double t = -b - sqrt(disc);
  if (t <= 1e-6) return false;
  tout = t;
  for (int k = 0; k < 3; k++) n[k] = (oc[k] + t * d[k]) / r;
  return true;
}

// nearest intersection t > 1e-6 of the ray o + t d with the geom; n = surface normal there
__device__ inline bool r_hit(const RObj& ob, const double* o, const double* d, double& tout, double* n) {
  if (ob.kind == 2) return r_sphere(ob.c, ob.a, o, d, tout, n);
  if (ob.kind == 3) {
    // axis u from c to e, length L; side: |(oc + t d) - ((oc + t d).u) u| = r with 0 <= s <= L
    double u[3] = {ob.e[0] - ob.c[0], ob.e[1] - ob.c[1], ob.e[2] - ob.c[2]};
    const double L = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int k = 0; k < 3; k++) u[k] /= L;
    const double oc[3] = {o[0] - ob.c[0], o[1] - ob.c[1], o[2] - ob.c[2]};
    const double du = d[0] * u[0] + d[1] * u[1] + d[2] * u[2], ou = oc[0] * u[0] + oc[1] * u[1] + oc[2] * u[2];
    const double dd[3] = {d[0] - du * u[0], d[1] - du * u[1], d[2] - du * u[2]};
    const double oo[3] = {oc[0] - ou * u[0], oc[1] - ou * u[1], oc[2] - ou * u[2]};
    const double A = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2], B = oo[0] * dd[0] + oo[1] * dd[1] + oo[2] * dd[2];
    const double C = oo[0] * oo[0] + oo[1] * oo[1] + oo[2] * oo[2] - ob.a * ob.a;
    double best = 1e30;
    bool found = false;
    if (A > 0) {
      const double disc = B * B - A * C;
      if (disc > 0) {
        const double t = (-B - sqrt(disc)) / A, sax = ou + t * du;
        if (t > 1e-6 && sax >= 0 && sax <= L) {
          best = t; found = true;
          for (int k = 0; k < 3; k++) n[k] = (oo[k] + t * dd[k]) / ob.a;
        }
      }
    }
    if (ob.b != 0) {   // capsule: the two end spheres
      double t, m[3];
      if (r_sphere(ob.c, ob.a, o, d, t, m) && t < best) { best = t; found = true; n[0] = m[0]; n[1] = m[1]; n[2] = m[2]; }
      if (r_sphere(ob.e, ob.a, o, d, t, m) && t < best) { best = t; found = true; n[0] = m[0]; n[1] = m[1]; n[2] = m[2]; }
    } else if (du != 0) {   // cylinder: the flat cap facing the ray
      const double s = du < 0 ? 1.0 : 0.0, t = (s * L - ou) / du;
      const double q[3] = {oo[0] + t * dd[0], oo[1] + t * dd[1], oo[2] + t * dd[2]};
      if (t > 1e-6 && t < best && q[0] * q[0] + q[1] * q[1] + q[2] * q[2] <= ob.a * ob.a) {
        best = t; found = true;
        const double sg = du < 0 ? 1.0 : -1.0;
        n[0] = sg * u[0]; n[1] = sg * u[1]; n[2] = sg * u[2];
      }
    }
    if (!found) return false;
    tout = best;
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
  if (n >= R_MAXOBJ) return;
  RObj& q = ob[n++];
  q.kind = kind; q.c[0] = x; q.c[1] = y; q.c[2] = z; q.a = a; q.b = b; q.h = h;
  q.cs = cos(yaw); q.sn = sin(yaw); q.rgb[0] = cr; q.rgb[1] = cg; q.rgb[2] = cb; q.alpha = al;
  q.e[0] = q.e[1] = q.e[2] = 0;
}
// rod from body-frame points a, b of the body with frame (R, p)
__device__ inline void r_rod(RObj* ob, int& n, const double* R, const double* p, double ax, double ay, double az, double bx,
                             double by, double bz, double r, bool capsule, double cr, double cg, double cb, double al) {
  if (n >= R_MAXOBJ) return;
  RObj& q = ob[n++];
  q.kind = 3; q.a = r; q.b = capsule ? 1 : 0; q.h = 0; q.cs = 1; q.sn = 0;
  for (int k = 0; k < 3; k++) {
    q.c[k] = p[k] + R[3 * k] * ax + R[3 * k + 1] * ay + R[3 * k + 2] * az;
    q.e[k] = p[k] + R[3 * k] * bx + R[3 * k + 1] * by + R[3 * k + 2] * bz;
  }
  q.rgb[0] = cr; q.rgb[1] = cg; q.rgb[2] = cb; q.alpha = al;
}

// out: [N][H][W][3] uint8.  obs / cost (device pointers or nullptr): last observation [N][obs_dim] and cost flags for
// the overlays (lidar rings, cost indicator).
__global__ __launch_bounds__(256) void k_render_rgb(const float* __restrict__ S, const int32_t* __restrict__ I, int N,
                                                    int robot, int cam_id, int W, int H, int flags,
                                                    const float* __restrict__ obs, int obs_dim,
                                                    const uint8_t* __restrict__ cost, uint8_t* __restrict__ out) {
  __shared__ RObj ob[R_MAXOBJ];
  __shared__ float4 bsph[R_MAXOBJ];   // bounding sphere: centre - camera origin, radius with margin
  __shared__ RCam cam;
  __shared__ int nob_s;
  const size_t i = blockIdx.x;
  if (threadIdx.x == 0) {
    auto F = [&](int k) { return (double)S[saddr(k, (size_t)N, i)]; };
    const uint32_t meta = (uint32_t)I[iaddr(DI_META, (size_t)N, i)], tstate = (uint32_t)I[iaddr(DI_TSTATE, (size_t)N, i)];
    const int task = meta & 15, nH = meta >> 4 & 15, nV = meta >> 8 & 15, nP = meta >> 12 & 3, nB = meta >> 14 & 7,
              box_kind = meta >> 17 & 3;
    double R[9], p[3];
    double yaw = F(SAG_F_ROBOT + 2);
    if (robot == SAG_ROBOT_DOGGO) {
      DgState D;
      dg_load(D, S, (size_t)N, i);
      dg_quat2mat(D.quat, R);
      p[0] = D.pos[0]; p[1] = D.pos[1]; p[2] = D.pos[2];
    } else {
      const double c = cos(yaw), s = sin(yaw);
      R[0] = c; R[1] = -s; R[2] = 0; R[3] = s; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
      p[0] = F(SAG_F_ROBOT); p[1] = F(SAG_F_ROBOT + 1); p[2] = PT_Z;
    }
    r_camera(robot, cam_id, R, p, cam);
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
    if (box_kind == SAG_BOX_BOX) {
      // push_box.py:28-72: the box (half .2) and its four corner columns (half .1 x .1 x .2 at (+-.2, +-.2)), one rgba
      const double bx = F(SAG_F_BOX), by = F(SAG_F_BOX + 1), byaw = F(SAG_F_BOX + 2), c = cos(byaw), s = sin(byaw);
      r_obj(ob, n, 0, bx, by, 0.2, 0.2, 0.2, 0.2, byaw, 1, 1, 0, 0.25);
      for (int q = 0; q < 4; q++) {
        const double lx = (q & 1) ? -0.2 : 0.2, ly = (q & 2) ? -0.2 : 0.2;
        r_obj(ob, n, 0, bx + c * lx - s * ly, by + s * lx + c * ly, 0.2, 0.1, 0.1, 0.2, byaw, 1, 1, 0, 0.25);
      }
    }
    else if (box_kind == SAG_BOX_ROD) r_obj(ob, n, 0, F(SAG_F_BOX), F(SAG_F_BOX + 1), 0.08, 0.08, 0.3, 0.08, F(SAG_F_BOX + 2), 1, 1, 1, 1.0);
    else if (box_kind == SAG_BOX_BALL) r_obj(ob, n, 2, F(SAG_F_BOX), F(SAG_F_BOX + 1), 0.14, 0.14, 0, 0, 0.0, 1, 1, 1, 1.0);
    // ---- the robot's own geoms (default rgba 1 0 0 1 in the three XMLs) ----------------------------------
    if (robot == SAG_ROBOT_POINT) {          // point.xml:18-19: sphere r .1, arrow box half .05 at (.1, 0, 0)
      r_obj(ob, n, 2, p[0], p[1], p[2], 0.1, 0, 0, 0.0, 1, 0, 0, 1.0);
      r_obj(ob, n, 0, p[0] + 0.1 * R[0], p[1] + 0.1 * R[3], p[2], 0.05, 0.05, 0.05, yaw, 1, 0, 0, 1.0);
    } else if (robot == SAG_ROBOT_CAR) {     // car.xml:16-32
      const double BX[5][6] = {{0, 0, 0, .1, .1, .05}, {0, .15, 0, .1, .01, .05}, {0, .125, 0, .01, .025, .03},
                               {0, -.165, 0, .05, .01, .05}, {0, -.13, .04, .05, .03, .01}};
      for (int g = 0; g < 5; g++)
        r_obj(ob, n, 0, p[0] + R[0] * BX[g][0] + R[1] * BX[g][1], p[1] + R[3] * BX[g][0] + R[4] * BX[g][1], p[2] + BX[g][2],
              BX[g][3], BX[g][4], BX[g][5], yaw, 1, 0, 0, 1.0);
      r_rod(ob, n, R, p, -.155, .1, -.05, -.105, .1, -.05, 0.05, false, 1, 0, 0, 1.0);   // left wheel (cylinder along x)
      r_rod(ob, n, R, p, .105, .1, -.05, .155, .1, -.05, 0.05, false, 1, 0, 0, 1.0);     // right wheel
      r_obj(ob, n, 2, p[0] + R[1] * -.1, p[1] + R[4] * -.1, p[2] - .05, 0.05, 0, 0, 0.0, 1, 0, 0, 1.0);   // rear ball
    } else {                                 // doggo.xml: two torso cylinders, twelve capsules (ankles blue / green)
      struct { double R[DG_NB][9], p[DG_NB][3]; } K;
      DgState D;
      dg_load(D, S, (size_t)N, i);
      dg_frames(D, K.R, K.p);
      for (int g = 0; g < DG_NGEOM; g++) {
        const int b = g_dg.geom_body[g];
        const bool ankle = g_dg.geom_ankle[g] != 0, front = g_dg.geom_ankle[g] == 1;
        r_rod(ob, n, K.R[b], K.p[b], g_dg.geom_a[g][0], g_dg.geom_a[g][1], g_dg.geom_a[g][2], g_dg.geom_b[g][0],
              g_dg.geom_b[g][1], g_dg.geom_b[g][2], g_dg.geom_r[g], g_dg.geom_capsule[g] != 0,
              ankle ? 0.0 : 1.0, ankle && !front ? 1.0 : 0.0, ankle && front ? 1.0 : 0.0, 1.0);
      }
    }
    // ---- overlays of the human view (render.py): three lidar rings above the robot, the cost indicator --------
    if (flags & SAG_RENDER_OVERLAYS) {
      const float* o = obs ? obs + i * (size_t)obs_dim : nullptr;
      for (int ring = 0; ring < 3; ring++) {          // obstacles (red, z .5), goal (green, .56), objects (blue, .62)
        const int col0 = ring == 0 ? 0 : (ring == 1 ? 32 : 16);
        for (int j = 0; j < SAG_LIDAR_BINS; j++) {
          const double th = 2.0 * PI_D * (j + 0.5) / SAG_LIDAR_BINS, lx = 0.15 * cos(th), ly = 0.15 * sin(th), lz = 0.5 + 0.06 * ring;
          double al = (o ? (double)o[col0 + j] : 0.0) + 0.1;
          if (al > 1) al = 1;
          r_obj(ob, n, 2, p[0] + R[0] * lx + R[1] * ly + R[2] * lz, p[1] + R[3] * lx + R[4] * ly + R[5] * lz,
                p[2] + R[6] * lx + R[7] * ly + R[8] * lz, 0.025, 0, 0, 0.0, ring == 0 ? al : 0, ring == 1 ? al : 0, ring == 2 ? al : 0, al);
        }
      }
      if (cost && cost[i]) r_obj(ob, n, 2, p[0], p[1], p[2], 0.25, 0, 0, 0.0, 1, 0, 0, 0.5);
    }
    nob_s = n;
  }
  __syncthreads();
  const int nob = nob_s;
  for (int k = threadIdx.x; k < nob; k += 256) {
    const RObj& q = ob[k];
    double c[3] = {q.c[0], q.c[1], q.c[2]}, rb;
    if (q.kind == 0) rb = sqrt(q.a * q.a + q.b * q.b + q.h * q.h);
    else if (q.kind == 1) rb = sqrt(q.a * q.a + q.h * q.h);
    else if (q.kind == 2) rb = q.a;
    else {
      const double e[3] = {q.e[0] - q.c[0], q.e[1] - q.c[1], q.e[2] - q.c[2]};
      rb = 0.5 * sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + q.a;
      for (int a = 0; a < 3; a++) c[a] += 0.5 * e[a];
    }
    bsph[k] = make_float4((float)(c[0] - cam.o[0]), (float)(c[1] - cam.o[1]), (float)(c[2] - cam.o[2]), (float)(1.02 * rb + 0.02));
  }
  __syncthreads();
  // may the ray (unit direction f) hit geom k at all?  (distance of the sphere's centre from the ray's line, and not wholly behind)
  auto may_hit = [&](int k, float fx, float fy, float fz) {
    const float4 b = bsph[k];
    const float proj = b.x * fx + b.y * fy + b.z * fz;
    const float perp2 = (b.x * b.x + b.y * b.y + b.z * b.z) - proj * proj;
    return perp2 <= b.w * b.w && proj >= -b.w;
  };
  uint8_t* img = out + i * (size_t)W * H * 3;
  const double aspect = (double)W / (double)H;
  // Pixels in 8 x 8 tiles, a tile per wavefront pass: the 64 rays of a wavefront then span an eighth of the image's width instead of a
  // whole row, and the wavefront runs the fp64 intersection of an object only when one of ITS rays passes the sphere test - with rows,
  // some lane's ray reached nearly every object at that elevation.  Same rays, same arithmetic per pixel.
  const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
#pragma unroll 1
  for (int q = threadIdx.x; q < tiles_x * tiles_y * 64; q += 256) {
    const int tile = q >> 6, w_ = q & 63, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int r = ty * 8 + (w_ >> 3), c = tx * 8 + (w_ & 7);
    if (r >= H || c >= W) continue;
    const int px = r * W + c;
    const double u = ((c + 0.5) / (0.5 * W) - 1.0) * cam.tanh_ * aspect, v = (1.0 - (r + 0.5) / (0.5 * H)) * cam.tanh_;
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = u * cam.X[k] + v * cam.Y[k] - cam.Z[k];
    r_norm(d);
    const float fx = (float)d[0], fy = (float)d[1], fz = (float)d[2];
    double best = 1e30, col[3] = {0, 0, 0}, t;
    bool hit = false;
    // ONE pass over the geoms (sphere test and intersection once each): an opaque hit moves the surface in; a translucent hit goes into
    // the list of the nearest R_MAXLAYERS translucent hits, sorted by distance (equal distances keep geom order).  The layers that count
    // are those in front of the FINAL opaque surface - a prefix of the sorted list, cut once the surface is known (below): the same set
    // the former second pass collected with `t < best`.
    double lt[R_MAXLAYERS];
    int lk[R_MAXLAYERS], nl = 0;
#pragma unroll 1
    for (int k = 0; k < nob; k++) {
      double n[3];
      if (!may_hit(k, fx, fy, fz) || !r_hit(ob[k], cam.o, d, t, n)) continue;
      if (ob[k].alpha >= 1.0) {
        if (t < best) { best = t; hit = true; r_shade(ob[k].rgb, n, d, col); }
      } else {
        int pos = nl < R_MAXLAYERS ? nl++ : (t < lt[R_MAXLAYERS - 1] ? R_MAXLAYERS - 1 : -1);
        if (pos < 0) continue;
        while (pos > 0 && lt[pos - 1] > t) { lt[pos] = lt[pos - 1]; lk[pos] = lk[pos - 1]; pos--; }
        lt[pos] = t; lk[pos] = k;
      }
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
    // translucent geoms in front of the opaque surface, composited back to front
    while (nl > 0 && !(lt[nl - 1] < best)) nl--;
    for (int q = nl - 1; q >= 0; q--) {
      double n[3], sc[3];
      r_hit(ob[lk[q]], cam.o, d, t, n);
      r_shade(ob[lk[q]].rgb, n, d, sc);
      for (int k = 0; k < 3; k++) col[k] = ob[lk[q]].alpha * sc[k] + (1 - ob[lk[q]].alpha) * col[k];
    }
    for (int k = 0; k < 3; k++) {
      const double x = col[k] < 0 ? 0 : (col[k] > 1 ? 1 : col[k]);
      img[px * 3 + k] = (uint8_t)(x * 255.0 + 0.5);
    }
  }
}

}  // namespace sag
