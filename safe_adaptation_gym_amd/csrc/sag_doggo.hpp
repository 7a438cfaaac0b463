// Doggo (assets/xmls/doggo.xml): 3-D articulated quadruped - model tables, state load / store and
// small helpers shared by the physics kernel (sag_doggo_coop.hpp), the post-processing step and the renderer.
//
// Specification: DESIGN.md §4 "Doggo" (the CPU restatement used by the tests states the same model):
// free base + waist hinge + 4 x (hip_z, hip_y, ankle); composite-rigid-body mass matrix and
// recursive Newton-Euler bias in world axes about the base origin; dense Cholesky; joint limits,
// floor and object contacts as MuJoCo-style soft constraints solved by warm-started projected Gauss-Seidel;
// the XML's 14 collision geoms (2 torso cylinders, 12 capsules: round 4; 17 spheres before); horizontal contacts
// against the extruded planar world.
//
// Arithmetic: fp64 for the articulated solve.  The mass matrix spans base mass 4e-2 kg to ankle inertia 1e-5 kg m^2
// and the contact rows need M^-1 J^T: in fp32 the Cholesky solve loses 4-5 of its 7 digits.  State in HBM
// stays fp32 like every other field.  (Round 2 also carried a lane-per-env form of the kernel; round 3 dropped it:
// one implementation, checked against the oracle and on the host sanitizer build.)
#pragma once

namespace sag {

constexpr int DG_NB = 10, DG_NV = 19, DG_NJ = 13, DG_NFP = 16, DG_MAXROWS = 3 * 12 + DG_NJ + 1, DG_NGEOM = 14;
constexpr int DG_PAIR_CAP = 5;              // contacts per (robot geom, world object): oracle DG_PAIR_CAP
constexpr double DG_SITE_R = 0.036;         // touch sites (doggo.xml:8)
constexpr float DG_FACE_TIE = 1e-6f;        // oracle DG_FACE_TIE
// projected Gauss-Seidel sweeps of a forward evaluation that starts from the previous one's forces / of the first
// (cold) one of an env-step: oracle DG_PGS_ITERS, DG_PGS_ITERS0
constexpr int DG_PGS_ITERS = 24, DG_PGS_ITERS0 = 48;
constexpr double DG_GEAR = 0.0125, DG_STIFF = 0.01, DG_Z0 = 0.22, DG_PI = 3.14159265358979323846, DG_GRAV = 9.81;

// model tables, built on the host in fp64 (sag_api.hip: dg_build_model) from the XML numbers
struct DgPhys {                    // what the dynamics read (the cooperative kernel keeps a copy in LDS)
  int parent[DG_NB];
  double bpos[DG_NB][3];
  int dof_body[DG_NV];
  double axis[DG_NJ][3];          // normalised
  double lo[DG_NJ], hi[DG_NJ], springref[DG_NJ];
  int act_joint[12];
  double m[DG_NB], com[DG_NB][3], I[DG_NB][9];  // inertia about the com, body axes
  // the XML's geoms = the collision geometry (and what the renderer draws): body, axis end points in the body frame,
  // radius, capsule (1) / cylinder (0), touch slot of the site at the geom's START (ankle_ka; the one at its end is + 4) or -1
  int geom_body[DG_NGEOM], geom_capsule[DG_NGEOM], geom_touch[DG_NGEOM];
  double geom_a[DG_NGEOM][3], geom_b[DG_NGEOM][3], geom_r[DG_NGEOM];
  // the 16 floor contact points (oracle DG_FLOORPTS, same order), each an end of a geom's axis:
  //   geom | end << 4 | rim << 5 (cylinder: lowest rim point) | merged << 6 (two coincident end spheres: one row, half
  //   the regulariser) | (touch slot + 1) << 7 | half << 11 (the sensor reads half of the merged knee row)
  int fp_code[DG_NFP];
  unsigned anc[DG_NB];            // bit a set: body a is on the path root -> b (inclusive)
};
struct DgModel : DgPhys {
  // rendering: ankle colour class (0 default red, 1 front ankles blue, 2 rear ankles green: doggo.xml:26,40,58,72)
  int geom_ankle[DG_NGEOM];
};
static __constant__ DgModel g_dg;   // uploaded by sag_create

struct DgState {
  double pos[3], quat[4], q[DG_NJ];
  double vlin[3], wloc[3], qd[DG_NJ];
};
struct DgVec { double a[3], l[3]; };
struct DgInertia { double m, mc[3], I[6]; };  // about O, world axes; I: xx xy xz yy yz zz

__device__ inline void dg_cross(const double* a, const double* b, double* o) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ inline double dg_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline void dg_matvec(const double* R, const double* v, double* o) {
  const double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
               z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ inline void dg_matmul(const double* A, const double* B, double* O) {
  double T[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) T[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
#pragma unroll
  for (int k = 0; k < 9; k++) O[k] = T[k];
}
__device__ inline void dg_quat2mat(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
__device__ inline void dg_axis_rot(const double* u, double ang, double* R) {
  const double c = cos(ang), s = sin(ang), t = 1 - c;
  R[0] = c + t * u[0] * u[0]; R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
  R[3] = t * u[0] * u[1] + s * u[2]; R[4] = c + t * u[1] * u[1]; R[5] = t * u[1] * u[2] - s * u[0];
  R[6] = t * u[0] * u[2] - s * u[1]; R[7] = t * u[1] * u[2] + s * u[0]; R[8] = c + t * u[2] * u[2];
}

// record fields -> state.  A zero quaternion means "upright at ROBOT yaw, z = .22, joints 0" (sag.h)
__device__ inline void dg_load(DgState& D, const float* S, size_t N, size_t i) {
  auto F = [&](int k) { return (double)S[saddr(k, N, i)]; };
  D.pos[0] = F(SAG_F_ROBOT); D.pos[1] = F(SAG_F_ROBOT + 1); D.pos[2] = F(SAG_F_ROBOT_EXT);
  for (int k = 0; k < 4; k++) D.quat[k] = F(SAG_F_ROBOT_EXT + 1 + k);
  const double n2 = D.quat[0] * D.quat[0] + D.quat[1] * D.quat[1] + D.quat[2] * D.quat[2] + D.quat[3] * D.quat[3];
  const bool fresh = n2 < 0.25;
  if (fresh) {
    const double yaw = F(SAG_F_ROBOT + 2);
    D.pos[2] = DG_Z0;
    D.quat[0] = cos(0.5 * yaw); D.quat[1] = 0; D.quat[2] = 0; D.quat[3] = sin(0.5 * yaw);
  }
  D.vlin[0] = F(SAG_F_ROBOT + 3); D.vlin[1] = F(SAG_F_ROBOT + 4); D.vlin[2] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 5);
  for (int k = 0; k < 3; k++) D.wloc[k] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 6 + k);
  for (int j = 0; j < DG_NJ; j++) {
    D.q[j] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 9 + j);
    D.qd[j] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 22 + j);
  }
}
__device__ inline void dg_store(const DgState& D, float* S, size_t N, size_t i, float& yaw_out, float& wz_out) {
  double R[9], ww[3];
  dg_quat2mat(D.quat, R);
  dg_matvec(R, D.wloc, ww);
  auto W = [&](int k, double v) { S[saddr(k, N, i)] = (float)v; };
  yaw_out = (float)atan2(R[3], R[0]); wz_out = (float)ww[2];
  W(SAG_F_ROBOT, D.pos[0]); W(SAG_F_ROBOT + 1, D.pos[1]); W(SAG_F_ROBOT + 2, yaw_out);
  W(SAG_F_ROBOT + 3, D.vlin[0]); W(SAG_F_ROBOT + 4, D.vlin[1]); W(SAG_F_ROBOT + 5, wz_out);
  W(SAG_F_ROBOT_EXT, D.pos[2]);
  for (int k = 0; k < 4; k++) W(SAG_F_ROBOT_EXT + 1 + k, D.quat[k]);
  W(SAG_F_ROBOT_EXT + 5, D.vlin[2]);
  for (int k = 0; k < 3; k++) W(SAG_F_ROBOT_EXT + 6 + k, D.wloc[k]);
  for (int j = 0; j < DG_NJ; j++) { W(SAG_F_ROBOT_EXT + 9 + j, D.q[j]); W(SAG_F_ROBOT_EXT + 22 + j, D.qd[j]); }
}

__device__ inline void dg_qd(const DgState& D, double* qd) {
  for (int k = 0; k < 3; k++) { qd[k] = D.vlin[k]; qd[3 + k] = D.wloc[k]; }
  for (int j = 0; j < DG_NJ; j++) qd[6 + j] = D.qd[j];
}


__device__ inline void dg_inertia_apply(const DgInertia& I, const DgVec& v, DgVec& f) {
  double t[3];
  dg_cross(v.a, I.mc, t);
  for (int k = 0; k < 3; k++) f.l[k] = I.m * v.l[k] + t[k];
  const double n0 = I.I[0] * v.a[0] + I.I[1] * v.a[1] + I.I[2] * v.a[2],
               n1 = I.I[1] * v.a[0] + I.I[3] * v.a[1] + I.I[4] * v.a[2],
               n2 = I.I[2] * v.a[0] + I.I[4] * v.a[1] + I.I[5] * v.a[2];
  dg_cross(I.mc, v.l, t);
  f.a[0] = n0 + t[0]; f.a[1] = n1 + t[1]; f.a[2] = n2 + t[2];
}
__device__ inline double dg_sdot(const DgVec& s, const DgVec& f) { return dg_dot(s.a, f.a) + dg_dot(s.l, f.l); }


__device__ inline void dg_mcross(const DgVec& v, const DgVec& s, DgVec& o) {
  double t1[3], t2[3];
  dg_cross(v.a, s.a, o.a);
  dg_cross(v.a, s.l, t1);
  dg_cross(v.l, s.a, t2);
  for (int k = 0; k < 3; k++) o.l[k] = t1[k] + t2[k];
}




// body frames of the kinematic tree (world rotation R[b], origin p[b]): the renderer's forward kinematics
__device__ inline void dg_frames(const DgState& D, double (*R)[9], double (*p)[3]) {
  const DgModel& M = g_dg;
  dg_quat2mat(D.quat, R[0]);
  for (int k = 0; k < 3; k++) p[0][k] = D.pos[k];
  int dof = 6;
#pragma unroll 1
  for (int b = 1; b < DG_NB; b++) {
    const int P = M.parent[b];
    double off[3];
    dg_matvec(R[P], M.bpos[b], off);
    for (int k = 0; k < 3; k++) p[b][k] = p[P][k] + off[k];
    for (int k = 0; k < 9; k++) R[b][k] = R[P][k];
    while (dof < DG_NV && M.dof_body[dof] == b) {
      double Rj[9];
      dg_axis_rot(M.axis[dof - 6], D.q[dof - 6], Rj);
      dg_matmul(R[b], Rj, R[b]);
      dof++;
    }
  }
}

__device__ inline double dg_impedance(double depth) {
  double x = depth / (double)SOL_WIDTH; if (x > 1) x = 1;
  const double y = x < 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return (double)SOL_D0 + ((double)SOL_D1 - (double)SOL_D0) * y;
}

// planar free bodies as the Doggo code sees them: [x y yaw vx vy w ax ay aw] x NBODY (fp32, the
// arithmetic of the planar world) + inverse inertia
struct DgWorld {
  float fb[NBODY][9];
  float minv[NBODY][6];
  uint32_t valid;   // free bodies that exist
};

  // overflow: a row did not fit (reported as record flag bit 2)




// circle (sphere footprint) vs the geoms of one planar body: contacts with normal from the circle
// (A) to the body (B); conventions of the planar narrowphase (cc_contact / cb_contact above)
struct DgHit { float nx, ny, px, py, depth; };
__device__ inline int dg_circle_geom(float ax, float ay, float ra, const Geom& g, float bx, float by, float cb,
                                     float sb, DgHit& h) {
  if (!g.box) {
    const float dx = bx - ax, dy = by - ay, d2 = dx * dx + dy * dy, rs = ra + g.a;
    if (d2 >= rs * rs) return 0;
    const float d = sqrtf(d2);
    float nx = 1, ny = 0;
    if (d > 1e-12f) { nx = dx / d; ny = dy / d; }
    h.nx = nx; h.ny = ny; h.depth = rs - d;
    h.px = ax + nx * (ra - 0.5f * h.depth); h.py = ay + ny * (ra - 0.5f * h.depth);
    return 1;
  }
  const float wx = ax - bx, wy = ay - by;
  const float lx = cb * wx + sb * wy, ly = -sb * wx + cb * wy;
  float qx = clampf(lx, -g.a, g.a), qy = clampf(ly, -g.b, g.b);
  const float ddx = lx - qx, ddy = ly - qy;
  float onx, ony, depth;
  if (ddx == 0 && ddy == 0) {
    const float px = g.a - fabsf(lx), py = g.b - fabsf(ly);
    // (the x face also when the y face is nearer by less than DG_FACE_TIE: the deepest point of a capsule's axis inside a
    //  box is where the two are EQUAL - dg_seg_box_t's diagonal kinks -, and rounding must not pick the face there)
    if (px <= py + DG_FACE_TIE) { onx = lx >= -DG_FACE_TIE ? 1.f : -1.f; ony = 0; depth = ra + px; qx = onx * g.a; }
    else { onx = 0; ony = ly >= -DG_FACE_TIE ? 1.f : -1.f; depth = ra + py; qy = ony * g.b; }
  } else {
    const float d2 = ddx * ddx + ddy * ddy;
    if (d2 >= ra * ra) return 0;
    const float d = sqrtf(d2);
    onx = ddx / d; ony = ddy / d; depth = ra - d;
  }
  h.nx = -(cb * onx - sb * ony); h.ny = -(sb * onx + cb * ony);
  h.px = bx + cb * qx - sb * qy; h.py = by + sb * qx + cb * qy;
  h.depth = depth;
  return 1;
}

// ---- capsules and cylinders against the extruded planar world (oracle: dg_box_sd, dg_seg_box_t, dg_collide_body) ----
// signed distance of the point (x, y) in a box's frame to the box |x| <= hx, |y| <= hy (negative inside)
__device__ inline float dg_box_sd(float x, float y, float hx, float hy) {
  const float qx = fabsf(x) - hx, qy = fabsf(y) - hy;
  const float ox = qx > 0 ? qx : 0.f, oy = qy > 0 ? qy : 0.f;
  float in = qx > qy ? qx : qy;
  if (in > 0) in = 0.f;
  return sqrtf(ox * ox + oy * oy) + in;
}
// The parameter t in [t0, t1] of the segment a + t d (box frame) with the smallest signed distance to the box: the
// distance to a convex set is convex along a line, so the minimiser is an end of the range, the foot of a corner on the
// line, or a kink (the box's axes and the diagonals |x| - hx = |y| - hy inside it).  Twelve candidates in the oracle's
// order, the first of the smallest wins.
__device__ inline float dg_seg_box_t(float ax, float ay, float dx, float dy, float hx, float hy, float t0, float t1) {
  const float dd = dx * dx + dy * dy, e = hx - hy;
  float best_t = t0, best = dg_box_sd(ax + t0 * dx, ay + t0 * dy, hx, hy);
  auto tryt = [&](float t) {
    if (!(t > t0)) t = t0;
    if (t > t1) t = t1;
    const float sd = dg_box_sd(ax + t * dx, ay + t * dy, hx, hy);
    if (sd < best) { best = sd; best_t = t; }
  };
  tryt(t1);
  if (dd > 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float cx = (k == 0 || k == 3) ? hx : -hx, cy = k < 2 ? hy : -hy;
      tryt(((cx - ax) * dx + (cy - ay) * dy) / dd);
    }
    if (dx != 0) tryt(-ax / dx);
    if (dy != 0) tryt(-ay / dy);
    if (dx - dy != 0) { tryt((e - (ax - ay)) / (dx - dy)); tryt((-e - (ax - ay)) / (dx - dy)); }
    if (dx + dy != 0) { tryt((e - (ax + ay)) / (dx + dy)); tryt((-e - (ax + ay)) / (dx + dy)); }
  }
  return best_t;
}
// vertices of box P strictly inside box Q -> hits with the outward normal of Q's least-penetrated face times `sign`
// (oracle verts_in_box; vertex order (+,+), (-,+), (-,-), (+,-)); appends to out[n..cap)
__device__ inline void dg_verts_in_box(float pxc, float pyc, float cp, float sp, float phx, float phy, float qxc, float qyc,
                                       float cq, float sq, float qhx, float qhy, float sign, DgHit* out, int& n, int cap) {
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    const float sx = (k == 0 || k == 3) ? phx : -phx, sy = (k < 2) ? phy : -phy;
    const float vx = pxc + cp * sx - sp * sy, vy = pyc + sp * sx + cp * sy;
    const float wx = vx - qxc, wy = vy - qyc;
    const float lx = cq * wx + sq * wy, ly = -sq * wx + cq * wy;
    const float dx = qhx - fabsf(lx), dy = qhy - fabsf(ly);
    if (dx <= 0 || dy <= 0) continue;
    float onx, ony, depth;
    if (dx < dy) { onx = lx >= 0 ? 1.f : -1.f; ony = 0; depth = dx; }
    else { onx = 0; ony = ly >= 0 ? 1.f : -1.f; depth = dy; }
    if (n < cap) {
      out[n].nx = sign * (cq * onx - sq * ony); out[n].ny = sign * (sq * onx + cq * ony);
      out[n].px = vx; out[n].py = vy; out[n].depth = depth;
    }
    n++;
  }
}
// oriented rectangle (A: centre, cos / sin, half extents) vs one footprint geom (B): hits with the normal from A to B, in
// the oracle's geom_pair order; returns how many the pair has (only the first cap - n0 are stored)
__device__ inline int dg_rect_geom(float ax, float ay, float ca, float sa, float ahx, float ahy, const Geom& g, float bx, float by,
                                   float cb, float sb, DgHit* out, int n0, int cap) {
  int n = n0;
  if (!g.box) {   // circle_box(circle = B, box = A), normal flipped
    Geom r; r.box = 1; r.ox = 0; r.oy = 0; r.a = ahx; r.b = ahy; r.r = 0;
    DgHit h;
    if (dg_circle_geom(bx, by, g.a, r, ax, ay, ca, sa, h)) {
      if (n < cap) { h.nx = -h.nx; h.ny = -h.ny; out[n] = h; }
      n++;
    }
    return n - n0;
  }
  dg_verts_in_box(ax, ay, ca, sa, ahx, ahy, bx, by, cb, sb, g.a, g.b, -1.f, out, n, cap);
  dg_verts_in_box(bx, by, cb, sb, g.a, g.b, ax, ay, ca, sa, ahx, ahy, 1.f, out, n, cap);
  return n - n0;
}

}  // namespace sag
