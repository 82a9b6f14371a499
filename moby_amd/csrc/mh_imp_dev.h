// Device helpers shared by the island pipeline (mh_impact.hip) and the large-world stepper (mh_big.hip): 3-vectors,
// the velocity of a body point, the tangent basis and the X block of a free body, in the oracle's operation order
// (oracle/world.hpp: point_vel, orthonormal_basis, inertia_world, inv_inertia).
#pragma once
#include <hip/hip_runtime.h>
#ifndef MH_DEV
#define MH_DEV __device__ __forceinline__
#endif

namespace mh { namespace imp {

struct P3 { double x, y, z; };
MH_DEV P3 p3(double x, double y, double z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
MH_DEV P3 operator+(P3 a, P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
MH_DEV P3 operator-(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
MH_DEV P3 operator-(P3 a) { return p3(-a.x, -a.y, -a.z); }
MH_DEV P3 operator*(P3 a, double s) { return p3(a.x * s, a.y * s, a.z * s); }
MH_DEV P3 operator/(P3 a, double s) { return p3(a.x / s, a.y / s, a.z / s); }
MH_DEV double dot3(P3 a, P3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MH_DEV P3 cross3(P3 a, P3 b) { return p3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

MH_DEV P3 ld3(const double* p) { return p3(p[0], p[1], p[2]); }
// velocity of the body point at world point p (oracle World::point_vel)
MH_DEV P3 point_vel(const double* st, int b, P3 p) {
  if (b < 0) return p3(0.0, 0.0, 0.0);
  const double* s = st + 13 * b;
  return ld3(s + 7) + cross3(ld3(s + 10), p - ld3(s));
}
// Vector3d::determine_orthonormal_basis as pinned by the oracle (World::orthonormal_basis)
MH_DEV void basis(P3 n, P3& s, P3& t) {
  const double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
  P3 e;
  if (ax <= ay && ax <= az) e = p3(1, 0, 0); else if (ay <= az) e = p3(0, 1, 0); else e = p3(0, 0, 1);
  s = cross3(n, e); s = s / sqrt(dot3(s, s));
  t = cross3(n, s);
}
// X block of a body: 1/m and the inverse world inertia, inverse_SPD order (oracle World::inv_inertia, linalg.hpp)
MH_DEV void inv_inertia(const double* st, const double* J, double m, double* out /*10*/, double* Jw_out = nullptr /*9*/) {
  const double x = st[3], y = st[4], z = st[5], w = st[6];
  double R[9];
  R[0] = 1.0 - 2.0 * (y*y + z*z); R[1] = 2.0 * (x*y - z*w);       R[2] = 2.0 * (x*z + y*w);
  R[3] = 2.0 * (x*y + z*w);       R[4] = 1.0 - 2.0 * (x*x + z*z); R[5] = 2.0 * (y*z - x*w);
  R[6] = 2.0 * (x*z - y*w);       R[7] = 2.0 * (y*z + x*w);       R[8] = 1.0 - 2.0 * (x*x + y*y);
  double Tm[9], Jw[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tm[3*i+j] = R[3*i+j] * J[j];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    Jw[3*i+j] = (Tm[3*i] * R[3*j] + Tm[3*i+1] * R[3*j+1]) + Tm[3*i+2] * R[3*j+2];
  Jw[1] = Jw[3]; Jw[2] = Jw[6]; Jw[5] = Jw[7];
  if (Jw_out) for (int i = 0; i < 9; i++) Jw_out[i] = Jw[i];
  { const double l = sqrt(m); const double e = 1.0 / l; out[0] = e / l; }
  double l00, l10, l20, l11, l21, l22;
  { double ajj = Jw[0]; ajj = sqrt(ajj); l00 = ajj; l10 = Jw[3] / ajj; l20 = Jw[6] / ajj; }
  { double ajj = Jw[4]; ajj = ajj - l10 * l10; ajj = sqrt(ajj); l11 = ajj; double s = Jw[7]; s = s - l20 * l10; l21 = s / ajj; }
  { double ajj = Jw[8]; ajj = ajj - l20 * l20; ajj = ajj - l21 * l21; ajj = sqrt(ajj); l22 = ajj; }
  double Ai[9];
  for (int c = 0; c < 3; c++) {
    double b0 = (c == 0) ? 1.0 : 0.0, b1 = (c == 1) ? 1.0 : 0.0, b2 = (c == 2) ? 1.0 : 0.0;
    b0 = b0 / l00; b1 = b1 - b0 * l10; b2 = b2 - b0 * l20;
    b1 = b1 / l11; b2 = b2 - b1 * l21;
    b2 = b2 / l22;
    { double s = b2; b2 = s / l22; }
    { double s = b1; s = s - l21 * b2; b1 = s / l11; }
    { double s = b0; s = s - l10 * b1; s = s - l20 * b2; b0 = s / l00; }
    Ai[0 + 3*c] = b0; Ai[1 + 3*c] = b1; Ai[2 + 3*c] = b2;
  }
  Ai[0 + 3*1] = Ai[1 + 3*0]; Ai[0 + 3*2] = Ai[2 + 3*0]; Ai[1 + 3*2] = Ai[2 + 3*1];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[1 + 3*r + c] = Ai[r + 3*c];
}

// ---- implicit joints (include/moby_hip_stack.h; oracle World::body_vec / joint_rows / joint_eval / joint_jac) -------------
struct JointTab { int nj; const int* type; const int* in; const int* out; const double* anchor_in; const double* anchor_out;
                  const double* vec_in; const double* vec_out; };
MH_DEV P3 jt_body_vec(const double* st, int nb, int b, const double* u) {      // R u for a dynamic body, u for the static world
  if (b < 0 || b >= nb) return p3(u[0], u[1], u[2]);
  const double x = st[13*b+3], y = st[13*b+4], z = st[13*b+5], w = st[13*b+6];
  double R[9];
  R[0] = 1.0 - 2.0 * (y*y + z*z); R[1] = 2.0 * (x*y - z*w);       R[2] = 2.0 * (x*z + y*w);
  R[3] = 2.0 * (x*y + z*w);       R[4] = 1.0 - 2.0 * (x*x + z*z); R[5] = 2.0 * (y*z - x*w);
  R[6] = 2.0 * (x*z - y*w);       R[7] = 2.0 * (y*z + x*w);       R[8] = 1.0 - 2.0 * (x*x + y*y);
  return p3((R[0]*u[0] + R[1]*u[1]) + R[2]*u[2], (R[3]*u[0] + R[4]*u[1]) + R[5]*u[2], (R[6]*u[0] + R[7]*u[1]) + R[8]*u[2]);
}
MH_DEV int jt_rows(int type) { return (type == 0 || type == 3) ? 3 : (type == 4 ? 4 : ((type == 1 || type == 5) ? 5 : 6)); }   // MH_IJOINT_SPHERICAL 0, _REVOLUTE 1, _FIXED 2, _PLANAR 3, _UNIVERSAL 4, _PRISMATIC 5
MH_DEV int jt_pos_rows(int type) { return type == 3 ? 1 : (type == 5 ? 2 : 3); }
MH_DEV int jt_dir_slot(int type, int k) { return type == 3 ? 2 : k; }
MH_DEV void jt_eval(const JointTab& jt, const double* st, int nb, int j, double* C) {
  const int bi = jt.in[j], bo = jt.out[j];
  const P3 ri = jt_body_vec(st, nb, bi, jt.anchor_in + 3 * j), ro = jt_body_vec(st, nb, bo, jt.anchor_out + 3 * j);
  const P3 pi = (bi >= 0 && bi < nb) ? ld3(st + 13 * bi) + ri : ri, po = (bo >= 0 && bo < nb) ? ld3(st + 13 * bo) + ro : ro;
  const P3 dd = pi - po;
  const int np = jt_pos_rows(jt.type[j]);
  if (np != 3) for (int k = 0; k < np; k++) C[k] = dot3(jt_body_vec(st, nb, bi, jt.vec_in + 9 * j + 3 * jt_dir_slot(jt.type[j], k)), dd);
  else { C[0] = dd.x; C[1] = dd.y; C[2] = dd.z; }
  const int nori = jt_rows(jt.type[j]) - np;
  for (int k = 0; k < nori; k++) C[np + k] = dot3(jt_body_vec(st, nb, bi, jt.vec_in + 9 * j + 3 * k), jt_body_vec(st, nb, bo, jt.vec_out + 9 * j + 3 * k));
}
MH_DEV void jt_jac(const JointTab& jt, const double* st, int nb, int j, bool inboard, double* Cq) {   // rows x 6, row-major
  const int bi = jt.in[j], bo = jt.out[j];
  const P3 r = inboard ? jt_body_vec(st, nb, bi, jt.anchor_in + 3 * j) : jt_body_vec(st, nb, bo, jt.anchor_out + 3 * j);
  const double sg = inboard ? 1.0 : -1.0;
  const int np = jt_pos_rows(jt.type[j]);
  if (np != 3) for (int k = 0; k < np; k++) {
    const P3 u = jt_body_vec(st, nb, bi, jt.vec_in + 9 * j + 3 * jt_dir_slot(jt.type[j], k));
    const P3 ri = jt_body_vec(st, nb, bi, jt.anchor_in + 3 * j), ro = jt_body_vec(st, nb, bo, jt.anchor_out + 3 * j);
    const P3 pi = (bi >= 0 && bi < nb) ? ld3(st + 13 * bi) + ri : ri, po = (bo >= 0 && bo < nb) ? ld3(st + 13 * bo) + ro : ro;
    const P3 e = u * sg;
    P3 ang = cross3(r, e);
    if (inboard) ang = ang + cross3(u, pi - po);
    Cq[6*k] = e.x; Cq[6*k+1] = e.y; Cq[6*k+2] = e.z; Cq[6*k+3] = ang.x; Cq[6*k+4] = ang.y; Cq[6*k+5] = ang.z;
  } else for (int k = 0; k < 3; k++) {
    const P3 e = p3(k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0);
    const P3 rxe = cross3(r, e);
    Cq[6*k] = e.x; Cq[6*k+1] = e.y; Cq[6*k+2] = e.z; Cq[6*k+3] = rxe.x; Cq[6*k+4] = rxe.y; Cq[6*k+5] = rxe.z;
  }
  const int nori = jt_rows(jt.type[j]) - np;
  for (int k = 0; k < nori; k++) {
    P3 axb = cross3(jt_body_vec(st, nb, bi, jt.vec_in + 9 * j + 3 * k), jt_body_vec(st, nb, bo, jt.vec_out + 9 * j + 3 * k));
    if (!inboard) axb = -axb;
    Cq[6*(np+k)] = 0.0; Cq[6*(np+k)+1] = 0.0; Cq[6*(np+k)+2] = 0.0; Cq[6*(np+k)+3] = axb.x; Cq[6*(np+k)+4] = axb.y; Cq[6*(np+k)+5] = axb.z;
  }
}

}} // namespace mh::imp