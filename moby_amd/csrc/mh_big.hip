// mh_big.hip -- many-worlds stepper for LARGE worlds (include/moby_hip_stack.h; BASELINE config 4: box stacks).
//
// One 256-thread workgroup per world runs everything of TimeSteppingSimulator::step that is not an LCP solve:
//   k_mini_pre    do_mini_step up to the impact handler (TSS:114-209): the conservative-advancement loop (broad phase,
//                 pairwise distances, calc_next_CA_Euler_step, position integration with the OLD velocity), forward
//                 dynamics, velocity integration, distances, find_unilateral_constraints -> the world's contact list
//   [island pipeline of mh_impact.hip: process_constraints over every island, LCPs through the LCP entry]
//   k_mini_post   current_time += h, next mini-step?
//   k_stab_begin / k_stab_prep / k_stab_update / k_stab_end
//                 ConstraintStabilization::stabilize (CStab:167-254): compute_problem_data's contact list (synthetic
//                 contacts for separated pairs, CStab:316-343), [island pipeline in MH_CORE_STAB mode: determine_dq],
//                 update_q with Ridders' method and the backtracking line search (CStab:1056-1216, 1322-1379)
// Threads are BODY lanes (integration, swept bounds, forward dynamics), PAIR lanes (overlap test, signed distance,
// conservative-advancement bound, contact generation with a block scan for the reference's list order) phase by
// phase; per-world control flow is block-uniform.  The host only asks, once per mini-step / stabilisation iteration,
// whether any world goes on (one int read back).  Arithmetic order follows oracle/world.hpp statement by statement.
#include <hip/hip_runtime.h>
#include <cmath>
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../include/moby_hip_stack.h"
#include "mh_host.h"
#include "mh_imp_core.h"
#include "mh_imp_dev.h"

namespace mh { namespace big {

using namespace mh::imp;

constexpr int T = 256;
constexpr int NBMAX = MH_BIG_MAX_BODIES;
constexpr int NPMAX = MH_BIG_MAX_PAIRS;
constexpr double NEAR_ZERO_ = 1.4901161193847656e-08;   // Constants.h:21
constexpr double INF_ = 1.7976931348623157e308;         // std::numeric_limits<double>::max()
constexpr double BILATERAL_EPS_ = 1e-6;                 // ConstraintStabilization::bilateral_eps (CStab:62)

struct Dev {
  int B, nb, has_ground, npairs, ncmax, nk;
  const int* geom_type; const double* geom_dim; const double* mass; const double* inertia;
  double plane_R[9], plane_o[3], gravity[3];
  const int* pair_a; const int* pair_b; const int* pair_model;
  const double* cp_eps; const double* cp_mu; const double* cp_muv; const double* cp_comp;
  double min_step, thresh, cstab_eps; unsigned cstab_maxit;
  double* state; double* qsave; double* vsave; double* qstab; double* dq;
  int* ptc; int* nptc;                       // ConstraintSimulator::_pairs_to_check
  mh_contact* contacts; int* ncount; double* cdist;
  double* hdone; double* hmini; int* mini_active; unsigned* guard;
  int* stab_active; unsigned* stab_iter;
  // implicit joints (scene tables) and the jointed islands of Simulator::find_islands, fixed per scene (host, create):
  // island i holds kk_nbod[i] bodies (kk_body + i * MH_IJOINT_MAX_BODIES, ascending ids) and kk_nj[i] joints
  // (kk_joint + i * KKJ, scene order) with kk_m[i] equations; jointed[b] marks the bodies k_mini_pre leaves to k_kkt_fwd
  int nj, kk_nisl, kk_mmax, jrows;           // jrows: equations of all joints (scene-wide C vector, joint order)
  const int* jrow0;                          // first row of each joint in that vector
  const int* jtype; const int* jin; const int* jout; const double* janchor_in; const double* janchor_out; const double* jvec_in; const double* jvec_out;
  const int* kk_nbod; const int* kk_body; const int* kk_nj; const int* kk_joint; const int* kk_m; const unsigned char* jointed;
  double* kk_JiM;                            // B x kk_nisl x (kk_mmax x 6 MH_IJOINT_MAX_BODIES) scratch
  double* time; unsigned long long* steps; unsigned long long* mini_steps; unsigned long long* stab_iters;
  int* status; int* anyflag;
  const int* thrown;          // the island pipeline's (mh_imp_core.h): an exception of the impact handler in the mini-step's process_constraints
};

MH_DEV double norm3(P3 a) { return sqrt(dot3(a, a)); }
MH_DEV double comp3(P3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// ---- block helpers ---------------------------------------------------------------------------------------
__shared__ double s_rd[T];
__shared__ int s_ri[T];
MH_DEV double blk_min(double v) {
  const int t = threadIdx.x;
  s_rd[t] = v; __syncthreads();
  for (int w = T / 2; w > 0; w >>= 1) { if (t < w) s_rd[t] = (s_rd[t + w] < s_rd[t]) ? s_rd[t + w] : s_rd[t]; __syncthreads(); }
  const double r = s_rd[0]; __syncthreads();
  return r;
}
MH_DEV int blk_excl_scan(int v, int& total) {          // thread order; total uniform
  const int t = threadIdx.x;
  s_ri[t] = v; __syncthreads();
  for (int off = 1; off < T; off <<= 1) { const int o = (t >= off) ? s_ri[t - off] : 0; __syncthreads(); s_ri[t] += o; __syncthreads(); }
  const int inc = s_ri[t]; total = s_ri[T - 1]; __syncthreads();
  return inc - v;
}
MH_DEV int blk_or(int v) { int tot; (void)blk_excl_scan(v ? 1 : 0, tot); return tot; }

// ---- one world (oracle World) ----------------------------------------------------------------------------------
struct W {
  const Dev& d; double* st;
  MH_DEV W(const Dev& dd, double* s) : d(dd), st(s) {}
  MH_DEV bool enabled(int b) const { return b >= 0 && b < d.nb; }
  MH_DEV int dyn(int b) const { return enabled(b) ? b : -1; }          // mh_imp_dev's static marker
  MH_DEV P3 X(int b) const { return ld3(st + 13 * b); }
  MH_DEV P3 Vl(int b) const { return ld3(st + 13 * b + 7); }
  MH_DEV P3 Wa(int b) const { return ld3(st + 13 * b + 10); }
  MH_DEV bool is_box(int b) const { return b < d.nb && d.geom_type[b] == MH_GEOM_BOX; }
  MH_DEV bool is_pin(int b) const { return b >= 0 && b < d.nb && d.geom_type[b] == MH_GEOM_PIN; }
  MH_DEV bool vertex_face(int p) const { return d.pair_model[p] == MH_PAIR_VERTEX_FACE; }
  MH_DEV void rot(int b, double* R) const {
    const double x = st[13*b+3], y = st[13*b+4], z = st[13*b+5], w = st[13*b+6];
    R[0] = 1.0 - 2.0 * (y*y + z*z); R[1] = 2.0 * (x*y - z*w);       R[2] = 2.0 * (x*z + y*w);
    R[3] = 2.0 * (x*y + z*w);       R[4] = 1.0 - 2.0 * (x*x + z*z); R[5] = 2.0 * (y*z - x*w);
    R[6] = 2.0 * (x*z - y*w);       R[7] = 2.0 * (y*z + x*w);       R[8] = 1.0 - 2.0 * (x*x + y*y);
  }
  MH_DEV P3 pvel(int b, P3 p) const { return point_vel(st, dyn(b), p); }
  MH_DEV P3 plane_n() const { return p3(d.plane_R[1], d.plane_R[4], d.plane_R[7]); }
  MH_DEV P3 to_plane(P3 p) const {
    const double* R = d.plane_R; const P3 q = p - p3(d.plane_o[0], d.plane_o[1], d.plane_o[2]);
    return p3((R[0]*q.x + R[3]*q.y) + R[6]*q.z, (R[1]*q.x + R[4]*q.y) + R[7]*q.z, (R[2]*q.x + R[5]*q.y) + R[8]*q.z);
  }
  MH_DEV P3 from_plane(P3 p) const {
    const double* R = d.plane_R;
    return p3(d.plane_o[0] + ((R[0]*p.x + R[1]*p.y) + R[2]*p.z), d.plane_o[1] + ((R[3]*p.x + R[4]*p.y) + R[5]*p.z),
              d.plane_o[2] + ((R[6]*p.x + R[7]*p.y) + R[8]*p.z));
  }
  // CCD::construct_bounding_sphere (CCD.cpp:1040-1063)
  MH_DEV double bounding_radius(int b) const {
    if (is_box(b)) return norm3(p3(d.geom_dim[3*b] / 2.0, d.geom_dim[3*b+1] / 2.0, d.geom_dim[3*b+2] / 2.0));
    if (is_pin(b)) return 0.0;
    return d.geom_dim[3*b];
  }
  // BoxPrimitive::get_vertices order (BoxPrimitive.cpp:358-365), global frame
  MH_DEV P3 box_vertex(int b, int i) const {
    const double hx = d.geom_dim[3*b] * 0.5, hy = d.geom_dim[3*b+1] * 0.5, hz = d.geom_dim[3*b+2] * 0.5;
    const double px = (i & 4) ? -hx : hx, py = (i & 2) ? -hy : hy, pz = (i & 1) ? -hz : hz;
    double R[9]; rot(b, R);
    const P3 c = X(b);
    return p3(c.x + ((R[0]*px + R[1]*py) + R[2]*pz), c.y + ((R[3]*px + R[4]*py) + R[5]*pz), c.z + ((R[6]*px + R[7]*py) + R[8]*pz));
  }
  // PendulumColdetPlugin: the body point geom_dim (body frame) in the global frame
  MH_DEV P3 pin_point(int b) const {
    const double px = d.geom_dim[3*b], py = d.geom_dim[3*b+1], pz = d.geom_dim[3*b+2];
    double R[9]; rot(b, R);
    const P3 c = X(b);
    return p3(c.x + ((R[0]*px + R[1]*py) + R[2]*pz), c.y + ((R[3]*px + R[4]*py) + R[5]*pz), c.z + ((R[6]*px + R[7]*py) + R[8]*pz));
  }
  // support plane of a vertex-face pair: through the centre of body L's +Y face, normal = L's +Y axis
  MH_DEV P3 face_n(int L) const { double R[9]; rot(L, R); return p3(R[1], R[4], R[7]); }
  MH_DEV double face_height(int L, P3 p) const { return dot3(face_n(L), p - X(L)) - d.geom_dim[3*L+1] * 0.5; }

  struct PD { int pair, a, b; double dist; P3 pa, pb; };
  // CollisionGeometry::calc_signed_dist (CollisionGeometry.cpp:236-250) and the primitives behind it
  MH_DEV PD signed_dist(int p) const {
    PD r; r.pair = p; r.a = d.pair_a[p]; r.b = d.pair_b[p];
    if (is_pin(r.a)) {                                   // plugin :65-82: -|p|; "pA" = the body point, "pB" = the origin (:130-136)
      const P3 g = pin_point(r.a);
      r.dist = -norm3(g - p3(0.0, 0.0, 0.0));
      r.pa = g; r.pb = p3(0.0, 0.0, 0.0);
      return r;
    }
    if (vertex_face(p)) {
      double md = INF_;
      const P3 nL = face_n(r.a);
      for (int i = 0; i < 8; i++) {
        const P3 g = box_vertex(r.b, i);
        const double h = face_height(r.a, g);
        if (h < md) { md = h; r.pb = g; r.pa = g - nL * h; }
      }
      r.dist = md;
      return r;
    }
    if (is_box(r.a)) {                                   // PlanePrimitive::calc_signed_dist(polyhedral) (PlanePrimitive.cpp:338-376)
      double md = INF_;
      for (int i = 0; i < 8; i++) {
        const P3 g = box_vertex(r.a, i);
        const P3 pp = to_plane(g);
        if (pp.y < md) { md = pp.y; r.pa = g; r.pb = from_plane(p3(pp.x, 0.0, pp.z)); }
      }
      r.dist = md;
      return r;
    }
    if (enabled(r.a) && enabled(r.b)) {                  // SpherePrimitive.cpp:104-136
      const P3 ca = X(r.a), cb = X(r.b);
      const double ra = d.geom_dim[3*r.a], rb = d.geom_dim[3*r.b];
      const P3 ab = cb - ca;
      const double len = norm3(ab);
      r.dist = len - ra - rb;
      const P3 u = ab / len;
      const double sa = (r.dist > 0.0) ? ra : ra + r.dist, sb = (r.dist > 0.0) ? rb : rb + r.dist;
      r.pa = ca + u * sa;
      r.pb = cb - u * sb;
    } else {                                             // PlanePrimitive.cpp:385-411
      const int s = enabled(r.a) ? r.a : r.b;
      const P3 cp = to_plane(X(s));
      const double rad = d.geom_dim[3*s];
      const double low = cp.y + (-1.0 * rad);
      r.dist = low;
      const P3 on_plane = from_plane(p3(cp.x, 0.0, cp.z));
      const P3 on_sphere = from_plane(p3(cp.x, low, cp.z));
      if (s == r.a) { r.pa = on_sphere; r.pb = on_plane; } else { r.pa = on_plane; r.pb = on_sphere; }
    }
    return r;
  }
  // CCD::find_contacts (CCD.inl:3-83): emit(point, normal, geom1, geom2, signed_violation) per contact, in list order
  template <class F> MH_DEV void find_contacts(int p, double TOL, F emit) const {
    const int a = d.pair_a[p], b = d.pair_b[p];
    if (is_pin(a)) {                                     // plugin :84-110: six contacts at the midpoint, violation min(0, -p[axis])
      const P3 g = pin_point(a);
      const P3 pt = (g + p3(0.0, 0.0, 0.0)) * 0.5;
      for (int i = 0; i < 6; i++) {
        const P3 n = (i == 0) ? p3(0, 1, 0) : (i == 1 ? p3(0, -1, 0) : (i == 2 ? p3(0, 0, 1) : (i == 3 ? p3(0, 0, -1) : (i == 4 ? p3(1, 0, 0) : p3(-1, 0, 0)))));
        const double pv = (i < 2) ? g.y : ((i < 4) ? g.z : g.x);
        emit(pt, n, a, b, (-pv < 0.0) ? -pv : 0.0);
      }
      return;
    }
    if (vertex_face(p)) {
      const P3 nL = face_n(a);
      for (int i = 0; i < 8; i++) {
        const P3 g = box_vertex(b, i);
        const double h = face_height(a, g);
        if (!(h <= TOL)) continue;
        emit(g, -nL, a, b, h);
      }
      return;
    }
    if (is_box(a)) {                                     // find_contacts_plane_generic (CCD.inl:848-886)
      for (int i = 0; i < 8; i++) {
        const P3 g = box_vertex(a, i);
        const P3 pp = to_plane(g);
        if (!(pp.y <= TOL)) continue;
        emit(g, -plane_n(), b, a, pp.y);
      }
      return;
    }
    if (enabled(a) && enabled(b)) {                      // CCD.inl:1164-1207
      const P3 cA = X(a), cB = X(b);
      const double rA = d.geom_dim[3*a], rB = d.geom_dim[3*b];
      const P3 dd = cA - cB;
      const double len = norm3(dd);
      const double dist = len - rA - rB;
      if (dist > TOL) return;
      const P3 n = dd / len;
      const P3 closest_A = cA - n * rA, closest_B = cB + n * rB;
      emit((closest_A + closest_B) * 0.5, n, a, b, dist);
    } else {                                             // CCD.inl:804-847
      const int s = enabled(a) ? a : b, pl = enabled(a) ? b : a;
      const P3 cp = to_plane(X(s));
      const double rad = d.geom_dim[3*s];
      const double dist = cp.y - rad;
      if (dist > TOL) return;
      emit(from_plane(p3(cp.x, 0.5 * (cp.y - rad), cp.z)), plane_n(), s, pl, dist);
    }
  }
  MH_DEV double contact_vel(int g1, int g2, P3 p, P3 dir) const { return dot3(dir, pvel(g1, p) - pvel(g2, p)); }
  // CCD::calc_max_dist (CCD.cpp:585-609): velocity at the global origin
  MH_DEV double calc_max_dist(int b, P3 n, double rmax) const {
    if (!enabled(b)) return 0.0;
    const P3 xd0 = Vl(b) + cross3(X(b), Wa(b));
    return dot3(n, xd0) + norm3(cross3(Wa(b), n)) * rmax;
  }
  MH_DEV double rmax_of(int b) const {                   // CCD::_rmax (CCD.cpp:739); box: the FULL diagonal (BoxPrimitive.h:44)
    if (!enabled(b)) return 0.0;
    if (is_box(b)) { const double x = d.geom_dim[3*b], y = d.geom_dim[3*b+1], z = d.geom_dim[3*b+2]; return sqrt((x*x + y*y) + z*z); }
    return d.geom_dim[3*b];
  }
  static MH_DEV bool rel_equal(double x, double y) {     // CompGeom.h:110
    const double ax = fabs(x), ay = fabs(y);
    double m = (ay > 1.0) ? ay : 1.0; m = (ax > m) ? ax : m;
    return fabs(x - y) <= NEAR_ZERO_ * m;
  }
  static MH_DEV bool collinear(P3 a, P3 b, P3 c) {       // CompGeom.cpp:1923-1931
    return rel_equal((c.z-a.z)*(b.y-a.y), (b.z-a.z)*(c.y-a.y)) && rel_equal((b.z-a.z)*(c.x-a.x), (b.x-a.x)*(c.z-a.z)) &&
           rel_equal((b.x-a.x)*(c.y-a.y), (b.y-a.y)*(c.x-a.x));
  }
  // CCD::calc_next_CA_Euler_step_polyhedron_plane (CCD.cpp:410-468)
  MH_DEV double next_CA_box_plane(int bx, P3 normal, double offset0, int support) const {
    double R[9]; rot(bx, R);
    auto tb = [&](P3 v) { return p3((R[0]*v.x + R[3]*v.y) + R[6]*v.z, (R[1]*v.x + R[4]*v.y) + R[7]*v.z, (R[2]*v.x + R[5]*v.y) + R[8]*v.z); };
    const P3 nP = tb(normal);
    const P3 p0 = normal * offset0;
    const double offset = dot3(nP, tb(p0 - X(bx)));
    const P3 wrel = enabled(support) ? Wa(bx) - Wa(support) : Wa(bx);
    const P3 vrel = enabled(support) ? Vl(bx) - pvel(support, X(bx)) : Vl(bx);
    const double av_norm = norm3(tb(wrel));
    const double lv_dot_n = -dot3(nP, tb(vrel));
    const double hx = d.geom_dim[3*bx] * 0.5, hy = d.geom_dim[3*bx+1] * 0.5, hz = d.geom_dim[3*bx+2] * 0.5;
    double max_step = INF_;
    for (int i = 0; i < 8; i++) {
      const P3 vtx = p3((i & 4) ? -hx : hx, (i & 2) ? -hy : hy, (i & 1) ? -hz : hz);
      const double r = norm3(vtx);
      const double dist = dot3(nP, vtx) - offset;
      if (dist < NEAR_ZERO_) continue;
      const double sp = lv_dot_n + av_norm * r;
      const double speed = (0.0 > sp) ? 0.0 : sp;
      const double cand = dist / speed;
      max_step = (cand < max_step) ? cand : max_step;
    }
    return max_step;
  }
  // CCD::calc_next_CA_Euler_step_generic (CCD.cpp:238-405)
  MH_DEV double next_CA_generic(const PD& pd) const {
    if (is_pin(pd.a)) return INF_;                       // PendulumColdetPlugin::calc_next_CA_Euler_step (plugin :139-142)
    int cnt = 0; bool approaching = false;
    P3 c3[3]; P3 n0 = p3(0, 0, 0);
    find_contacts(pd.pair, NEAR_ZERO_, [&](P3 p, P3 n, int g1, int g2, double) {
      if (cnt == 0) n0 = n;
      if (cnt < 3) c3[cnt] = p;
      cnt++;
      if (contact_vel(g1, g2, p, n) < -NEAR_ZERO_) approaching = true;
    });
    if (cnt == 0) return INF_;
    if (approaching) return 0.0;
    if (is_box(pd.a)) {
      if (cnt >= 3 && !collinear(c3[0], c3[1], c3[2])) return INF_;       // only the FIRST three are ever tested (CCD.cpp:313-318)
      const double dd = dot3(n0, c3[0]);
      if (vertex_face(pd.pair)) return next_CA_box_plane(pd.b, -n0, -dd, pd.a);
      return next_CA_box_plane(pd.a, -n0, -dd, -1);
    }
    return INF_;
  }
  // CCD::calc_CA_Euler_step_generic (CCD.cpp:169-235)
  MH_DEV double CA_generic(const PD& pd) const {
    if (pd.dist <= 0.0) return next_CA_generic(pd);
    const P3 d0 = pd.pa - pd.pb;
    const P3 n0 = d0 / norm3(d0);
    const double tA = calc_max_dist(pd.a, -n0, rmax_of(pd.a));
    const double tB = calc_max_dist(pd.b, n0, rmax_of(pd.b));
    double total = tA + tB;
    if (total < 0.0) total = 0.0;
    const double cand = pd.dist / total;
    return (cand < INF_) ? cand : INF_;
  }
  // CCD::calc_CA_Euler_step_sphere (CCD.cpp:138-166)
  MH_DEV double CA_step(const PD& pd) const {
    if (is_box(pd.a) || is_pin(pd.a)) return CA_generic(pd);
    if (pd.dist > NEAR_ZERO_) return CA_generic(pd);
    int cnt = 0; double v0 = 0.0;
    find_contacts(pd.pair, NEAR_ZERO_, [&](P3 p, P3 n, int g1, int g2, double) { if (cnt == 0) v0 = contact_vel(g1, g2, p, n); cnt++; });
    if (cnt == 1 && fabs(v0) < NEAR_ZERO_ * 10) return INF_;
    return CA_generic(pd);
  }
  // generalized velocity in eEuler form: [xd ; qd], qd = 1/2 (0,w) (x) q
  MH_DEV void euler_vel(int b, double* qd) const {
    const P3 v = Vl(b), w = Wa(b);
    const double x = st[13*b+3], y = st[13*b+4], z = st[13*b+5], ww = st[13*b+6];
    qd[0] = v.x; qd[1] = v.y; qd[2] = v.z;
    qd[3] = 0.5 * ((ww * w.x + z * w.y) - y * w.z);
    qd[4] = 0.5 * ((ww * w.y + x * w.z) - z * w.x);
    qd[5] = 0.5 * ((ww * w.z + y * w.x) - x * w.y);
    qd[6] = 0.5 * (((-x * w.x) - y * w.y) - z * w.z);
  }
  MH_DEV void set_coords(int b, const double* q) {       // stores x and the NORMALISED quaternion
    for (int i = 0; i < 3; i++) st[13*b+i] = q[i];
    const double nrm = sqrt(((q[3]*q[3] + q[4]*q[4]) + q[5]*q[5]) + q[6]*q[6]);
    for (int i = 3; i < 7; i++) st[13*b+i] = q[i] / nrm;
  }
};

// CCD::broad_phase (CCD.cpp:702-876) over the candidate list -> list[] (pair order), count returned (uniform)
__shared__ double s_lo[3 * (NBMAX + 1)], s_hi[3 * (NBMAX + 1)];
MH_DEV int broad_phase(const W& w, double dt, int* list) {
  const Dev& d = w.d; const int t = threadIdx.x;
  const int ntot = d.nb + (d.has_ground ? 1 : 0);
  for (int b = t; b < ntot; b += T) {
    if (!w.enabled(b)) { for (int k = 0; k < 3; k++) { s_lo[3*b+k] = -INF_; s_hi[3*b+k] = INF_; } continue; }
    const P3 c = w.X(b);
    const P3 vdt = w.Vl(b) * dt, wdt = w.Wa(b) * dt;
    const P3 lin = vdt + cross3(c, wdt);
    const P3 p2 = c + lin;
    const double r = w.bounding_radius(b);
    for (int k = 0; k < 3; k++) {
      const double a = comp3(c, k), e = comp3(p2, k);
      s_lo[3*b+k] = ((a < e) ? a : e) - r;
      s_hi[3*b+k] = ((a > e) ? a : e) + r;
    }
  }
  __syncthreads();
  int keep = 0;
  const int p = t;                                         // npairs <= T: one pair per thread
  if (p < d.npairs) {
    const int i = d.pair_a[p], j = d.pair_b[p];
    bool ov = true;
    for (int k = 0; k < 3; k++) if (!(s_lo[3*i+k] <= s_hi[3*j+k] && s_lo[3*j+k] <= s_hi[3*i+k])) ov = false;
    keep = (ov && (w.enabled(i) || w.enabled(j))) ? 1 : 0;
    if (w.is_pin(i) || w.is_pin(j)) keep = (w.is_pin(i) && j == d.nb) ? 1 : 0;   // the plugin's broad phase: always, and nothing else (:53-58)
  }
  int total;
  const int o = blk_excl_scan(keep, total);
  if (keep) list[o] = p;
  __syncthreads();
  return total;
}

// ---------------------------------------------------------------------------------------------------------
// TimeSteppingSimulator::do_mini_step (TSS:114-222) up to the contact list
__global__ __launch_bounds__(T)
void k_mini_pre(Dev d, double dt_step)
{
  const int b = blockIdx.x, t = threadIdx.x;
  if (!d.mini_active[b]) { if (t == 0) d.ncount[b] = 0; return; }
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int nb = d.nb;
  double* qsave = d.qsave + (size_t)b * nb * 7;
  int* ptc = d.ptc + (size_t)b * d.npairs;
  __shared__ double s_dist[NPMAX];
  __shared__ int s_stall;
  const double dt = dt_step - d.hdone[b];
  for (int i = t; i < nb * 7; i += T) qsave[i] = w.st[13 * (i / 7) + (i % 7)];
  if (t == 0) s_stall = 0;
  __syncthreads();
  double h = 0.0;
  unsigned ca_guard = 0;
  int np = d.nptc[b];
  while (h < dt) {
    if (++ca_guard > MH_CA_HARD_CAP) { if (t == 0) s_stall = 1; break; }     // a step that no longer advances h would spin forever
    np = broad_phase(w, dt - h, ptc);
    double ca = INF_;
    if (t < np) { const W::PD pd = w.signed_dist(ptc[t]); ca = w.CA_step(pd); }
    const double CA = blk_min(ca);
    if (CA <= 0.0) break;
    double tc = (d.min_step > CA) ? d.min_step : CA;
    tc = ((dt - h) < tc) ? (dt - h) : tc;
    __syncthreads();
    for (int bb = t; bb < nb; bb += T) {
      w.set_coords(bb, qsave + 7 * bb);
      double qd[7], q[7]; w.euler_vel(bb, qd);
      for (int i = 0; i < 7; i++) { q[i] = qd[i] * (h + tc); q[i] = q[i] + qsave[7 * bb + i]; }
      w.set_coords(bb, q);
    }
    h += tc;
    __syncthreads();
  }
  __syncthreads();
  // forward dynamics + velocity integration by h (TSS:173-192): xdd = (g m) / m ; wd = Jw^-1 (0 - w x (Jw w))
  for (int bb = t; bb < nb; bb += T) {
    if (d.jointed && d.jointed[bb]) continue;            // an island with implicit joints: Simulator::solve (k_kkt_fwd)
    const double m = d.mass[bb];
    const P3 f = p3(d.gravity[0] * m, d.gravity[1] * m, d.gravity[2] * m);
    const P3 xdd = f / m;
    double xi[10], Jw[9];
    inv_inertia(w.st + 13 * bb, d.inertia + 3 * bb, m, xi, Jw);
    const P3 om = w.Wa(bb);
    const P3 Jww = p3((Jw[0]*om.x + Jw[1]*om.y) + Jw[2]*om.z, (Jw[3]*om.x + Jw[4]*om.y) + Jw[5]*om.z, (Jw[6]*om.x + Jw[7]*om.y) + Jw[8]*om.z);
    const P3 tau = -cross3(om, Jww);
    const double* Ji = xi + 1;
    const P3 wd = p3((Ji[0]*tau.x + Ji[1]*tau.y) + Ji[2]*tau.z, (Ji[3]*tau.x + Ji[4]*tau.y) + Ji[5]*tau.z, (Ji[6]*tau.x + Ji[7]*tau.y) + Ji[8]*tau.z);
    const P3 v1 = w.Vl(bb) + xdd * h, w1 = om + wd * h;
    double* s = w.st + 13 * bb;
    s[7] = v1.x; s[8] = v1.y; s[9] = v1.z; s[10] = w1.x; s[11] = w1.y; s[12] = w1.z;
  }
  __syncthreads();
  // calc_pairwise_distances (TSS:206), find_unilateral_constraints (CSim:488-537)
  int cnt = 0;
  if (t < np) {
    const W::PD pd = w.signed_dist(ptc[t]);
    s_dist[t] = pd.dist;
    if (pd.dist < d.thresh) w.find_contacts(ptc[t], d.thresh, [&](P3, P3, int, int, double) { cnt++; });
  }
  int total;
  int o = blk_excl_scan(cnt, total);
  if (total > d.ncmax) { if (t == 0) d.status[b] |= MH_WORLD_UNSUPPORTED; total = 0; cnt = 0; }
  if (cnt > 0) {
    const int p = ptc[t];
    mh_contact* C = d.contacts + (size_t)b * d.ncmax;
    w.find_contacts(p, d.thresh, [&](P3 pt, P3 n, int g1, int g2, double) {
      mh_contact& c = C[o++];
      c.point[0] = pt.x; c.point[1] = pt.y; c.point[2] = pt.z; c.normal[0] = n.x; c.normal[1] = n.y; c.normal[2] = n.z;
      c.body1 = g1; c.body2 = g2;
      c.mu_coulomb = d.cp_mu[p]; c.mu_viscous = d.cp_muv[p]; c.epsilon = d.cp_eps[p]; c.compliance = d.cp_comp[p]; c.nk = d.nk; c.pad = 0;
    });
  }
  if (t == 0) { d.ncount[b] = total; d.nptc[b] = np; d.hmini[b] = h; if (s_stall) d.status[b] |= MH_WORLD_STALLED; }
}

// ---------------------------------------------------------------------------------------------------------
// Simulator::calc_fwd_dyn for the islands with implicit joints (Sim:482-602): Simulator::solve (Sim:608-805) and the
// velocity integration of do_mini_step (TSS:181-192).  One 64-thread workgroup per (island, world); the operation order
// is oracle/world.hpp::solve_kkt's (dense products from 0 over ascending indices; the greedy full-rank selection is ONE
// incremental Cholesky: a row appended at the end leaves the leading factor unchanged, so re-factorising from scratch
// for every candidate, as the reference does, gives the same numbers).
constexpr int KKB = MH_IJOINT_MAX_BODIES, KKM = MH_IJOINT_MAX_EQNS, KKJ = MH_IJOINT_MAX_JOINTS, KKT_T = 64;
MH_DEV P3 body_vec(const W& w, int b, const double* u) {      // R u for a dynamic body, u for the static world
  if (!w.enabled(b)) return p3(u[0], u[1], u[2]);
  double R[9]; w.rot(b, R);
  return p3((R[0]*u[0] + R[1]*u[1]) + R[2]*u[2], (R[3]*u[0] + R[4]*u[1]) + R[5]*u[2], (R[6]*u[0] + R[7]*u[1]) + R[8]*u[2]);
}
MH_DEV int joint_rows(int type) { return (type == MH_IJOINT_SPHERICAL || type == MH_IJOINT_PLANAR) ? 3 : (type == MH_IJOINT_UNIVERSAL ? 4 : ((type == MH_IJOINT_REVOLUTE || type == MH_IJOINT_PRISMATIC) ? 5 : 6)); }
MH_DEV int joint_pos_rows(int type) { return type == MH_IJOINT_PLANAR ? 1 : (type == MH_IJOINT_PRISMATIC ? 2 : 3); }
MH_DEV int joint_dir_slot(int type, int k) { return type == MH_IJOINT_PLANAR ? 2 : k; }
// calc_constraint_jacobian (oracle World::joint_jac): rows x 6, row-major, into Cq[36]
MH_DEV void joint_jac(const W& w, int j, bool inboard, double* Cq) {
  const Dev& d = w.d;
  const int bi = d.jin[j], bo = d.jout[j];
  const P3 r = inboard ? body_vec(w, bi, d.janchor_in + 3 * j) : body_vec(w, bo, d.janchor_out + 3 * j);
  const double sg = inboard ? 1.0 : -1.0;
  const int np = joint_pos_rows(d.jtype[j]);
  if (np != 3) for (int k = 0; k < np; k++) {             // planar / prismatic: rows along inboard-fixed directions
    const P3 u = body_vec(w, bi, d.jvec_in + 9 * j + 3 * joint_dir_slot(d.jtype[j], k));
    const P3 ri = body_vec(w, bi, d.janchor_in + 3 * j), ro = body_vec(w, bo, d.janchor_out + 3 * j);
    const P3 pi = w.enabled(bi) ? w.X(bi) + ri : ri, po = w.enabled(bo) ? w.X(bo) + ro : ro;
    const P3 e = u * sg;
    P3 ang = cross3(r, e);
    if (inboard) ang = ang + cross3(u, pi - po);
    Cq[6*k] = e.x; Cq[6*k+1] = e.y; Cq[6*k+2] = e.z; Cq[6*k+3] = ang.x; Cq[6*k+4] = ang.y; Cq[6*k+5] = ang.z;
  } else for (int k = 0; k < 3; k++) {
    const P3 e = p3(k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0);
    const P3 rxe = cross3(r, e);
    Cq[6*k] = e.x; Cq[6*k+1] = e.y; Cq[6*k+2] = e.z; Cq[6*k+3] = rxe.x; Cq[6*k+4] = rxe.y; Cq[6*k+5] = rxe.z;
  }
  const int nori = joint_rows(d.jtype[j]) - np;
  for (int k = 0; k < nori; k++) {
    P3 axb = cross3(body_vec(w, bi, d.jvec_in + 9 * j + 3 * k), body_vec(w, bo, d.jvec_out + 9 * j + 3 * k));
    if (!inboard) axb = -axb;
    Cq[6*(np+k)] = 0.0; Cq[6*(np+k)+1] = 0.0; Cq[6*(np+k)+2] = 0.0; Cq[6*(np+k)+3] = axb.x; Cq[6*(np+k)+4] = axb.y; Cq[6*(np+k)+5] = axb.z;
  }
}

// evaluate_constraints (oracle World::joint_eval)
MH_DEV void joint_eval(const W& w, int j, double* C) {
  const Dev& d = w.d;
  const int bi = d.jin[j], bo = d.jout[j];
  const P3 ri = body_vec(w, bi, d.janchor_in + 3 * j), ro = body_vec(w, bo, d.janchor_out + 3 * j);
  const P3 pi = w.enabled(bi) ? w.X(bi) + ri : ri, po = w.enabled(bo) ? w.X(bo) + ro : ro;
  const P3 dd = pi - po;
  const int np = joint_pos_rows(d.jtype[j]);
  if (np != 3) for (int k = 0; k < np; k++) C[k] = dot3(body_vec(w, bi, d.jvec_in + 9 * j + 3 * joint_dir_slot(d.jtype[j], k)), dd);
  else { C[0] = dd.x; C[1] = dd.y; C[2] = dd.z; }
  const int nori = joint_rows(d.jtype[j]) - np;
  for (int k = 0; k < nori; k++) C[np + k] = dot3(body_vec(w, bi, d.jvec_in + 9 * j + 3 * k), body_vec(w, bo, d.jvec_out + 9 * j + 3 * k));
}
constexpr int JROWS = MH_BIG_MAX_JOINT_ROWS;
// evaluate_bilateral_constraints (CStab:133-160): C of every joint -> s_c[0 .. jrows), returns max |C| (uniform)
MH_DEV double eval_bilateral(const W& w, double* s_c) {
  const Dev& d = w.d; const int t = threadIdx.x;
  double mx = 0.0;
  for (int j = t; j < d.nj; j += T) {
    double c6[6]; joint_eval(w, j, c6);
    const int rows = joint_rows(d.jtype[j]);
    for (int k = 0; k < rows; k++) { s_c[d.jrow0[j] + k] = c6[k]; const double a = fabs(c6[k]); mx = (a > mx) ? a : mx; }
  }
  return -blk_min(-mx);
}

__global__ __launch_bounds__(KKT_T)
void k_kkt_fwd(Dev d)
{
  const int isl = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  if (!d.mini_active[b]) return;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const double dt = d.hmini[b];                          // calc_fwd_dyn(h), then v += a h (TSS:179-192)
  if (!(dt > 0.0)) return;                               // h = 0 (conservative advancement at a resting contact): a * 0, the island keeps its velocities (deviation 9)
  const int nbod = d.kk_nbod[isl], njl = d.kk_nj[isl], m = d.kk_m[isl], ngc = 6 * nbod;
  const int* bodies = d.kk_body + isl * KKB;
  const int* joints = d.kk_joint + isl * KKJ;
  __shared__ double s_iM[KKB * 36], s_f[6 * KKB], s_v[6 * KKB], s_iMf[6 * KKB];
  __shared__ double s_w[2 * KKJ * 36];                   // Jacobian blocks: joint jl, side sd at (2 jl + sd) * 36
  __shared__ int s_brow[2 * KKJ], s_boff[2 * KKJ], s_brows[2 * KKJ];   // first row, gc offset (-1: static side), rows
  __shared__ double s_A[KKM * KKM], s_L[KKM * KKM];      // J iM J' (row-major, ld m) and its Cholesky factor (ld m)
  __shared__ double s_JiMf[KKM], s_Jv[KKM], s_lam[KKM];
  __shared__ int s_act[KKM], s_nact;
  double* JiM = d.kk_JiM + ((size_t)b * d.kk_nisl + isl) * ((size_t)d.kk_mmax * 6 * KKB);    // m x ngc, row-major
  for (int i = t; i < nbod; i += KKT_T) {
    const int bb = bodies[i];
    double xi[10], Jw[9];
    inv_inertia(w.st + 13 * bb, d.inertia + 3 * bb, d.mass[bb], xi, Jw);
    double* Bm = s_iM + 36 * i;
    for (int k = 0; k < 36; k++) Bm[k] = 0.0;
    for (int k = 0; k < 3; k++) Bm[7 * k] = xi[0];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Bm[6 * (3 + r) + 3 + c] = xi[1 + 3 * r + c];
    const double mm = d.mass[bb];
    double* f = s_f + 6 * i;
    f[0] = d.gravity[0] * mm; f[1] = d.gravity[1] * mm; f[2] = d.gravity[2] * mm;
    const P3 om = w.Wa(bb);
    const P3 Jww = p3((Jw[0]*om.x + Jw[1]*om.y) + Jw[2]*om.z, (Jw[3]*om.x + Jw[4]*om.y) + Jw[5]*om.z, (Jw[6]*om.x + Jw[7]*om.y) + Jw[8]*om.z);
    const P3 tau = -cross3(om, Jww);
    f[3] = tau.x; f[4] = tau.y; f[5] = tau.z;
    for (int k = 0; k < 6; k++) s_v[6 * i + k] = w.st[13 * bb + 7 + k];
    for (int r = 0; r < 6; r++) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + Bm[6 * r + k] * f[k]; s_iMf[6 * i + r] = acc * dt; }
  }
  if (t == 0) {                                          // block table: rows of each joint side
    int eq = 0;
    for (int jl = 0; jl < njl; jl++) {
      const int j = joints[jl], rows = joint_rows(d.jtype[j]);
      const int sides[2] = { d.jin[j], d.jout[j] };
      for (int sd = 0; sd < 2; sd++) {
        int off = -1;
        if (w.enabled(sides[sd])) for (int i = 0; i < nbod; i++) if (bodies[i] == sides[sd]) off = 6 * i;
        s_brow[2 * jl + sd] = eq; s_boff[2 * jl + sd] = off; s_brows[2 * jl + sd] = rows;
      }
      eq += rows;
    }
  }
  __syncthreads();
  for (int k = t; k < 2 * njl; k += KKT_T) if (s_boff[k] >= 0) joint_jac(w, joints[k >> 1], (k & 1) == 0, s_w + 36 * k);
  for (int e = t; e < m * ngc; e += KKT_T) JiM[e] = 0.0;
  __syncthreads();
  // JiM = J iM: block k covers rows s_brow .. + rows, columns s_boff .. + 6
  for (int k = 0; k < 2 * njl; k++) {
    if (s_boff[k] < 0) continue;
    const double* Bm = s_iM + 36 * (s_boff[k] / 6);
    for (int e = t; e < s_brows[k] * 6; e += KKT_T) {
      const int r = e / 6, c = e - 6 * r;
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * r + q] * Bm[6 * q + c];
      JiM[(size_t)(s_brow[k] + r) * ngc + s_boff[k] + c] = acc;
    }
  }
  __syncthreads();
  // J iM J' (row, c): the row's blocks in order (inboard, outboard), each a 6-term product; J v and J iM f dt likewise
  for (int e = t; e < m * m; e += KKT_T) {
    const int row = e / m, c = e - row * m;
    double tot = 0.0;
    for (int k = 0; k < 2 * njl; k++) {
      if (s_boff[k] < 0 || row < s_brow[k] || row >= s_brow[k] + s_brows[k]) continue;
      const int r = row - s_brow[k];
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * r + q] * JiM[(size_t)c * ngc + s_boff[k] + q];
      tot = tot + acc;
    }
    s_A[row * m + c] = tot;
  }
  for (int row = t; row < m; row += KKT_T) {
    double tot = 0.0;
    for (int k = 0; k < 2 * njl; k++) {
      if (s_boff[k] < 0 || row < s_brow[k] || row >= s_brow[k] + s_brows[k]) continue;
      const int r = row - s_brow[k];
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * r + q] * s_v[s_boff[k] + q];
      tot = tot + acc;
    }
    s_Jv[row] = tot;
    double acc = 0.0;
    for (int g = 0; g < ngc; g++) acc = acc + JiM[(size_t)row * ngc + g] * s_f[g];
    s_JiMf[row] = acc * dt;
  }
  __syncthreads();
  if (t == 0) {
    // the biggest full-rank leading set (Sim:728-755), dpotf2 'L' arithmetic row by row; then lambda (Sim:769-789)
    int k = 0;
    for (int i = 0; i < m; i++) {
      if (k == ngc) break;
      bool ok = true;
      for (int j = 0; j < k; j++) {
        double tv = s_A[i * m + s_act[j]];
        for (int p = 0; p < j; p++) tv = tv - s_L[k * m + p] * s_L[j * m + p];
        s_L[k * m + j] = tv / s_L[j * m + j];
      }
      double ajj = s_A[i * m + i];
      for (int p = 0; p < k; p++) ajj = ajj - s_L[k * m + p] * s_L[k * m + p];
      if (!(ajj > 0.0)) ok = false;
      if (ok) { s_L[k * m + k] = sqrt(ajj); s_act[k] = i; k++; }
    }
    for (int r = 0; r < k; r++) s_lam[r] = s_JiMf[s_act[r]] + s_Jv[s_act[r]];
    for (int c = 0; c < k; c++) {                        // chol_solve: L y = b, L' x = y
      s_lam[c] = s_lam[c] / s_L[c * m + c];
      const double bk = s_lam[c];
      for (int i = c + 1; i < k; i++) s_lam[i] = s_lam[i] - bk * s_L[i * m + c];
    }
    for (int c = k - 1; c >= 0; c--) {
      double tv = s_lam[c];
      for (int i = c + 1; i < k; i++) tv = tv - s_L[i * m + c] * s_lam[i];
      s_lam[c] = tv / s_L[c * m + c];
    }
    s_nact = k;
  }
  __syncthreads();
  const int k = s_nact;
  for (int g = t; g < ngc; g += KKT_T) {
    double acc = 0.0;
    for (int r = 0; r < k; r++) acc = acc + JiM[(size_t)s_act[r] * ngc + g] * s_lam[r];
    const double a = ((-acc) + s_iMf[g]) / dt;
    const int bb = bodies[g / 6], q = g - 6 * (g / 6);
    w.st[13 * bb + 7 + q] = w.st[13 * bb + 7 + q] + a * dt;
  }
}

// the tail of do_mini_step (TSS:215) and of the loop in step_si_Euler (TSS:433-455)
__global__ void k_mini_post(Dev d, double dt_step)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B || !d.mini_active[b]) return;
  // an exception of the impact handler (mh_imp_core.h: thrown) leaves do_mini_step before TSS:215 and step_si_Euler's loop with it: the world's run is over
  if (d.thrown[b]) { d.mini_active[b] = 0; return; }
  const double h = d.hmini[b];
  d.time[b] += h; d.mini_steps[b] += 1ull;
  const double hd = d.hdone[b] + h;
  d.hdone[b] = hd;
  bool more = hd < dt_step;
  if (more && ++d.guard[b] > 100000u) { d.status[b] |= MH_WORLD_STALLED; more = false; }   // the reference would spin forever
  d.mini_active[b] = more ? 1 : 0;
  if (more) atomicOr(d.anyflag, 1);
}

__global__ void k_step_begin(Dev d)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  // a world that carries MH_WORLD_LCP_FAILED is not stepped: the exception ended the run of the simulator that owned it (oracle World::step)
  d.hdone[b] = 0.0; d.guard[b] = 0u; d.mini_active[b] = (d.status[b] & MH_WORLD_LCP_FAILED) ? 0 : 1;
}

// pairwise distances on the simulator's pair list -> s_uc[0 .. np), returns the minimum (CStab:88-131)
MH_DEV double eval_unilateral(const W& w, const int* ptc, int np, double* s_uc) {
  const int t = threadIdx.x;
  double v = INF_;
  if (t < np) { v = w.signed_dist(ptc[t]).dist; s_uc[t] = v; }
  return blk_min(v);
}
MH_DEV void set_q_scaled(W& w, const double* q, const double* dq, double tt, int nb) {   // qs = dq * t + q ; set_q(qs)
  const int t = threadIdx.x;
  for (int bb = t; bb < nb; bb += T) {
    double qs[7];
    for (int k = 0; k < 7; k++) { qs[k] = dq[7 * bb + k] * tt; qs[k] = qs[k] + q[7 * bb + k]; }
    w.set_coords(bb, qs);
  }
  __syncthreads();
}

// ConstraintStabilization::stabilize, before the loop (CStab:181-191)
__global__ __launch_bounds__(T)
void k_stab_begin(Dev d, int in_step)
{
  const int b = blockIdx.x, t = threadIdx.x;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int nb = d.nb;
  __shared__ double s_uc[NPMAX];
  if (d.cstab_maxit == 0u) { if (t == 0) d.stab_active[b] = 0; return; }
  if (in_step && (d.status[b] & MH_WORLD_LCP_FAILED)) { if (t == 0) d.stab_active[b] = 0; return; }     // (step() was left by the exception: TSS:97 is not reached)
  for (int i = t; i < nb * 6; i += T) d.vsave[(size_t)b * nb * 6 + i] = w.st[13 * (i / 6) + 7 + (i % 6)];
  for (int i = t; i < nb * 7; i += T) d.qstab[(size_t)b * nb * 7 + i] = w.st[13 * (i / 7) + (i % 7)];
  const double mu = eval_unilateral(w, d.ptc + (size_t)b * d.npairs, d.nptc[b], s_uc);
  __shared__ double s_c[JROWS];
  const double mb = (d.nj > 0) ? eval_bilateral(w, s_c) : 0.0;
  if (t == 0) { d.stab_iter[b] = 0u; const int act = (mu < d.cstab_eps || mb > BILATERAL_EPS_) ? 1 : 0; d.stab_active[b] = act; if (act) atomicOr(d.anyflag, 1); }
}

// top of the loop + compute_problem_data's contact list (CStab:197-221, 306-343, 364-378)
__global__ __launch_bounds__(T)
void k_stab_prep(Dev d)
{
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) d.ncount[b] = 0;
  if (!d.stab_active[b]) return;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int nb = d.nb;
  __shared__ int s_cp[NPMAX];
  __shared__ int s_stop;
  if (t == 0) {
    s_stop = 0;
    const unsigned it = d.stab_iter[b];
    if (it == d.cstab_maxit) s_stop = 1;
    else if (it == MH_CSTAB_HARD_CAP) { d.status[b] |= MH_WORLD_STALLED; s_stop = 1; }   // the reference's default cap is UINT_MAX
    if (s_stop) d.stab_active[b] = 0;
  }
  __syncthreads();
  if (s_stop) return;
  for (int i = t; i < nb * 6; i += T) w.st[13 * (i / 6) + 7 + (i % 6)] = 0.0;
  __syncthreads();
  const int np = broad_phase(w, 0.0, s_cp);                 // its own broad phase with dt = 0
  int cnt = 0;
  W::PD pd;
  if (t < np) {
    pd = w.signed_dist(s_cp[t]);
    if (pd.dist >= NEAR_ZERO_) cnt = 1;                       // separated: one synthetic contact (CStab:316-331)
    else w.find_contacts(s_cp[t], NEAR_ZERO_, [&](P3, P3, int, int, double) { cnt++; });
  }
  int total;
  int o = blk_excl_scan(cnt, total);
  if (total > d.ncmax) { if (t == 0) d.status[b] |= MH_WORLD_UNSUPPORTED; total = 0; cnt = 0; }
  if (cnt > 0) {
    const int p = s_cp[t];
    mh_contact* C = d.contacts + (size_t)b * d.ncmax;
    double* cd = d.cdist + (size_t)b * d.ncmax;
    auto put = [&](P3 pt, P3 n, int g1, int g2, double dist) {
      cd[o] = dist;
      mh_contact& c = C[o++];
      c.point[0] = pt.x; c.point[1] = pt.y; c.point[2] = pt.z; c.normal[0] = n.x; c.normal[1] = n.y; c.normal[2] = n.z;
      c.body1 = g1; c.body2 = g2;
      c.mu_coulomb = d.cp_mu[p]; c.mu_viscous = d.cp_muv[p]; c.epsilon = d.cp_eps[p]; c.compliance = d.cp_comp[p]; c.nk = d.nk; c.pad = 0;
    };
    if (pd.dist >= NEAR_ZERO_) { const P3 nn = pd.pb - pd.pa; put(pd.pa, nn / norm3(nn), pd.a, pd.b, pd.dist); }
    else w.find_contacts(p, NEAR_ZERO_, put);
  }
  if (t == 0) d.ncount[b] = total;
}

// The islands tied by implicit joints and touched by no contact of the stabiliser's list ("remaining islands", UC:1158-1191;
// the jointed islands are fixed per scene): set_bilateral_only_constraint_data (CStab:531-700) -- Jfull, the greedy
// full-rank set on J J' - sqrt(eps) I (ICH:1698-1739), J iM and J iM J' on the active rows (ICH:1657-1660), Jx_v = C
// (CStab:475-486) -- then update_from_stacked's bilateral step (ICH:356-374): (J iM J') lambda = C, v = 0 + (0 - iM J' lambda).
// A jointed island that a contact touches belongs to a contact island: mh_impact.hip's k_bilat_X (compute_X's general case).
// Operation order: oracle World::build_bilateral / bilateral_only_dq.
__global__ __launch_bounds__(KKT_T)
void k_stab_bilat(Dev d)
{
  const int isl = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  if (!d.stab_active[b]) return;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int nbod = d.kk_nbod[isl], njl = d.kk_nj[isl], m = d.kk_m[isl], ngc = 6 * nbod;
  const int* bodies = d.kk_body + isl * KKB;
  const int* joints = d.kk_joint + isl * KKJ;
  __shared__ double s_iM[KKB * 36];
  __shared__ double s_w[2 * KKJ * 36];
  __shared__ int s_brow[2 * KKJ], s_boff[2 * KKJ], s_brows[2 * KKJ];
  __shared__ double s_A[KKM * KKM], s_L[KKM * KKM];      // J J', then J iM J' on the active rows; the Cholesky factor
  __shared__ double s_C[KKM], s_lam[KKM];
  __shared__ int s_act[KKM], s_nact, s_touched, s_fail;
  double* JiM = d.kk_JiM + ((size_t)b * d.kk_nisl + isl) * ((size_t)d.kk_mmax * 6 * KKB);
  if (t == 0) { s_touched = 0; s_fail = 0; }
  __syncthreads();
  {
    const mh_contact* C = d.contacts + (size_t)b * d.ncmax;
    const int ncnt = d.ncount[b];
    for (int i = t; i < ncnt; i += KKT_T) {
      const int g1 = C[i].body1, g2 = C[i].body2;
      for (int k = 0; k < nbod; k++) if (bodies[k] == g1 || bodies[k] == g2) s_touched = 1;
    }
  }
  __syncthreads();
  if (s_touched) return;                                 // part of a contact island: the island pipeline's k_bilat_X (general compute_X)
  for (int i = t; i < nbod; i += KKT_T) {
    const int bb = bodies[i];
    double xi[10];
    inv_inertia(w.st + 13 * bb, d.inertia + 3 * bb, d.mass[bb], xi);
    double* Bm = s_iM + 36 * i;
    for (int k = 0; k < 36; k++) Bm[k] = 0.0;
    for (int k = 0; k < 3; k++) Bm[7 * k] = xi[0];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Bm[6 * (3 + r) + 3 + c] = xi[1 + 3 * r + c];
  }
  if (t == 0) {
    int eq = 0;
    for (int jl = 0; jl < njl; jl++) {
      const int j = joints[jl], rows = joint_rows(d.jtype[j]);
      const int sides[2] = { d.jin[j], d.jout[j] };
      for (int sd = 0; sd < 2; sd++) {
        int off = -1;
        if (w.enabled(sides[sd])) for (int i = 0; i < nbod; i++) if (bodies[i] == sides[sd]) off = 6 * i;
        s_brow[2 * jl + sd] = eq; s_boff[2 * jl + sd] = off; s_brows[2 * jl + sd] = rows;
      }
      eq += rows;
    }
  }
  for (int jl = t; jl < njl; jl += KKT_T) {               // Jx_v: the island's rows of C (joint order)
    double c6[6]; joint_eval(w, joints[jl], c6);
    int eq = 0;
    for (int k = 0; k < jl; k++) eq += joint_rows(d.jtype[joints[k]]);
    for (int k = 0; k < joint_rows(d.jtype[joints[jl]]); k++) s_C[eq + k] = c6[k];
  }
  __syncthreads();
  for (int k = t; k < 2 * njl; k += KKT_T) if (s_boff[k] >= 0) joint_jac(w, joints[k >> 1], (k & 1) == 0, s_w + 36 * k);
  for (int e = t; e < m * ngc; e += KKT_T) JiM[e] = 0.0;
  __syncthreads();
  // J J' (r, c): block pairs on the same body, in block order
  for (int e = t; e < m * m; e += KKT_T) {
    const int row = e / m, c = e - row * m;
    double tot = 0.0;
    for (int k = 0; k < 2 * njl; k++) {
      if (s_boff[k] < 0 || row < s_brow[k] || row >= s_brow[k] + s_brows[k]) continue;
      for (int k2 = 0; k2 < 2 * njl; k2++) {
        if (s_boff[k2] != s_boff[k] || c < s_brow[k2] || c >= s_brow[k2] + s_brows[k2]) continue;
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * (row - s_brow[k]) + q] * s_w[36 * k2 + 6 * (c - s_brow[k2]) + q];
        tot = tot + acc;
      }
    }
    s_A[row * m + c] = tot;
  }
  for (int k = 0; k < 2 * njl; k++) {                     // J iM
    if (s_boff[k] < 0) continue;
    const double* Bm = s_iM + 36 * (s_boff[k] / 6);
    for (int e = t; e < s_brows[k] * 6; e += KKT_T) {
      const int r = e / 6, c = e - 6 * r;
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * r + q] * Bm[6 * q + c];
      JiM[(size_t)(s_brow[k] + r) * ngc + s_boff[k] + c] = acc;
    }
  }
  __syncthreads();
  if (t == 0) {                                          // get_full_rank_implicit_constraints: incremental Cholesky of J J' - sqrt(eps) I
    int k = 0;
    for (int i = 0; i < m; i++) {
      if (k == ngc) break;
      for (int j = 0; j < k; j++) {
        double tv = s_A[i * m + s_act[j]];
        for (int p = 0; p < j; p++) tv = tv - s_L[k * m + p] * s_L[j * m + p];
        s_L[k * m + j] = tv / s_L[j * m + j];
      }
      double ajj = s_A[i * m + i] - NEAR_ZERO_;
      for (int p = 0; p < k; p++) ajj = ajj - s_L[k * m + p] * s_L[k * m + p];
      if (ajj > 0.0) { s_L[k * m + k] = sqrt(ajj); s_act[k] = i; k++; }
    }
    s_nact = k;
  }
  __syncthreads();
  const int k = s_nact;
  // J iM J' on the active rows -> s_A (k x k, ld m)
  for (int e = t; e < k * k; e += KKT_T) {
    const int r = e / k, c = e - r * k, row = s_act[r];
    double tot = 0.0;
    for (int kb = 0; kb < 2 * njl; kb++) {
      if (s_boff[kb] < 0 || row < s_brow[kb] || row >= s_brow[kb] + s_brows[kb]) continue;
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * kb + 6 * (row - s_brow[kb]) + q] * JiM[(size_t)s_act[c] * ngc + s_boff[kb] + q];
      tot = tot + acc;
    }
    s_L[r * m + c] = tot;                                  // (the greedy factor is not needed any more)
  }
  __syncthreads();
  if (t == 0) {                                          // factor_chol + solve_chol_fast (ICH:365-367): dpotf2 'L', then L y = C, L' x = y
    bool ok = true;
    for (int j = 0; j < k && ok; j++) {
      double ajj = s_L[j * m + j];
      for (int p = 0; p < j; p++) ajj = ajj - s_L[j * m + p] * s_L[j * m + p];
      if (!(ajj > 0.0)) { ok = false; break; }
      ajj = sqrt(ajj);
      s_L[j * m + j] = ajj;
      for (int i = j + 1; i < k; i++) {
        double tv = s_L[i * m + j];
        for (int p = 0; p < j; p++) tv = tv - s_L[i * m + p] * s_L[j * m + p];
        s_L[i * m + j] = tv / ajj;
      }
    }
    if (!ok) { s_fail = 1; d.status[b] |= MH_WORLD_STAB_FAILED; }
    else {
      for (int r = 0; r < k; r++) s_lam[r] = s_C[s_act[r]];
      for (int c = 0; c < k; c++) {
        s_lam[c] = s_lam[c] / s_L[c * m + c];
        const double bk = s_lam[c];
        for (int i = c + 1; i < k; i++) s_lam[i] = s_lam[i] - bk * s_L[i * m + c];
      }
      for (int c = k - 1; c >= 0; c--) {
        double tv = s_lam[c];
        for (int i = c + 1; i < k; i++) tv = tv - s_L[i * m + c] * s_lam[i];
        s_lam[c] = tv / s_L[c * m + c];
      }
    }
  }
  __syncthreads();
  if (s_fail) return;
  for (int g = t; g < ngc; g += KKT_T) {
    double acc = 0.0;
    for (int r = 0; r < k; r++) acc = acc + JiM[(size_t)s_act[r] * ngc + g] * s_lam[r];
    const int bb = bodies[g / 6], q = g - 6 * (g / 6);
    w.st[13 * bb + 7 + q] = 0.0 + (0.0 - acc);
  }
}

// determine_dq's read-back, update_q, the loop's tail (CStab:962-969, 1056-1216, 230-242)
__global__ __launch_bounds__(T)
void k_stab_update(Dev d)
{
  const int b = blockIdx.x, t = threadIdx.x;
  if (!d.stab_active[b]) return;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int nb = d.nb;
  const int* ptc = d.ptc + (size_t)b * d.npairs; const int np = d.nptc[b];
  double* q = d.qstab + (size_t)b * nb * 7;
  double* dq = d.dq + (size_t)b * nb * 7;
  __shared__ double s_uc[NPMAX], s_old[NPMAX], s_full[NPMAX];
  __shared__ unsigned char s_br[NPMAX];
  __shared__ double s_c[JROWS], s_cold[JROWS], s_cfull[JROWS], s_cvio;
  const int jrows = d.jrows;
  // the bodies' velocities are X Cn^T z (- iM J' lambda) now: dq = their eEuler form (bodies outside every island stay at 0)
  for (int bb = t; bb < nb; bb += T) { double qd[7]; w.euler_vel(bb, qd); for (int k = 0; k < 7; k++) dq[7 * bb + k] = qd[k]; }
  __syncthreads();
  auto cvio_of = [&](const double* c) -> double {           // sqrt(sum C^2), summed in row order (CStab:1068-1071, 1182-1185)
    if (t == 0) { double a = 0.0; for (int i = 0; i < jrows; i++) a = a + c[i] * c[i]; s_cvio = sqrt(a); }
    __syncthreads();
    const double r = s_cvio;
    __syncthreads();
    return r;
  };
  // ---- update_q (CStab:1056-1216) ----
  (void)eval_unilateral(w, ptc, np, s_old);
  double old_cvio = 0.0;
  if (d.nj > 0) { (void)eval_bilateral(w, s_cold); old_cvio = cvio_of(s_cold); }
  for (int bb = t; bb < nb; bb += T) { double qs[7]; for (int k = 0; k < 7; k++) { qs[k] = dq[7 * bb + k]; qs[k] = qs[k] + q[7 * bb + k]; } w.set_coords(bb, qs); }
  __syncthreads();
  (void)eval_unilateral(w, ptc, np, s_full);
  if (t < np) s_br[t] = ((s_old[t] < 0.0 && s_full[t] > 0.0) || (s_old[t] > 0.0 && s_full[t] < 0.0)) ? 1 : 0;
  if (d.nj > 0) (void)eval_bilateral(w, s_cfull);
  __syncthreads();
  auto eval_at = [&](double x, int idx) -> double {          // CStab:1281-1298
    set_q_scaled(w, q, dq, x, nb);
    (void)eval_unilateral(w, ptc, np, s_uc);
    const double r = s_uc[idx];
    __syncthreads();
    return r;
  };
  auto sign2 = [](double x, double y) { return (y > 0.0) ? fabs(x) : -fabs(x); };
  // ridders_unilateral (CStab:1322-1379)
  auto ridders = [&](double x1, double x2, double fl, double fh, int idx) -> double {
    const double TOL = 1e-4;
    double ans = INF_, fm, fnew, s, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = eval_at(xm, idx);
        s = sqrt(fm * fm - fl * fh);
        if (s == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s);
        ans = xnew;
        fnew = eval_at(ans, idx);
        if (fabs(fnew) < TOL && fnew >= 0.0) return xnew;
        if (sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
        else if (sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
        else if (sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
      }
    } else {
      if (fl == 0.0) return x1;
      if (fh == 0.0) return x2;
    }
    return 0.0;
  };
  auto eval_b_at = [&](double x, int idx) -> double {        // CStab:1300-1318
    set_q_scaled(w, q, dq, x, nb);
    (void)eval_bilateral(w, s_c);
    const double r = s_c[idx];
    __syncthreads();
    return r;
  };
  // ridders_bilateral (CStab:1382-1439)
  auto ridders_b = [&](double x1, double x2, double fl, double fh, int idx) -> double {
    const double TOL = 1e-6;
    double ans = INF_, fm, fnew, s, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = eval_b_at(xm, idx);
        s = sqrt(fm * fm - fl * fh);
        if (s == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s);
        ans = xnew;
        fnew = eval_b_at(ans, idx);
        if (fabs(fnew) < TOL) return ans;
        if (sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
        else if (sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
        else if (sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
      }
    } else {
      if (fl == 0.0) return x1;
      if (fh == 0.0) return x2;
    }
    return 0.0;
  };
  double tt = 1.0;
  for (int i = 0; i < np; i++) {
    if (!s_br[i]) continue;
    const double root = ridders(0.0, tt, s_old[i], s_full[i], i);
    if (root > 0.0 && root < 1.0) tt = (root < tt) ? root : tt;
  }
  for (int i = 0; i < jrows; i++) {                          // CStab:1132-1145
    const double co = s_cold[i], cf = s_cfull[i];
    if (!((cf < 0.0 && co > 0.0) || (cf > 0.0 && co < 0.0))) continue;
    const double root = ridders_b(0.0, tt, co, cf, i);
    if (root > 0.0 && root < 1.0) tt = (root < tt) ? root : tt;
  }
  set_q_scaled(w, q, dq, tt, nb);
  (void)eval_unilateral(w, ptc, np, s_uc);
  if (d.nj > 0) (void)eval_bilateral(w, s_c);
  const double BETA = 0.6;
  bool failed = false;
  while (true) {
    const int worse = (t < np && !s_br[t] && s_uc[t] < 0.0 && s_old[t] > s_uc[t]) ? 1 : 0;
    if (!blk_or(worse)) {                                    // CStab:1180-1192 (no joints: cvio = 0 < bilateral_eps)
      if (d.nj == 0) break;
      const double cvio = cvio_of(s_c);
      if (cvio < BILATERAL_EPS_ || cvio < old_cvio) break;
    }
    tt *= BETA;
    if (tt < NEAR_ZERO_) { failed = true; break; }
    set_q_scaled(w, q, dq, tt, nb);
    if (d.nj > 0) (void)eval_bilateral(w, s_c);
    (void)eval_unilateral(w, ptc, np, s_uc);
  }
  if (failed) { if (t == 0) { d.status[b] |= MH_WORLD_STAB_FAILED; d.stab_active[b] = 0; } return; }
  __syncthreads();
  for (int i = t; i < nb * 7; i += T) { double v = dq[i] * tt; v = v + q[i]; q[i] = v; }      // q = qstar
  const double mu = eval_unilateral(w, ptc, np, s_uc);
  const double mb = (d.nj > 0) ? eval_bilateral(w, s_c) : 0.0;
  if (t == 0) {
    d.stab_iter[b] += 1u; d.stab_iters[b] += 1ull;
    const int act = (mu < d.cstab_eps || mb > BILATERAL_EPS_) ? 1 : 0;
    d.stab_active[b] = act;
    if (act) atomicOr(d.anyflag, 1);
  }
}

// restore the generalized velocities (CStab:246); in step mode the step is complete
__global__ void k_stab_end(Dev d, int count_step)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const int nb = d.nb;
  if (count_step && (d.status[b] & MH_WORLD_LCP_FAILED)) return;      // (nothing was saved, the step is not counted: see k_stab_begin)
  if (d.cstab_maxit != 0u) {
    double* st = d.state + (size_t)b * nb * 13;
    for (int i = t; i < nb * 6; i += blockDim.x) st[13 * (i / 6) + 7 + (i % 6)] = d.vsave[(size_t)b * nb * 6 + i];
  }
  if (t == 0 && count_step) d.steps[b] += 1ull;
}

// a standalone stabilize(): the simulator's pair list is the broad phase of the resident state (dt = 0)
__global__ __launch_bounds__(T)
void k_pairs_now(Dev d)
{
  const int b = blockIdx.x;
  W w(d, d.state + (size_t)b * d.nb * 13);
  const int np = broad_phase(w, 0.0, d.ptc + (size_t)b * d.npairs);
  if (threadIdx.x == 0) d.nptc[b] = np;
}

}} // namespace mh::big

// ---------------------------------------------------------------------------------------------------------
struct mh_big_batch {
  int device;                // the HIP device the batch lives on (current at create); every entry point runs there (MH_ON_DEVICE)
  mh::big::Dev d;
  mh_imp_core core;
  int B, nb, cap;
  std::vector<void*> allocs;
  uint32_t* d_rng;
  int* h_flag;        // pinned
};

namespace {
int read_flag(mh_big_batch* bb, hipStream_t s, int* out)
{
  MH_HIP(hipMemcpyAsync(bb->h_flag, bb->d.anyflag, sizeof(int), hipMemcpyDeviceToHost, s));
  MH_HIP(hipStreamSynchronize(s));
  *out = *bb->h_flag;
  return MH_OK;
}

// ConstraintStabilization::stabilize for every world (the pair list is already in d.ptc)
int run_stabilize(mh_big_batch* bb, hipStream_t s, int count_step)
{
  namespace bg = mh::big;
  const int B = bb->B;
  MH_HIP(hipMemsetAsync(bb->d.anyflag, 0, sizeof(int), s));
  hipLaunchKernelGGL(bg::k_stab_begin, dim3(B), dim3(bg::T), 0, s, bb->d, count_step);
  MH_HIP(hipGetLastError());
  int any = 0;
  int rc = read_flag(bb, s, &any);
  if (rc != MH_OK) return rc;
  while (any) {
    MH_HIP(hipMemsetAsync(bb->d.anyflag, 0, sizeof(int), s));
    hipLaunchKernelGGL(bg::k_stab_prep, dim3(B), dim3(bg::T), 0, s, bb->d);
    MH_HIP(hipGetLastError());
    rc = mh_imp_core_process(&bb->core, s, MH_CORE_STAB);
    if (rc != MH_OK) return rc;
    if (bb->d.kk_nisl > 0) hipLaunchKernelGGL(bg::k_stab_bilat, dim3(bb->d.kk_nisl, B), dim3(bg::KKT_T), 0, s, bb->d);
    hipLaunchKernelGGL(bg::k_stab_update, dim3(B), dim3(bg::T), 0, s, bb->d);
    MH_HIP(hipGetLastError());
    rc = read_flag(bb, s, &any);
    if (rc != MH_OK) return rc;
  }
  hipLaunchKernelGGL(bg::k_stab_end, dim3(B), dim3(64), 0, s, bb->d, count_step);
  MH_HIP(hipGetLastError());
  return MH_OK;
}
} // namespace

extern "C" {

int mh_big_batch_device(const mh_big_batch* bb) { return bb ? bb->device : fail(MH_ERR_INVALID_ARG, "null batch"); }

int mh_big_batch_destroy(mh_big_batch* bb)
{
  if (!bb) return MH_OK;
  MH_ON_DEVICE(bb);
  (void)hipDeviceSynchronize();
  mh_imp_core_destroy(&bb->core);
  for (void* p : bb->allocs) if (p) (void)hipFree(p);
  if (bb->h_flag) (void)hipHostFree(bb->h_flag);
  delete bb;
  return MH_OK;
}

int mh_big_batch_lcp_capacity(const mh_big_batch* bb) { return bb ? bb->cap : 0; }

int mh_big_batch_lu_work(mh_big_batch* bb, double* work, int reset)
{
  if (!bb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(bb);
  return mh_imp_core_lu_work(&bb->core, work, reset);
}

int mh_big_batch_create(const mh_big_scene* sc, int B, mh_big_batch** out)
{
  namespace bg = mh::big;
  if (!out) return fail(MH_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (!sc) return fail(MH_ERR_INVALID_ARG, "null scene");
  if (B <= 0) return fail(MH_ERR_INVALID_ARG, "batch must be > 0");
  const int nb = sc->nb, np = sc->npairs;
  if (nb < 1 || nb > MH_BIG_MAX_BODIES) return fail(MH_ERR_INVALID_ARG, "nb = %d outside [1, %d]", nb, MH_BIG_MAX_BODIES);
  if (np < 0 || np > MH_BIG_MAX_PAIRS) return fail(MH_ERR_INVALID_ARG, "npairs = %d outside [0, %d]", np, MH_BIG_MAX_PAIRS);
  if (!sc->geom_type || !sc->geom_dim || !sc->mass || !sc->inertia) return fail(MH_ERR_INVALID_ARG, "null body table");
  if (np > 0 && (!sc->pair_a || !sc->pair_b || !sc->pair_model || !sc->cp_epsilon || !sc->cp_mu_coulomb || !sc->cp_mu_viscous || !sc->cp_compliance))
    return fail(MH_ERR_INVALID_ARG, "null pair table");
  if (sc->nk < 4 || (sc->nk & 1)) return fail(MH_ERR_INVALID_ARG, "nk must be even and >= 4 (ContactParameters.cpp:128-135), got %d", sc->nk);
  for (int b = 0; b < nb; b++) {
    if (sc->geom_type[b] != MH_GEOM_SPHERE && sc->geom_type[b] != MH_GEOM_BOX && sc->geom_type[b] != MH_GEOM_PIN) return fail(MH_ERR_INVALID_ARG, "body %d: geometry type %d is not built here (sphere, box, pin)", b, sc->geom_type[b]);
    if ((sc->geom_type[b] != MH_GEOM_PIN && !(sc->geom_dim[3*b] > 0.0)) || !(sc->mass[b] > 0.0)) return fail(MH_ERR_INVALID_ARG, "body %d: size and mass must be > 0", b);
    if (sc->geom_type[b] == MH_GEOM_BOX && (!(sc->geom_dim[3*b+1] > 0.0) || !(sc->geom_dim[3*b+2] > 0.0))) return fail(MH_ERR_INVALID_ARG, "body %d: box edge lengths must be > 0", b);
    for (int k = 0; k < 3; k++) if (!(sc->inertia[3*b+k] > 0.0)) return fail(MH_ERR_INVALID_ARG, "body %d: inertia must be > 0", b);
  }
  int ncmax = 0;
  for (int p = 0; p < np; p++) {
    const int a = sc->pair_a[p], bq = sc->pair_b[p];
    if (!(0 <= a && a < bq && bq <= nb) || (bq == nb && !sc->has_ground)) return fail(MH_ERR_INVALID_ARG, "pair %d: (%d, %d) is not a < b <= nb", p, a, bq);
    if (p > 0 && !(sc->pair_a[p-1] < a || (sc->pair_a[p-1] == a && sc->pair_b[p-1] < bq))) return fail(MH_ERR_INVALID_ARG, "pair %d: the list must be sorted lexicographically, without repeats", p);
    const bool boxa = sc->geom_type[a] == MH_GEOM_BOX, boxb = bq < nb && sc->geom_type[bq] == MH_GEOM_BOX;
    const bool pina = sc->geom_type[a] == MH_GEOM_PIN, pinb = bq < nb && sc->geom_type[bq] == MH_GEOM_PIN;
    if ((pina && bq != nb) || pinb) return fail(MH_ERR_INVALID_ARG, "pair %d: a pin geometry only pairs with the static world body", p);
    if (sc->pair_model[p] == MH_PAIR_VERTEX_FACE) { if (!(boxa && boxb)) return fail(MH_ERR_INVALID_ARG, "pair %d: the vertex-face model needs two boxes", p); }
    else if (sc->pair_model[p] != MH_PAIR_CLOSED_FORM) return fail(MH_ERR_INVALID_ARG, "pair %d: unknown model %d", p, sc->pair_model[p]);
    else if ((boxa || boxb) && bq != nb) return fail(MH_ERR_INVALID_ARG, "pair %d: box-box / box-sphere contact is only built as MH_PAIR_VERTEX_FACE", p);
    ncmax += (boxa || boxb) ? 4 : (pina ? 3 : 1);                        // a box face rests on at most 4 vertices ... of a box in general position; 8 fit below
  }
  // implicit joints: tables, then the islands of Simulator::find_islands (Sim:956-1045) that contain a joint -- bodies
  // connected by joints whose two links are dynamic, from the lowest id, sorted (Sim:501); a joint belongs to the island
  // of either of its dynamic links (Sim:506-520)
  const int nj = sc->njoints;
  std::vector<int> kk_nbod, kk_body, kk_nj, kk_joint, kk_m;
  std::vector<unsigned char> jointed((size_t)nb, 0);
  int kk_mmax = 0;
  if (nj < 0) return fail(MH_ERR_INVALID_ARG, "njoints = %d", nj);
  if (nj > 0) {
    if (!sc->joint_type || !sc->joint_inboard || !sc->joint_outboard || !sc->joint_anchor_in || !sc->joint_anchor_out || !sc->joint_vec_in || !sc->joint_vec_out)
      return fail(MH_ERR_INVALID_ARG, "null joint table");
    if (nj > MH_BIG_MAX_JOINTS) return fail(MH_ERR_INVALID_ARG, "njoints = %d > %d", nj, MH_BIG_MAX_JOINTS);
    std::vector<std::vector<int> > adj((size_t)nb);
    for (int j = 0; j < nj; j++) {
      const int a = sc->joint_inboard[j], bq = sc->joint_outboard[j], ty = sc->joint_type[j];
      if (ty != MH_IJOINT_SPHERICAL && ty != MH_IJOINT_REVOLUTE && ty != MH_IJOINT_FIXED && ty != MH_IJOINT_PLANAR && ty != MH_IJOINT_UNIVERSAL && ty != MH_IJOINT_PRISMATIC) return fail(MH_ERR_INVALID_ARG, "joint %d: type %d (MH_IJOINT_*)", j, ty);
      if (a < 0 || a > nb || bq < 0 || bq > nb || a == bq) return fail(MH_ERR_INVALID_ARG, "joint %d: links (%d, %d) must be two different ids in [0, nb]", j, a, bq);
      if (a < nb && bq < nb) { adj[a].push_back(bq); adj[bq].push_back(a); }
    }
    std::vector<char> seen((size_t)nb, 0);
    for (int s0 = 0; s0 < nb; s0++) {
      if (seen[s0]) continue;
      std::vector<int> q; q.push_back(s0); seen[s0] = 1;
      for (size_t qi = 0; qi < q.size(); qi++) for (int nbr : adj[q[qi]]) if (!seen[nbr]) { seen[nbr] = 1; q.push_back(nbr); }
      std::sort(q.begin(), q.end());
      std::vector<int> ij; int m = 0;
      for (int j = 0; j < nj; j++) {
        const int a = sc->joint_inboard[j], bq = sc->joint_outboard[j];
        if ((a < nb && std::binary_search(q.begin(), q.end(), a)) || (bq < nb && std::binary_search(q.begin(), q.end(), bq))) {
          ij.push_back(j); m += (sc->joint_type[j] == MH_IJOINT_SPHERICAL || sc->joint_type[j] == MH_IJOINT_PLANAR) ? 3 : (sc->joint_type[j] == MH_IJOINT_UNIVERSAL ? 4 : ((sc->joint_type[j] == MH_IJOINT_REVOLUTE || sc->joint_type[j] == MH_IJOINT_PRISMATIC) ? 5 : 6));
        }
      }
      if (ij.empty()) continue;
      if ((int)q.size() > MH_IJOINT_MAX_BODIES || (int)ij.size() > MH_IJOINT_MAX_JOINTS || m > MH_IJOINT_MAX_EQNS)
        return fail(MH_ERR_UNSUPPORTED_N, "jointed island of %d bodies, %d joints, %d equations (limits %d, %d, %d)", (int)q.size(), (int)ij.size(), m,
                    MH_IJOINT_MAX_BODIES, MH_IJOINT_MAX_JOINTS, MH_IJOINT_MAX_EQNS);
      kk_nbod.push_back((int)q.size()); kk_nj.push_back((int)ij.size()); kk_m.push_back(m);
      q.resize(MH_IJOINT_MAX_BODIES, 0); ij.resize(MH_IJOINT_MAX_JOINTS, 0);
      kk_body.insert(kk_body.end(), q.begin(), q.end()); kk_joint.insert(kk_joint.end(), ij.begin(), ij.end());
      for (int i = 0; i < kk_nbod.back(); i++) jointed[(size_t)q[i]] = 1;
      if (m > kk_mmax) kk_mmax = m;
    }
  }
  ncmax *= 2;                                               // a box lying inside the tolerance band can put all 8 vertices in contact
  if (ncmax < 8) ncmax = 8;
  if (ncmax > MH_BIG_MAX_CONTACTS) ncmax = MH_BIG_MAX_CONTACTS;
  long cap = sc->lcp_n_max;
  if (cap == 0) { cap = 6L * ncmax + (long)ncmax * (sc->nk / 2); if (cap > MH_LCP_MAX_N_BLOCK) cap = MH_LCP_MAX_N_BLOCK; }
  if (cap < 1 || cap > MH_LCP_MAX_N_BLOCK) return fail(MH_ERR_UNSUPPORTED_N, "lcp_n_max = %ld outside [1, %d]", cap, MH_LCP_MAX_N_BLOCK);
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  mh_big_batch* bb = new mh_big_batch();
  if (hipGetDevice(&bb->device) != hipSuccess) { delete bb; return fail(MH_ERR_HIP, "hipGetDevice failed"); }
  bb->B = B; bb->nb = nb; bb->cap = (int)cap; bb->d_rng = nullptr; bb->h_flag = nullptr;
  std::memset(&bb->d, 0, sizeof(bb->d)); std::memset(&bb->core, 0, sizeof(bb->core));
  int rc = mh_imp_core_create(&bb->core, B, nb, ncmax, sc->nk, (int)cap);
  if (rc != MH_OK) { delete bb; return rc; }
  if (sc->impact_model != MH_IMPACT_MODEL_DS && sc->impact_model != MH_IMPACT_MODEL_AP) {
    mh_imp_core_destroy(&bb->core); delete bb;
    return fail(MH_ERR_INVALID_ARG, "impact_model %d (MH_IMPACT_MODEL_DS / _AP)", sc->impact_model);
  }
  bb->core.ap = (sc->impact_model == MH_IMPACT_MODEL_AP) ? 1 : 0;
  bool okall = true;
  auto A = [&](size_t bytes, bool zero) -> void* {
    void* p = nullptr;
    if (!okall) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { okall = false; return nullptr; }
    bb->allocs.push_back(p);
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) okall = false;
    return p;
  };
  auto U = [&](const void* src, size_t bytes) -> void* {
    void* p = A(bytes, false);
    if (p && bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) okall = false;
    return p;
  };
  bg::Dev& d = bb->d;
  d.B = B; d.nb = nb; d.has_ground = sc->has_ground ? 1 : 0; d.npairs = np; d.ncmax = ncmax; d.nk = sc->nk;
  d.geom_type = (const int*)U(sc->geom_type, nb * 4); d.geom_dim = (const double*)U(sc->geom_dim, nb * 24);
  d.mass = (const double*)U(sc->mass, nb * 8); d.inertia = (const double*)U(sc->inertia, nb * 24);
  std::memcpy(d.plane_R, sc->plane_R, sizeof(d.plane_R)); std::memcpy(d.plane_o, sc->plane_o, sizeof(d.plane_o)); std::memcpy(d.gravity, sc->gravity, sizeof(d.gravity));
  d.pair_a = (const int*)U(sc->pair_a, np * 4); d.pair_b = (const int*)U(sc->pair_b, np * 4); d.pair_model = (const int*)U(sc->pair_model, np * 4);
  d.cp_eps = (const double*)U(sc->cp_epsilon, np * 8); d.cp_mu = (const double*)U(sc->cp_mu_coulomb, np * 8);
  d.cp_muv = (const double*)U(sc->cp_mu_viscous, np * 8); d.cp_comp = (const double*)U(sc->cp_compliance, np * 8);
  d.min_step = sc->min_step_size; d.thresh = sc->contact_dist_thresh; d.cstab_eps = sc->cstab_eps; d.cstab_maxit = sc->cstab_max_iterations;
  d.nj = nj; d.kk_nisl = (int)kk_nbod.size(); d.kk_mmax = kk_mmax;
  if (nj > 0) {
    std::vector<int> jrow0((size_t)nj); int rows = 0;
    for (int j = 0; j < nj; j++) { jrow0[(size_t)j] = rows; rows += (sc->joint_type[j] == MH_IJOINT_SPHERICAL || sc->joint_type[j] == MH_IJOINT_PLANAR) ? 3 : (sc->joint_type[j] == MH_IJOINT_UNIVERSAL ? 4 : ((sc->joint_type[j] == MH_IJOINT_REVOLUTE || sc->joint_type[j] == MH_IJOINT_PRISMATIC) ? 5 : 6)); }
    d.jrows = rows; d.jrow0 = (const int*)U(jrow0.data(), nj * 4);

    d.jtype = (const int*)U(sc->joint_type, nj * 4); d.jin = (const int*)U(sc->joint_inboard, nj * 4); d.jout = (const int*)U(sc->joint_outboard, nj * 4);
    d.janchor_in = (const double*)U(sc->joint_anchor_in, nj * 24); d.janchor_out = (const double*)U(sc->joint_anchor_out, nj * 24);
    d.jvec_in = (const double*)U(sc->joint_vec_in, nj * 72); d.jvec_out = (const double*)U(sc->joint_vec_out, nj * 72);
    d.jointed = (const unsigned char*)U(jointed.data(), nb);
    if (d.kk_nisl > 0) {
      d.kk_nbod = (const int*)U(kk_nbod.data(), kk_nbod.size() * 4); d.kk_body = (const int*)U(kk_body.data(), kk_body.size() * 4);
      d.kk_nj = (const int*)U(kk_nj.data(), kk_nj.size() * 4); d.kk_joint = (const int*)U(kk_joint.data(), kk_joint.size() * 4);
      d.kk_m = (const int*)U(kk_m.data(), kk_m.size() * 4);
      d.kk_JiM = (double*)A((size_t)B * d.kk_nisl * kk_mmax * 6 * MH_IJOINT_MAX_BODIES * 8, true);
    }
    // joint edges of the constraint islands (UC:993-1008) and the stabiliser's general compute_X
    if (okall) {
      rc = mh_imp_core_enable_joints(&bb->core, nj, d.jtype, d.jin, d.jout, d.janchor_in, d.janchor_out, d.jvec_in, d.jvec_out, d.jointed);
      if (rc != MH_OK) okall = false;
    }
  }
  const size_t sB = (size_t)B;
  d.state = (double*)A(sB * nb * 13 * 8, true); d.qsave = (double*)A(sB * nb * 7 * 8, true); d.vsave = (double*)A(sB * nb * 6 * 8, true);
  d.qstab = (double*)A(sB * nb * 7 * 8, true); d.dq = (double*)A(sB * nb * 7 * 8, true);
  d.ptc = (int*)A(sB * (np ? np : 1) * 4, true); d.nptc = (int*)A(sB * 4, true);
  d.contacts = (mh_contact*)A(sB * ncmax * sizeof(mh_contact), true); d.ncount = (int*)A(sB * 4, true); d.cdist = (double*)A(sB * ncmax * 8, true);
  d.hdone = (double*)A(sB * 8, true); d.hmini = (double*)A(sB * 8, true); d.mini_active = (int*)A(sB * 4, true); d.guard = (unsigned*)A(sB * 4, true);
  d.stab_active = (int*)A(sB * 4, true); d.stab_iter = (unsigned*)A(sB * 4, true);
  d.time = (double*)A(sB * 8, true); d.steps = (unsigned long long*)A(sB * 8, true); d.mini_steps = (unsigned long long*)A(sB * 8, true);
  d.stab_iters = (unsigned long long*)A(sB * 8, true);
  d.status = (int*)A(sB * 4, true); d.anyflag = (int*)A(4, true);
  bb->d_rng = (uint32_t*)A(sB * MH_RAND_WORDS * 4, false);
  if (okall && hipHostMalloc((void**)&bb->h_flag, sizeof(int)) != hipSuccess) okall = false;
  if (okall) {
    std::vector<uint32_t> hr((size_t)B * MH_RAND_WORDS);
    mh_rand_seed(hr.data(), 1u);
    for (int b = 1; b < B; b++) std::memcpy(&hr[(size_t)b * MH_RAND_WORDS], hr.data(), MH_RAND_WORDS * 4);
    if (hipMemcpy(bb->d_rng, hr.data(), hr.size() * 4, hipMemcpyHostToDevice) != hipSuccess) okall = false;
  }
  if (!okall) { mh_big_batch_destroy(bb); return fail(MH_ERR_HIP, "device allocation / upload failed"); }
  mh_imp_core& c = bb->core;
  c.mass = d.mass; c.inertia = d.inertia; c.state = d.state; c.contacts = d.contacts; c.ncount = d.ncount; c.cdist = d.cdist;
  c.stab_eps = d.cstab_eps; c.rng = bb->d_rng; c.status = d.status; d.thrown = c.thrown;
  *out = bb;
  return MH_OK;
}

int mh_big_batch_upload(mh_big_batch* bb, const double* state, const mh_world_aux* aux)
{
  if (!bb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(bb);
  const size_t B = (size_t)bb->B;
  MH_HIP(hipDeviceSynchronize());
  if (state) MH_HIP(hipMemcpy(bb->d.state, state, B * bb->nb * 13 * 8, hipMemcpyHostToDevice));
  if (aux) {
    std::vector<uint32_t> rng(B * MH_RAND_WORDS); std::vector<double> tm(B); std::vector<int> st(B), zs(3 * B);
    std::vector<unsigned long long> steps(B), minis(B), stabs(B), cnt(5 * B);
    for (size_t b = 0; b < B; b++) {
      std::memcpy(&rng[b * MH_RAND_WORDS], aux[b].rng, MH_RAND_WORDS * 4);
      tm[b] = aux[b].time; st[b] = aux[b].status; steps[b] = aux[b].steps; minis[b] = aux[b].mini_steps; stabs[b] = aux[b].stab_iters;
      cnt[5*b] = aux[b].lcp_solves; cnt[5*b+1] = aux[b].lcp_rows; cnt[5*b+2] = aux[b].lcp_pivots; cnt[5*b+3] = aux[b].lcp_alg_bytes; cnt[5*b+4] = aux[b].stab_rows;
    }
    MH_HIP(hipMemcpy(bb->d_rng, rng.data(), rng.size() * 4, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->d.time, tm.data(), B * 8, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->d.status, st.data(), B * 4, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->d.steps, steps.data(), B * 8, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->d.mini_steps, minis.data(), B * 8, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->d.stab_iters, stabs.data(), B * 8, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->core.cnt, cnt.data(), B * 40, hipMemcpyHostToDevice));
    // the handler's warm-start SIZES travel with aux (as in the one-wavefront world batch): a fresh aux (all zero) therefore makes
    // the next solve cold whatever ran on this batch before; the vectors themselves (_zlast, the storage of _z) are
    // mh_big_batch_load_solver_state's -- a resume is download + save_solver_state, then upload + load_solver_state
    { std::vector<int> zl(B), zs2(B), zc(B);
      for (size_t b = 0; b < B; b++) { zl[b] = aux[b].zlast_size; zs2[b] = aux[b].zbuf_size; zc[b] = aux[b].zbuf_cap;
        if (zl[b] < 0 || zl[b] > bb->cap || zs2[b] < 0 || zc[b] < 0 || zc[b] > bb->cap)
          return fail(MH_ERR_INVALID_ARG, "aux[%zu]: warm-start sizes (%d, %d, %d) outside the batch's LCP capacity %d", b, zl[b], zs2[b], zc[b], bb->cap); }
      MH_HIP(hipMemcpy(bb->core.zlast_size, zl.data(), B * 4, hipMemcpyHostToDevice));
      MH_HIP(hipMemcpy(bb->core.zbuf_size, zs2.data(), B * 4, hipMemcpyHostToDevice));
      MH_HIP(hipMemcpy(bb->core.zbuf_cap, zc.data(), B * 4, hipMemcpyHostToDevice)); }
    std::vector<int> vsz(B); std::vector<double> vns(B * MH_NOSLIP_MAX);
    for (size_t b = 0; b < B; b++) { vsz[b] = aux[b].vns_size; std::memcpy(&vns[b * MH_NOSLIP_MAX], aux[b].vns, MH_NOSLIP_MAX * 8); }
    MH_HIP(hipMemcpy(bb->core.vns_size, vsz.data(), B * 4, hipMemcpyHostToDevice));
    MH_HIP(hipMemcpy(bb->core.vns, vns.data(), B * MH_NOSLIP_MAX * 8, hipMemcpyHostToDevice));
  }
  return MH_OK;
}

int mh_big_batch_download(mh_big_batch* bb, double* state, mh_world_aux* aux)
{
  if (!bb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(bb);
  const size_t B = (size_t)bb->B;
  MH_HIP(hipDeviceSynchronize());
  if (state) MH_HIP(hipMemcpy(state, bb->d.state, B * bb->nb * 13 * 8, hipMemcpyDeviceToHost));
  if (aux) {
    std::vector<uint32_t> rng(B * MH_RAND_WORDS); std::vector<double> tm(B); std::vector<int> st(B), zl(B), zs(B), zc(B);
    std::vector<unsigned long long> steps(B), minis(B), stabs(B), cnt(5 * B);
    MH_HIP(hipMemcpy(rng.data(), bb->d_rng, rng.size() * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(tm.data(), bb->d.time, B * 8, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(st.data(), bb->d.status, B * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(steps.data(), bb->d.steps, B * 8, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(minis.data(), bb->d.mini_steps, B * 8, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(stabs.data(), bb->d.stab_iters, B * 8, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(cnt.data(), bb->core.cnt, B * 40, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(zl.data(), bb->core.zlast_size, B * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(zs.data(), bb->core.zbuf_size, B * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(zc.data(), bb->core.zbuf_cap, B * 4, hipMemcpyDeviceToHost));
    std::vector<int> vsz(B); std::vector<double> vns(B * MH_NOSLIP_MAX);
    MH_HIP(hipMemcpy(vsz.data(), bb->core.vns_size, B * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(vns.data(), bb->core.vns, B * MH_NOSLIP_MAX * 8, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < B; b++) {
      std::memset(&aux[b], 0, sizeof(mh_world_aux));
      std::memcpy(aux[b].rng, &rng[b * MH_RAND_WORDS], MH_RAND_WORDS * 4);
      aux[b].time = tm[b]; aux[b].status = st[b]; aux[b].steps = steps[b]; aux[b].mini_steps = minis[b]; aux[b].stab_iters = stabs[b];
      aux[b].lcp_solves = cnt[5*b]; aux[b].lcp_rows = cnt[5*b+1]; aux[b].lcp_pivots = cnt[5*b+2]; aux[b].lcp_alg_bytes = cnt[5*b+3]; aux[b].stab_rows = cnt[5*b+4];
      aux[b].zlast_size = zl[b]; aux[b].zbuf_size = zs[b]; aux[b].zbuf_cap = zc[b];      // the vectors themselves: save_solver_state
      aux[b].vns_size = vsz[b]; std::memcpy(aux[b].vns, &vns[b * MH_NOSLIP_MAX], MH_NOSLIP_MAX * 8);   // _v of the no-slip model
    }
  }
  return MH_OK;
}

int mh_big_batch_step(mh_big_batch* bb, void* stream, double dt, int nsteps)
{
  namespace bg = mh::big;
  if (!bb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(bb);
  if (nsteps < 0) return fail(MH_ERR_INVALID_ARG, "negative step count");
  if (nsteps > 0 && !(dt > 0.0)) return fail(MH_ERR_INVALID_ARG, "dt must be > 0");
  hipStream_t s = (hipStream_t)stream;
  const int B = bb->B;
  for (int k = 0; k < nsteps; k++) {
    hipLaunchKernelGGL(bg::k_step_begin, dim3((B + 63) / 64), dim3(64), 0, s, bb->d);
    int any = 1;
    while (any) {                                           // step_si_Euler: while (h < dt) h += do_mini_step(dt - h)
      MH_HIP(hipMemsetAsync(bb->d.anyflag, 0, sizeof(int), s));
      hipLaunchKernelGGL(bg::k_mini_pre, dim3(B), dim3(bg::T), 0, s, bb->d, dt);
      if (bb->d.kk_nisl > 0) hipLaunchKernelGGL(bg::k_kkt_fwd, dim3(bb->d.kk_nisl, B), dim3(bg::KKT_T), 0, s, bb->d);
      MH_HIP(hipGetLastError());
      int rc = mh_imp_core_process(&bb->core, s, MH_CORE_IMPACT);
      if (rc != MH_OK) return rc;
      hipLaunchKernelGGL(bg::k_mini_post, dim3((B + 63) / 64), dim3(64), 0, s, bb->d, dt);
      MH_HIP(hipGetLastError());
      rc = read_flag(bb, s, &any);
      if (rc != MH_OK) return rc;
    }
    int rc = run_stabilize(bb, s, 1);
    if (rc != MH_OK) return rc;
  }
  return MH_OK;
}

int mh_big_batch_stabilize(mh_big_batch* bb, void* stream)
{
  namespace bg = mh::big;
  if (!bb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(bb);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bg::k_pairs_now, dim3(bb->B), dim3(bg::T), 0, s, bb->d);
  MH_HIP(hipGetLastError());
  return run_stabilize(bb, s, 0);
}

int mh_big_batch_save_solver_state(mh_big_batch* bb, double* zlast, double* zbuf, int* sizes3)
{
  if (!bb || !zlast || !zbuf || !sizes3) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(bb);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)bb->B, n = (size_t)bb->cap;
  std::vector<int> a(B), b2(B), c(B);
  MH_HIP(hipMemcpy(zlast, bb->core.zlast, B * n * 8, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(zbuf, bb->core.zbuf, B * n * 8, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(a.data(), bb->core.zlast_size, B * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(b2.data(), bb->core.zbuf_size, B * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(c.data(), bb->core.zbuf_cap, B * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < B; i++) { sizes3[3*i] = a[i]; sizes3[3*i+1] = b2[i]; sizes3[3*i+2] = c[i]; }
  return MH_OK;
}

int mh_big_batch_load_solver_state(mh_big_batch* bb, const double* zlast, const double* zbuf, const int* sizes3)
{
  if (!bb || !zlast || !zbuf || !sizes3) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(bb);
  const size_t B = (size_t)bb->B, n = (size_t)bb->cap;
  std::vector<int> a(B), b2(B), c(B);
  for (size_t i = 0; i < B; i++) {
    a[i] = sizes3[3*i]; b2[i] = sizes3[3*i+1]; c[i] = sizes3[3*i+2];
    if (a[i] < 0 || a[i] > (int)n || b2[i] < 0 || c[i] < 0 || c[i] > (int)n) return fail(MH_ERR_INVALID_ARG, "world %zu: vector sizes outside the capacity %zu", i, n);
  }
  MH_HIP(hipDeviceSynchronize());
  MH_HIP(hipMemcpy(bb->core.zlast, zlast, B * n * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(bb->core.zbuf, zbuf, B * n * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(bb->core.zlast_size, a.data(), B * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(bb->core.zbuf_size, b2.data(), B * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(bb->core.zbuf_cap, c.data(), B * 4, hipMemcpyHostToDevice));
  return MH_OK;
}

} // extern "C"
