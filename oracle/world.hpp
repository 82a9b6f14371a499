// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of one Moby world: TimeSteppingSimulator::step and
// everything it drives, for scenes of free rigid SPHERES plus one static
// PLANE (BASELINE configs 1 and 2).  Each function cites the reference code it
// follows (/root/reference/...).  Arithmetic that lives in Ravelin (poses,
// spatial transforms, rigid-body dynamics, quaternion maps, dense products) is
// NOT in the tree: it is restated here from first principles with
// an explicit operation order, which is the order the HIP kernels reproduce
// bit for bit.  Those parts are "parity unpinned" against a reference binary
// and pinned only through regress/sphere-stack.dat (6 digits) and physics
// property tests.
//
// Canonical orders (the reference's are heap-address dependent):
//   bodies by id 0..nb-1, ground = nb; pairs (i<j) lexicographic; geometry of
//   the lower id is "A" of a pair; islands start from the lowest body id.
//
// Reference quirks reproduced on purpose:
//   * swept bounding volumes and calc_max_dist use the velocity of the body
//     point at the GLOBAL ORIGIN (Pose3d::transform(GLOBAL, v) of a spatial
//     velocity: BoundingSphere.cpp:88, CCD.cpp:597);
//   * positions are integrated with the OLD velocity (TSS:156-164);
//   * lcp_fast warm-starts from whatever is in _z when sizes differ
//     (ICH-QP:158-162 + LCP.cpp:65);
//   * stabilisation measures progress on the SIMULATOR's pair list but builds
//     its LCP from its own broad phase with dt = 0 (CStab:97,364);
//   * restitution re-applies z on top of the compression impulse (ICH:578-602).
#ifndef ORACLE_WORLD_HPP
#define ORACLE_WORLD_HPP
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../include/moby_hip.h"
#include "../include/moby_hip_impact.h"
#include "../include/moby_hip_stack.h"
#include "lcp.hpp"

namespace oracle {

struct V3 { double x, y, z; };
static inline V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
static inline double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static const double NEAR_ZERO = 1.4901161193847656e-08;  // Constants.h:21
static const double INF = DBL_MAX;

struct Contact {            // UnilateralConstraint (eContact)
  int g1, g2;               // contact_geom1 / contact_geom2 body ids (nb = ground)
  int pair;                 // pair index (contact parameters)
  V3 p, n, s, t;            // contact_point, normal (from g2 toward g1), tangents
  double dist;              // signed_violation
  double mu, muv, eps, compliance; int nk;
  mutable double imp[3] = {0.0, 0.0, 0.0};   // accumulated (cn, cs, ct): what update_from_stacked adds to contact_impulse (ICH:298-410)
};

struct PairDist { int pair, a, b; double dist; V3 pa, pb; };  // PairwiseDistInfo (global points)

static int g_impact_model = MH_IMPACT_MODEL_DS;   // what the next World is built with: the reference's USE_AP build option
static inline int oracle_impact_model() { return g_impact_model; }
static const double BILATERAL_EPS = 1e-6;          // ConstraintStabilization::bilateral_eps (CStab:62)
static FILE* g_lcp_dump = nullptr;         // diagnostic (oracle_dbg_lcp_dump): every impact LCP of solve_impact_lcp with its inputs and pivot counts
static unsigned long long g_ca_iters = 0;   // diagnostic: conservative-advancement sub-steps taken

// What the stepper reads of a scene: the members of mh_scene under the same names, as pointers, so that scenes of any
// size (include/moby_hip_stack.h: mh_big_scene, explicit candidate-pair list) run through the same code.  Pair p of a
// small scene is the lexicographic (i<j) index over nb (+ ground) bodies, as before.
struct SceneView {
  int nb, has_ground;
  const int* geom_type; const double (*geom_dim)[3]; const double* mass; const double (*inertia)[3];
  const double* plane_R; const double* plane_o; const double* gravity;
  int npairs; const int* pair_a; const int* pair_b; const int* pair_model;     // pair_a == nullptr: all (i<j), lexicographic
  const int* pair_enabled; const double* cp_epsilon; const double* cp_mu_coulomb; const double* cp_mu_viscous;
  const double* cp_compliance; const int* cp_nk;
  double min_step_size, contact_dist_thresh, cstab_eps; unsigned cstab_max_iterations;
  // implicit joints (mh_big_scene's fields, under the same names); absent in small scenes
  int njoints; const int* joint_type; const int* joint_inboard; const int* joint_outboard;
  const double (*joint_anchor_in)[3]; const double (*joint_anchor_out)[3]; const double (*joint_vec_in)[9]; const double (*joint_vec_out)[9];
};

class World {
 public:
  SceneView view_;
  const SceneView* sc;
  double* st;               // nb * 13
  mh_world_aux* aux;
  int32_t* trace = nullptr; int trace_cap = 0; int trace_len = 0;  // concatenated LCP traces (tests)
  int impact_model = oracle_impact_model();                        // the reference's build option USE_AP (CMakeLists.txt:19)

  // body table and handler storage: the scene's / the aux record's by default; the impact-handler entry
  // (oracle_impact_process) and the big-scene stepper point them at caller arrays of any size instead
  int nb_; const double* mass_; const double (*inertia_)[3];
  double* zlast_; double* zbuf_; int lcp_cap_;

  World(const mh_scene* s, double* state, mh_world_aux* a) : sc(&view_), st(state), aux(a),
    nb_(s->nb), mass_(s->mass), inertia_(s->inertia), zlast_(a->zlast), zbuf_(a->zbuf), lcp_cap_(MH_LCP_MAX_N_WAVE) {
    const int ntot = s->nb + (s->has_ground ? 1 : 0);
    view_ = SceneView{ s->nb, s->has_ground, s->geom_type, s->geom_dim, s->mass, s->inertia, s->plane_R, s->plane_o, s->gravity,
                       ntot * (ntot - 1) / 2, nullptr, nullptr, nullptr,
                       s->pair_enabled, s->cp_epsilon, s->cp_mu_coulomb, s->cp_mu_viscous, s->cp_compliance, s->cp_nk,
                       s->min_step_size, s->contact_dist_thresh, s->cstab_eps, s->cstab_max_iterations,
                       0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  }
  World(int nb, const double* mass, const double (*inertia)[3], double* state, mh_world_aux* a, double* zlast, double* zbuf, int lcp_cap)
    : sc(nullptr), st(state), aux(a), nb_(nb), mass_(mass), inertia_(inertia), zlast_(zlast), zbuf_(zbuf), lcp_cap_(lcp_cap) {}
  // a big scene: the view is the caller's, the handler vectors _zlast / _z are caller arrays of lcp_cap doubles
  World(const SceneView& v, double* state, mh_world_aux* a, double* zlast, double* zbuf, int lcp_cap)
    : view_(v), sc(&view_), st(state), aux(a), nb_(v.nb), mass_(v.mass), inertia_(v.inertia), zlast_(zlast), zbuf_(zbuf), lcp_cap_(lcp_cap) {}

  // ---- state access -------------------------------------------------------
  V3 X(int b) const { return v3(st[13*b], st[13*b+1], st[13*b+2]); }
  V3 Vl(int b) const { return v3(st[13*b+7], st[13*b+8], st[13*b+9]); }
  V3 Wa(int b) const { return v3(st[13*b+10], st[13*b+11], st[13*b+12]); }
  void setX(int b, V3 v) { st[13*b] = v.x; st[13*b+1] = v.y; st[13*b+2] = v.z; }
  void setV(int b, V3 v) { st[13*b+7] = v.x; st[13*b+8] = v.y; st[13*b+9] = v.z; }
  void setW(int b, V3 v) { st[13*b+10] = v.x; st[13*b+11] = v.y; st[13*b+12] = v.z; }
  bool enabled(int b) const { return b >= 0 && b < nb_; }
  int nbodies_all() const { return sc->nb + (sc->has_ground ? 1 : 0); }
  static int pair_index(int i, int j, int ntot) { // i<j, lexicographic
    return i * ntot - (i * (i + 1)) / 2 + (j - i - 1);
  }
  // rotation matrix of body b from its quaternion (x,y,z,w)
  void rot(int b, double R[9]) const {
    const double x = st[13*b+3], y = st[13*b+4], z = st[13*b+5], w = st[13*b+6];
    R[0] = 1.0 - 2.0 * (y*y + z*z); R[1] = 2.0 * (x*y - z*w);       R[2] = 2.0 * (x*z + y*w);
    R[3] = 2.0 * (x*y + z*w);       R[4] = 1.0 - 2.0 * (x*x + z*z); R[5] = 2.0 * (y*z - x*w);
    R[6] = 2.0 * (x*z - y*w);       R[7] = 2.0 * (y*z + x*w);       R[8] = 1.0 - 2.0 * (x*x + y*y);
  }
  // velocity of the body point at world point p (spatial velocity transformed to a frame at p)
  V3 point_vel(int b, V3 p) const {
    if (!enabled(b)) return v3(0, 0, 0);
    return Vl(b) + cross(Wa(b), p - X(b));
  }
  V3 plane_n() const { return v3(sc->plane_R[1], sc->plane_R[4], sc->plane_R[7]); }
  V3 to_plane(V3 p) const { // R^T (p - o)
    const double* R = sc->plane_R; V3 d = p - v3(sc->plane_o[0], sc->plane_o[1], sc->plane_o[2]);
    return v3((R[0]*d.x + R[3]*d.y) + R[6]*d.z, (R[1]*d.x + R[4]*d.y) + R[7]*d.z, (R[2]*d.x + R[5]*d.y) + R[8]*d.z);
  }
  V3 from_plane(V3 p) const { // o + R p
    const double* R = sc->plane_R;
    return v3(sc->plane_o[0] + ((R[0]*p.x + R[1]*p.y) + R[2]*p.z),
              sc->plane_o[1] + ((R[3]*p.x + R[4]*p.y) + R[5]*p.z),
              sc->plane_o[2] + ((R[6]*p.x + R[7]*p.y) + R[8]*p.z));
  }

  // ---- collision detection --------------------------------------------------
  // CCD::broad_phase (CCD.cpp:702-876) with SSL bounds (SSL.cpp:550-579) of the
  // swept bounding sphere (BoundingSphere.cpp:71-95); the plane has an infinite
  // DummyBV (DummyBV.h:53-56).
  void broad_phase(double dt, std::vector<int>& pairs) const {
    const int ntot = nbodies_all();
    std::vector<double> lo(3 * (size_t)ntot), hi(3 * (size_t)ntot);
    for (int b = 0; b < ntot; b++) {
      if (!enabled(b)) { for (int k = 0; k < 3; k++) { lo[3*b+k] = -INF; hi[3*b+k] = INF; } continue; }
      const V3 c = X(b);
      const V3 vdt = Vl(b) * dt, wdt = Wa(b) * dt;
      const V3 lin = vdt + cross(c, wdt);            // linear part of (v dt) expressed at the global origin
      const V3 p2 = c + lin;
      const double r = bounding_radius(b);
      for (int k = 0; k < 3; k++) {
        const double a = comp(c, k), e = comp(p2, k);
        lo[3*b+k] = ((a < e) ? a : e) - r;
        hi[3*b+k] = ((a > e) ? a : e) + r;
      }
    }
    pairs.clear();
    for (int p = 0; p < sc->npairs; p++) {
        int i, j; pair_bodies(p, i, j);
        if (is_spokes(i) || is_spokes(j)) continue;   // removed from CCD's body list (coldet-plugin.cpp:58-66)
        if (is_pin(i) || is_pin(j)) {                 // PendulumColdetPlugin::broad_phase: exactly the pair (world, l1), always (:53-58)
          if (is_pin(i) && j == sc->nb && sc->pair_enabled[p]) pairs.push_back(p);
          continue;
        }
        if ((is_box(i) || is_box(j)) && j != sc->nb && !vertex_face(p)) {  // box-box (v-clip on a qhull polyhedron) / box-sphere: not built
          if (sc->pair_enabled[p]) aux->status |= MH_WORLD_UNSUPPORTED;
          continue;
        }
        bool ov = true;
        for (int k = 0; k < 3; k++) if (!(lo[3*i+k] <= hi[3*j+k] && lo[3*j+k] <= hi[3*i+k])) ov = false;
        if (!ov) continue;                          // needs overlap on all three axes (CCD.cpp:857)
        if (!sc->pair_enabled[p]) continue;
        if (!enabled(i) && !enabled(j)) continue;
        pairs.push_back(p);
      }
    // BladePlanePlugin::broad_phase appends (ground, wheel) unconditionally (coldet-plugin.cpp:72)
    if (sc->has_ground && !sc->pair_a)
      for (int i = 0; i < sc->nb; i++) if (is_spokes(i)) pairs.push_back(pair_index(i, sc->nb, ntot));
  }
  // MH_PAIR_VERTEX_FACE (include/moby_hip_stack.h): a box-box pair handled like box-plane, the plane being the +Y face of
  // the lower-id box ("support") and the polyhedron the higher-id box -- the build's documented model of a stacked pair
  bool vertex_face(int p) const { return sc->pair_model && sc->pair_model[p] == MH_PAIR_VERTEX_FACE; }
  bool is_spokes(int b) const { return b < sc->nb && sc->geom_type[b] == MH_GEOM_SPOKES; }
  bool is_box(int b) const { return b < sc->nb && sc->geom_type[b] == MH_GEOM_BOX; }
  bool is_pin(int b) const { return b >= 0 && b < sc->nb && sc->geom_type[b] == MH_GEOM_PIN; }
  // PendulumColdetPlugin: the body point pl1 = geom_dim (l1 frame) in the global frame
  V3 pin_point(int b) const {
    const double px = sc->geom_dim[b][0], py = sc->geom_dim[b][1], pz = sc->geom_dim[b][2];
    double R[9]; rot(b, R);
    const V3 c = X(b);
    return v3(c.x + ((R[0]*px + R[1]*py) + R[2]*pz), c.y + ((R[3]*px + R[4]*py) + R[5]*pz), c.z + ((R[6]*px + R[7]*py) + R[8]*pz));
  }
  // radius of the bounding sphere CCD::construct_bounding_sphere builds (CCD.cpp:1040-1063)
  double bounding_radius(int b) const {
    if (is_box(b)) { const V3 h = v3(sc->geom_dim[b][0] / 2.0, sc->geom_dim[b][1] / 2.0, sc->geom_dim[b][2] / 2.0); return norm(h); }
    if (is_pin(b)) return 0.0;                      // no primitive: the plugin's broad phase never looks at bounds
    return sc->geom_dim[b][0];
  }
  // BoxPrimitive::get_vertices order (BoxPrimitive.cpp:358-365), global frame
  V3 box_vertex(int b, int i) const {
    const double hx = sc->geom_dim[b][0] * 0.5, hy = sc->geom_dim[b][1] * 0.5, hz = sc->geom_dim[b][2] * 0.5;
    const double px = (i & 4) ? -hx : hx, py = (i & 2) ? -hy : hy, pz = (i & 1) ? -hz : hz;
    double R[9]; rot(b, R);
    const V3 c = X(b);
    return v3(c.x + ((R[0]*px + R[1]*py) + R[2]*pz), c.y + ((R[3]*px + R[4]*py) + R[5]*pz), c.z + ((R[6]*px + R[7]*py) + R[8]*pz));
  }
  // tip of spoke i in the global frame: p1 = (cos(theta) R, W/2, sin(theta) R) in the wheel frame,
  // theta = pi i 2 / N, W = 0 (coldet-plugin.cpp:104-113, params.h)
  V3 spoke_tip(int b, int i) const {
    const double Rr = sc->geom_dim[b][0]; const int N = (int)sc->geom_dim[b][1];
    const double theta = M_PI * i * 2.0 / N;
    const double px = std::cos(theta) * Rr, py = 0.0, pz = std::sin(theta) * Rr;
    double R[9]; rot(b, R);
    const V3 c = X(b);
    return v3(c.x + ((R[0]*px + R[1]*py) + R[2]*pz), c.y + ((R[3]*px + R[4]*py) + R[5]*pz), c.z + ((R[6]*px + R[7]*py) + R[8]*pz));
  }
  void pair_bodies(int p, int& a, int& b) const {
    if (sc->pair_a) { a = sc->pair_a[p]; b = sc->pair_b[p]; return; }
    const int ntot = nbodies_all();
    for (int i = 0; i < ntot; i++) for (int j = i + 1; j < ntot; j++) if (pair_index(i, j, ntot) == p) { a = i; b = j; return; }
    a = b = -1;
  }
  // the support plane of a vertex-face pair: through the centre of body L's +Y face, normal = L's +Y axis (world)
  V3 face_n(int L) const { double R[9]; rot(L, R); return v3(R[1], R[4], R[7]); }
  double face_height(int L, V3 p) const { return dot(face_n(L), p - X(L)) - sc->geom_dim[L][1] * 0.5; }
  // CollisionGeometry::calc_signed_dist (CollisionGeometry.cpp:236-250) ->
  // SpherePrimitive.cpp:104-136 / PlanePrimitive.cpp:385-411
  PairDist signed_dist(int p) const {
    PairDist d; d.pair = p; pair_bodies(p, d.a, d.b);
    if (is_spokes(d.a)) {
      // BladePlanePlugin::calc_signed_dist_wheel_plane (coldet-plugin.cpp:88-137).  The pair is
      // (ground, wheel) and the plugin hands back pA = the WHEEL point, pB = the ground point
      // (:336-339 swap the geometries, not the points): kept, it decides the sign in the
      // conservative-advancement step and the normal of the stabilisation contact.
      const int w = d.a;
      const int N = (int)sc->geom_dim[w][1];
      double min_dist = INF;
      for (int i = 0; i < N; i++) {
        const V3 g = spoke_tip(w, i);
        const V3 pp = to_plane(g);
        if (pp.y < min_dist) { min_dist = pp.y; d.pb = from_plane(v3(pp.x, 0.0, pp.z)); d.pa = g; }
      }
      d.dist = min_dist; d.a = sc->nb; d.b = w;
      return d;
    }
    if (is_pin(d.a)) {
      // PendulumColdetPlugin::calc_signed_dist_l1_world (plugin :65-82): -|p|, p = the body point in the world body's frame
      // (identity).  The pair is (world, l1) and the plugin is called with the geometries swapped (:130-136): "pA" receives the
      // l1 point, "pB" the world origin -- only ever read by the stabiliser's separated case, which -|p| never reaches
      const V3 g = pin_point(d.a);
      d.dist = -norm(g - v3(0.0, 0.0, 0.0));
      d.pa = g; d.pb = v3(0.0, 0.0, 0.0);
      return d;
    }
    if (vertex_face(p)) {
      // PlanePrimitive::calc_signed_dist(polyhedral) (PlanePrimitive.cpp:338-376) with the support face as the plane:
      // lowest vertex of the upper box over the face, first one wins ties; pa on the support, pb the vertex
      double min_dist = INF;
      const V3 nL = face_n(d.a);
      for (int i = 0; i < 8; i++) {
        const V3 g = box_vertex(d.b, i);
        const double h = face_height(d.a, g);
        if (h < min_dist) { min_dist = h; d.pb = g; d.pa = g - nL * h; }
      }
      d.dist = min_dist;
      return d;
    }
    if (is_box(d.a)) {
      // BoxPrimitive::calc_signed_dist -> PlanePrimitive::calc_signed_dist(polyhedral) (BoxPrimitive.cpp:156-162,
      // PlanePrimitive.cpp:338-376): lowest vertex in the plane frame, first one wins ties
      double min_dist = INF;
      for (int i = 0; i < 8; i++) {
        const V3 g = box_vertex(d.a, i);
        const V3 pp = to_plane(g);
        if (pp.y < min_dist) { min_dist = pp.y; d.pa = g; d.pb = from_plane(v3(pp.x, 0.0, pp.z)); }
      }
      d.dist = min_dist;
      return d;
    }
    if (enabled(d.a) && enabled(d.b)) {
      const V3 ca = X(d.a), cb = X(d.b);
      const double ra = sc->geom_dim[d.a][0], rb = sc->geom_dim[d.b][0];
      const V3 ab = cb - ca;
      const double len = norm(ab);
      d.dist = len - ra - rb;
      const V3 u = ab / len;
      const double sa = (d.dist > 0.0) ? ra : ra + d.dist, sb = (d.dist > 0.0) ? rb : rb + d.dist;
      d.pa = ca + u * sa;
      d.pb = cb - u * sb;
    } else {
      const int s = enabled(d.a) ? d.a : d.b;       // the sphere
      const V3 cp = to_plane(X(s));
      const double r = sc->geom_dim[s][0];
      const double low = cp.y + (-1.0 * r);
      d.dist = low;
      const V3 on_plane = from_plane(v3(cp.x, 0.0, cp.z));
      const V3 on_sphere = from_plane(v3(cp.x, low, cp.z));
      if (s == d.a) { d.pa = on_sphere; d.pb = on_plane; } else { d.pa = on_plane; d.pb = on_sphere; }
    }
    return d;
  }
  void calc_pairwise_distances(const std::vector<int>& pairs, std::vector<PairDist>& out) const {
    out.clear();
    for (int p : pairs) out.push_back(signed_dist(p));
  }
  // Vector3d::determine_orthonormal_basis (Ravelin; pinned choice: cross with the
  // axis of the smallest |component|)
  static void orthonormal_basis(V3 n, V3& s, V3& t) {
    const double ax = std::fabs(n.x), ay = std::fabs(n.y), az = std::fabs(n.z);
    V3 e;
    if (ax <= ay && ax <= az) e = v3(1, 0, 0); else if (ay <= az) e = v3(0, 1, 0); else e = v3(0, 0, 1);
    s = cross(n, e); s = s / norm(s);
    t = cross(n, s);
  }
  void fill_params(Contact& c) const {             // ConstraintSimulator::preprocess_constraint (CSim:390-416)
    c.mu = sc->cp_mu_coulomb[c.pair]; c.muv = sc->cp_mu_viscous[c.pair];
    c.eps = sc->cp_epsilon[c.pair]; c.compliance = sc->cp_compliance[c.pair]; c.nk = sc->cp_nk[c.pair];
  }
  // CCD::find_contacts (CCD.inl:3-83) -> sphere/sphere (CCD.inl:1164-1207),
  // sphere/plane (CCD.inl:804-847); create_contact (CollisionDetection.cpp:57-95)
  void find_contacts(int p, double TOL, std::vector<Contact>& out) const {
    int a, b; pair_bodies(p, a, b);
    Contact c; c.pair = p;
    if (is_spokes(a)) {
      // BladePlanePlugin::find_contacts_wheel_plane (coldet-plugin.cpp:211-288): one contact per
      // spoke tip below sim->contact_dist_thresh (the TOL argument is ignored, :214)
      const int N = (int)sc->geom_dim[a][1];
      for (int i = 0; i < N; i++) {
        const V3 g = spoke_tip(a, i);
        const V3 pp = to_plane(g);
        if (!(pp.y < sc->contact_dist_thresh)) continue;
        c.p = (g + from_plane(v3(pp.x, 0.0, pp.z))) * 0.5;
        c.n = plane_n(); c.g1 = a; c.g2 = b; c.dist = pp.y;
        orthonormal_basis(c.n, c.s, c.t);
        fill_params(c);
        out.push_back(c);
      }
      return;
    }
    if (is_pin(a)) {
      // PendulumColdetPlugin::find_contacts_l1_world (plugin :84-110): six contacts at the midpoint between the body point and
      // the origin, geom1 = l1, geom2 = world, normals +y -y +z -z +x -x, violation min(0, -p[axis]); TOL is ignored
      const V3 g = pin_point(a);
      const V3 pt = (g + v3(0.0, 0.0, 0.0)) * 0.5;
      const V3 ns[6] = { v3(0, 1, 0), v3(0, -1, 0), v3(0, 0, 1), v3(0, 0, -1), v3(1, 0, 0), v3(-1, 0, 0) };
      const double pv[6] = { g.y, g.y, g.z, g.z, g.x, g.x };
      for (int i = 0; i < 6; i++) {
        c.p = pt; c.n = ns[i]; c.g1 = a; c.g2 = b; c.dist = (-pv[i] < 0.0) ? -pv[i] : 0.0;   // std::min(0.0, -p[axis])
        orthonormal_basis(c.n, c.s, c.t);
        fill_params(c);
        out.push_back(c);
      }
      return;
    }
    if (vertex_face(p)) {
      // find_contacts_plane_generic (CCD.inl:848-886) with the support face as the plane: every vertex of the upper
      // box within TOL (<=) of it; contact point = the vertex, geom1 = support, geom2 = upper box, normal = -(face normal)
      const V3 nL = face_n(a);
      for (int i = 0; i < 8; i++) {
        const V3 g = box_vertex(b, i);
        const double h = face_height(a, g);
        if (!(h <= TOL)) continue;
        c.p = g; c.n = -nL; c.g1 = a; c.g2 = b; c.dist = h;
        orthonormal_basis(c.n, c.s, c.t);
        fill_params(c);
        out.push_back(c);
      }
      return;
    }
    if (is_box(a)) {
      // CCD::find_contacts_plane_generic(plane, box) (CCD.inl:848-886): every vertex within TOL (<=) of
      // the plane; contact point = the vertex, geom1 = plane, geom2 = box, normal = -(plane normal)
      for (int i = 0; i < 8; i++) {
        const V3 g = box_vertex(a, i);
        const V3 pp = to_plane(g);
        if (!(pp.y <= TOL)) continue;
        c.p = g; c.n = -plane_n(); c.g1 = b; c.g2 = a; c.dist = pp.y;
        orthonormal_basis(c.n, c.s, c.t);
        fill_params(c);
        out.push_back(c);
      }
      return;
    }
    if (enabled(a) && enabled(b)) {
      const V3 cA = X(a), cB = X(b);
      const double rA = sc->geom_dim[a][0], rB = sc->geom_dim[b][0];
      const V3 d = cA - cB;
      const double len = norm(d);
      const double dist = len - rA - rB;
      if (dist > TOL) return;
      const V3 n = d / len;
      const V3 closest_A = cA - n * rA, closest_B = cB + n * rB;
      c.p = (closest_A + closest_B) * 0.5;
      c.n = n; c.g1 = a; c.g2 = b; c.dist = dist;
    } else {
      const int s = enabled(a) ? a : b, pl = enabled(a) ? b : a;
      const V3 cp = to_plane(X(s));
      const double r = sc->geom_dim[s][0];
      const double dist = cp.y - r;
      if (dist > TOL) return;
      c.p = from_plane(v3(cp.x, 0.5 * (cp.y - r), cp.z));
      c.n = plane_n(); c.g1 = s; c.g2 = pl; c.dist = dist;
    }
    orthonormal_basis(c.n, c.s, c.t);
    fill_params(c);
    out.push_back(c);
  }
  // UnilateralConstraint::calc_contact_vel / calc_constraint_vel (UC:695-747,1357-1384)
  double contact_vel(const Contact& c, V3 dir) const {
    return dot(dir, point_vel(c.g1, c.p) - point_vel(c.g2, c.p));
  }
  // CCD::calc_max_dist (CCD.cpp:585-609): velocity at the global origin
  double calc_max_dist(int b, V3 n, double rmax) const {
    if (!enabled(b)) return 0.0;
    const V3 xd0 = Vl(b) + cross(X(b), Wa(b));
    const V3 w0 = Wa(b);
    return dot(n, xd0) + norm(cross(w0, n)) * rmax;
  }
  // CCD::_rmax (CCD.cpp:739): sphere radius; a spokes body is never seen by CCD::broad_phase, so
  // the std::map lookup default-constructs 0
  // (box: BoxPrimitive::get_bounding_radius, BoxPrimitive.h:44 -- the FULL diagonal)
  double rmax_of(int b) const {
    if (!enabled(b) || is_spokes(b)) return 0.0;
    if (is_box(b)) { const double x = sc->geom_dim[b][0], y = sc->geom_dim[b][1], z = sc->geom_dim[b][2]; return std::sqrt((x*x + y*y) + z*z); }
    return sc->geom_dim[b][0];
  }
  // CompGeom::collinear / rel_equal (CompGeom.cpp:1923-1931, CompGeom.h:110)
  static bool rel_equal(double x, double y) { const double m = std::max(std::fabs(x), std::max(std::fabs(y), 1.0)); return std::fabs(x - y) <= NEAR_ZERO * m; }
  static bool collinear(V3 a, V3 b, V3 c) {
    return rel_equal((c.z-a.z)*(b.y-a.y), (b.z-a.z)*(c.y-a.y)) && rel_equal((b.z-a.z)*(c.x-a.x), (b.x-a.x)*(c.z-a.z)) &&
           rel_equal((b.x-a.x)*(c.y-a.y), (b.y-a.y)*(c.x-a.x));
  }
  // CCD::calc_next_CA_Euler_step_polyhedron_plane (CCD.cpp:410-468) for box `bx` resting on the
  // plane: called with normal = -contact_normal = +plane normal, offset0 = -<contact normal, point>
  double next_CA_box_plane(int bx, V3 normal, double offset0, int support = -1) const {
    double R[9]; rot(bx, R);
    auto to_box_vec = [&](V3 v) { return v3((R[0]*v.x + R[3]*v.y) + R[6]*v.z, (R[1]*v.x + R[4]*v.y) + R[7]*v.z, (R[2]*v.x + R[5]*v.y) + R[8]*v.z); };
    const V3 nP = to_box_vec(normal);
    const V3 p0 = normal * offset0;
    const double offset = dot(nP, to_box_vec(p0 - X(bx)));
    // rv = velocity of the polyhedron relative to the plane's body, at the polyhedron's pose (CCD.cpp:388-394); the
    // ground plane does not move, a support box does
    const V3 wrel = enabled(support) ? Wa(bx) - Wa(support) : Wa(bx);
    const V3 vrel = enabled(support) ? Vl(bx) - point_vel(support, X(bx)) : Vl(bx);
    const double av_norm = norm(to_box_vec(wrel));
    const double lv_dot_n = -dot(nP, to_box_vec(vrel));
    const double hx = sc->geom_dim[bx][0] * 0.5, hy = sc->geom_dim[bx][1] * 0.5, hz = sc->geom_dim[bx][2] * 0.5;
    double max_step = INF;
    for (int i = 0; i < 8; i++) {
      const V3 vtx = v3((i & 4) ? -hx : hx, (i & 2) ? -hy : hy, (i & 1) ? -hz : hz);
      const double r = norm(vtx);
      const double dist = dot(nP, vtx) - offset;
      if (dist < NEAR_ZERO) continue;
      const double sp = lv_dot_n + av_norm * r;
      const double speed = (0.0 > sp) ? 0.0 : sp;
      const double cand = dist / speed;
      max_step = (cand < max_step) ? cand : max_step;
    }
    return max_step;
  }
  // CCD::calc_next_CA_Euler_step_generic (CCD.cpp:238-405) for sphere pairs
  double next_CA_generic(const PairDist& d) const {
    if (is_spokes(d.b)) return INF;                 // BladePlanePlugin::calc_next_CA_Euler_step (coldet-plugin.cpp:205-208)
    if (is_pin(d.a)) return INF;                    // PendulumColdetPlugin::calc_next_CA_Euler_step (plugin :139-142)
    std::vector<Contact> cs; find_contacts(d.pair, NEAR_ZERO, cs);
    if (cs.empty()) return INF;
    for (const Contact& c : cs) if (contact_vel(c, c.n) < -NEAR_ZERO) return 0.0;
    if (is_box(d.a)) {
      // CCD.cpp:285-323: three non-collinear contacts (only the FIRST three are ever tested, :313-318) => rest
      if (cs.size() >= 3 && !collinear(cs[0].p, cs[1].p, cs[2].p)) return INF;
      // geom1 is the plane: the "planeA / polyhedron B" branch (CCD.cpp:383-397)
      const Contact& c = cs[0];
      const double dd = dot(c.n, c.p);
      if (vertex_face(d.pair)) return next_CA_box_plane(d.b, -c.n, -dd, d.a);
      return next_CA_box_plane(d.a, -c.n, -dd);
    }
    return INF;
  }
  // CCD::calc_CA_Euler_step_generic (CCD.cpp:169-235)
  double CA_generic(const PairDist& d) const {
    if (d.dist <= 0.0) return next_CA_generic(d);
    const V3 d0 = d.pa - d.pb;
    const V3 n0 = d0 / norm(d0);
    const double tA = calc_max_dist(d.a, -n0, rmax_of(d.a));
    const double tB = calc_max_dist(d.b, n0, rmax_of(d.b));
    double total = tA + tB;
    if (total < 0.0) total = 0.0;
    const double cand = d.dist / total;
    return (cand < INF) ? cand : INF;               // std::min(maxt, dist/total)
  }
  // CCD::calc_CA_Euler_step_sphere (CCD.cpp:138-166)
  double CA_step(const PairDist& d) const {
    if (is_spokes(d.b) || is_box(d.a) || is_pin(d.a)) return CA_generic(d);   // no SpherePrimitive in the pair (CCD.cpp:127-133)
    if (d.dist > NEAR_ZERO) return CA_generic(d);
    std::vector<Contact> cs; find_contacts(d.pair, NEAR_ZERO, cs);
    if (cs.size() == 1 && std::fabs(contact_vel(cs[0], cs[0].n)) < NEAR_ZERO * 10) return INF;
    return CA_generic(d);
  }

  // ---- rigid body dynamics (Ravelin::RigidBodyd, restated) -----------------
  // world-frame inertia R J R^T and its SPD inverse (inverse_SPD, ICH:1607)
  void inertia_world(int b, double Jw[9]) const {
    double R[9]; rot(b, R);
    const double* J = inertia_[b];
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[3*i+j] = R[3*i+j] * J[j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      Jw[3*i+j] = (T[3*i] * R[3*j] + T[3*i+1] * R[3*j+1]) + T[3*i+2] * R[3*j+2];
    // enforce exact symmetry (upper := lower), as a symmetric-storage inertia has
    Jw[1] = Jw[3]; Jw[2] = Jw[6]; Jw[5] = Jw[7];
  }
  // X block of body b: blockdiag(inv(m I3), inv(Jw)), both via Cholesky (inverse_spd)
  void inv_inertia(int b, double& im, double Ji[9]) const {
    double Mm[1] = { mass_[b] };
    inverse_spd(1, Mm, 1); im = Mm[0];
    double Jw[9]; inertia_world(b, Jw);
    double A[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i + 3*j] = Jw[3*i+j];
    inverse_spd(3, A, 3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ji[3*i+j] = A[i + 3*j];
  }
  // DynamicBodyd::calc_fwd_dyn for a free body with gravity (Sim:552; GravityForce.cpp:33-69):
  // xdd = (g m) / m ; wd = Jw^-1 (0 - w x (Jw w))
  void fwd_dyn(int b, V3& xdd, V3& wd) const {
    const double m = sc->mass[b];
    const V3 f = v3(sc->gravity[0] * m, sc->gravity[1] * m, sc->gravity[2] * m);
    xdd = f / m;
    double Jw[9]; inertia_world(b, Jw);
    const V3 w = Wa(b);
    const V3 Jww = v3((Jw[0]*w.x + Jw[1]*w.y) + Jw[2]*w.z, (Jw[3]*w.x + Jw[4]*w.y) + Jw[5]*w.z, (Jw[6]*w.x + Jw[7]*w.y) + Jw[8]*w.z);
    const V3 tau = -cross(w, Jww);
    double im, Ji[9]; inv_inertia(b, im, Ji);
    wd = v3((Ji[0]*tau.x + Ji[1]*tau.y) + Ji[2]*tau.z, (Ji[3]*tau.x + Ji[4]*tau.y) + Ji[5]*tau.z, (Ji[6]*tau.x + Ji[7]*tau.y) + Ji[8]*tau.z);
  }
  // generalized velocity in eEuler form: [xd ; qd], qd = 1/2 (0,w) (x) q
  void euler_vel(int b, double qd[7]) const {
    const V3 v = Vl(b), w = Wa(b);
    const double x = st[13*b+3], y = st[13*b+4], z = st[13*b+5], ww = st[13*b+6];
    qd[0] = v.x; qd[1] = v.y; qd[2] = v.z;
    qd[3] = 0.5 * ((ww * w.x + z * w.y) - y * w.z);
    qd[4] = 0.5 * ((ww * w.y + x * w.z) - z * w.x);
    qd[5] = 0.5 * ((ww * w.z + y * w.x) - x * w.y);
    qd[6] = 0.5 * (((-x * w.x) - y * w.y) - z * w.z);
  }
  void get_coords(int b, double q[7]) const { for (int i = 0; i < 7; i++) q[i] = st[13*b+i]; }
  // set_generalized_coordinates_euler: stores x and the NORMALISED quaternion
  void set_coords(int b, const double q[7]) {
    for (int i = 0; i < 3; i++) st[13*b+i] = q[i];
    const double nrm = std::sqrt(((q[3]*q[3] + q[4]*q[4]) + q[5]*q[5]) + q[6]*q[6]);
    for (int i = 3; i < 7; i++) st[13*b+i] = q[i] / nrm;
  }

  // ---- implicit joints and the KKT forward dynamics (Simulator::solve, Sim:608-805) ---------------------------
  // Joint::num_constraint_eqns / evaluate_constraints / calc_constraint_jacobian are Ravelin's (source not in the tree:
  // parity unpinned).  Restated as the standard forms: 3 position rows C = p_in - p_out along the global axes, Jacobian
  // rows [e_k, r x e_k] (inboard, + ; outboard, -) exactly like a contact row (ICH:1847-1895); orientation rows
  // C = a . b with a fixed in the inboard and b in the outboard frame, Jacobian rows [0, a x b] (+ / -).
  int njoints() const { return sc ? sc->njoints : 0; }
  static int joint_rows(int type) { return (type == MH_IJOINT_SPHERICAL || type == MH_IJOINT_PLANAR) ? 3 : (type == MH_IJOINT_UNIVERSAL ? 4 : ((type == MH_IJOINT_REVOLUTE || type == MH_IJOINT_PRISMATIC) ? 5 : 6)); }
  static int joint_pos_rows(int type) { return type == MH_IJOINT_PLANAR ? 1 : (type == MH_IJOINT_PRISMATIC ? 2 : 3); }
  static int joint_dir_slot(int type, int k) { return type == MH_IJOINT_PLANAR ? 2 : k; }   // which a_k a direction row uses
  V3 body_vec(int b, const double* u) const {           // R u for a dynamic body, u for the static world
    if (!enabled(b)) return v3(u[0], u[1], u[2]);
    double R[9]; rot(b, R);
    return v3((R[0]*u[0] + R[1]*u[1]) + R[2]*u[2], (R[3]*u[0] + R[4]*u[1]) + R[5]*u[2], (R[6]*u[0] + R[7]*u[1]) + R[8]*u[2]);
  }
  void joint_eval(int j, double C[6]) const {
    const int bi = sc->joint_inboard[j], bo = sc->joint_outboard[j];
    const V3 ri = body_vec(bi, sc->joint_anchor_in[j]), ro = body_vec(bo, sc->joint_anchor_out[j]);
    const V3 pi = enabled(bi) ? X(bi) + ri : ri, po = enabled(bo) ? X(bo) + ro : ro;
    const V3 d = pi - po;
    const int np = joint_pos_rows(sc->joint_type[j]);
    if (np != 3) for (int k = 0; k < np; k++) C[k] = dot(body_vec(bi, sc->joint_vec_in[j] + 3 * joint_dir_slot(sc->joint_type[j], k)), d);   // planar / prismatic: along inboard-fixed directions
    else { C[0] = d.x; C[1] = d.y; C[2] = d.z; }
    const int nori = joint_rows(sc->joint_type[j]) - np;
    for (int k = 0; k < nori; k++) C[np + k] = dot(body_vec(bi, sc->joint_vec_in[j] + 3 * k), body_vec(bo, sc->joint_vec_out[j] + 3 * k));
  }
  void joint_jac(int j, bool inboard, double Cq[6][6]) const {
    const int bi = sc->joint_inboard[j], bo = sc->joint_outboard[j];
    const V3 r = inboard ? body_vec(bi, sc->joint_anchor_in[j]) : body_vec(bo, sc->joint_anchor_out[j]);
    const double sg = inboard ? 1.0 : -1.0;
    const int np = joint_pos_rows(sc->joint_type[j]);
    if (np != 3) for (int k = 0; k < np; k++) {
      // C = u . (p_in - p_out), u fixed in the inboard frame: dC/dt = u . (v_pin - v_pout) + (w_in x u) . d
      const V3 u = body_vec(bi, sc->joint_vec_in[j] + 3 * joint_dir_slot(sc->joint_type[j], k));
      const V3 ri = body_vec(bi, sc->joint_anchor_in[j]), ro = body_vec(bo, sc->joint_anchor_out[j]);
      const V3 pi = enabled(bi) ? X(bi) + ri : ri, po = enabled(bo) ? X(bo) + ro : ro;
      const V3 e = u * sg;
      V3 ang = cross(r, e);
      if (inboard) ang = ang + cross(u, pi - po);
      Cq[k][0] = e.x; Cq[k][1] = e.y; Cq[k][2] = e.z; Cq[k][3] = ang.x; Cq[k][4] = ang.y; Cq[k][5] = ang.z;
    } else for (int k = 0; k < 3; k++) {
      const V3 e = v3(k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0);
      const V3 rxe = cross(r, e);
      Cq[k][0] = e.x; Cq[k][1] = e.y; Cq[k][2] = e.z; Cq[k][3] = rxe.x; Cq[k][4] = rxe.y; Cq[k][5] = rxe.z;
    }
    const int nori = joint_rows(sc->joint_type[j]) - np;
    for (int k = 0; k < nori; k++) {
      V3 axb = cross(body_vec(bi, sc->joint_vec_in[j] + 3 * k), body_vec(bo, sc->joint_vec_out[j] + 3 * k));
      if (!inboard) axb = -axb;
      Cq[np + k][0] = 0.0; Cq[np + k][1] = 0.0; Cq[np + k][2] = 0.0; Cq[np + k][3] = axb.x; Cq[np + k][4] = axb.y; Cq[np + k][5] = axb.z;
    }
  }
  // Simulator::find_islands (Sim:956-1045): bodies connected by implicit joints whose two links are both enabled; islands
  // from the lowest body id, then sorted (Sim:501) -- body ids stand for the reference's pointer order
  void find_body_islands(std::vector<std::vector<int> >& islands) const {
    const int nb = nb_;
    std::vector<std::vector<int> > adj(nb);
    for (int j = 0; j < njoints(); j++) {
      const int a = sc->joint_inboard[j], b = sc->joint_outboard[j];
      if (enabled(a) && enabled(b)) { adj[a].push_back(b); adj[b].push_back(a); }
    }
    std::vector<char> seen(nb, 0);
    islands.clear();
    for (int s = 0; s < nb; s++) {
      if (seen[s]) continue;
      std::vector<int> q; q.push_back(s); seen[s] = 1;
      for (size_t qi = 0; qi < q.size(); qi++) for (int nbr : adj[q[qi]]) if (!seen[nbr]) { seen[nbr] = 1; q.push_back(nbr); }
      std::sort(q.begin(), q.end());
      islands.push_back(q);
    }
  }
  // generalized force of a free body (gravity + the gyroscopic term), as fwd_dyn uses them
  void gen_force(int b, double f[6]) const {
    const double m = sc->mass[b];
    f[0] = sc->gravity[0] * m; f[1] = sc->gravity[1] * m; f[2] = sc->gravity[2] * m;
    double Jw[9]; inertia_world(b, Jw);
    const V3 w = Wa(b);
    const V3 Jww = v3((Jw[0]*w.x + Jw[1]*w.y) + Jw[2]*w.z, (Jw[3]*w.x + Jw[4]*w.y) + Jw[5]*w.z, (Jw[6]*w.x + Jw[7]*w.y) + Jw[8]*w.z);
    const V3 tau = -cross(w, Jww);
    f[3] = tau.x; f[4] = tau.y; f[5] = tau.z;
  }
  // Simulator::solve (Sim:608-805): accelerations a (6 per island body) of one island with implicit joints.
  //   (J iM J') lambda = J v + J iM f dt on the largest leading full-rank row set (greedy Cholesky), a = (iM f dt - iM J' lambda) / dt
  // Dense products accumulate from 0 over ascending indices.  Returns false when the island is beyond the built sizes.
  bool solve_kkt(const std::vector<int>& island, const std::vector<int>& joints, double dt, std::vector<double>& a) const {
    const int nbod = (int)island.size(), ngc = 6 * nbod;
    int m = 0;
    for (int j : joints) m += joint_rows(sc->joint_type[j]);
    if (nbod > MH_IJOINT_MAX_BODIES || (int)joints.size() > MH_IJOINT_MAX_JOINTS || m > MH_IJOINT_MAX_EQNS) return false;
    auto gc_of_body = [&](int b) { for (int i = 0; i < nbod; i++) if (island[i] == b) return 6 * i; return -1; };
    std::vector<double> iM((size_t)nbod * 36, 0.0), f(ngc), v(ngc), iMf(ngc);
    for (int i = 0; i < nbod; i++) {
      const int b = island[i];
      double im, Ji[9]; inv_inertia(b, im, Ji);
      double* B = &iM[(size_t)i * 36];                                   // row-major 6 x 6, blockdiag(im I, Ji)
      for (int k = 0; k < 3; k++) B[7 * k] = im;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) B[6 * (3 + r) + 3 + c] = Ji[3 * r + c];
      gen_force(b, &f[6 * i]);
      const V3 vl = Vl(b), wa = Wa(b);
      double* vv = &v[6 * i]; vv[0] = vl.x; vv[1] = vl.y; vv[2] = vl.z; vv[3] = wa.x; vv[4] = wa.y; vv[5] = wa.z;
      for (int r = 0; r < 6; r++) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + B[6 * r + k] * f[6 * i + k]; iMf[6 * i + r] = acc; }
    }
    for (int g = 0; g < ngc; g++) iMf[g] = iMf[g] * dt;
    // Jacobian blocks: per joint the inboard block (if enabled) then the outboard block (Sim:679-706)
    struct Blk { int row, off, rows; double w[6][6]; };
    std::vector<Blk> blocks;
    int eq = 0;
    for (int j : joints) {
      const int rows = joint_rows(sc->joint_type[j]);
      const int sides[2] = { sc->joint_inboard[j], sc->joint_outboard[j] };
      for (int sd = 0; sd < 2; sd++) {
        if (!enabled(sides[sd])) continue;
        Blk k; k.row = eq; k.off = gc_of_body(sides[sd]); k.rows = rows;
        joint_jac(j, sd == 0, k.w);
        blocks.push_back(k);
      }
      eq += rows;
    }
    // JiM = J iM (m x ngc), iMJT its transpose
    std::vector<double> JiM((size_t)m * ngc, 0.0);
    for (const Blk& k : blocks) {
      const double* B = &iM[(size_t)(k.off / 6) * 36];
      for (int r = 0; r < k.rows; r++) for (int c = 0; c < 6; c++) {
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + k.w[r][q] * B[6 * q + c];
        JiM[(size_t)(k.row + r) * ngc + k.off + c] = acc;
      }
    }
    // JiMJT (m x m) = J (JiM)', JiMf = JiM f dt, Jv = J v
    std::vector<double> JiMJT((size_t)m * m, 0.0), JiMf(m, 0.0), Jv(m, 0.0);
    for (const Blk& k : blocks) for (int r = 0; r < k.rows; r++) {
      for (int c = 0; c < m; c++) {
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + k.w[r][q] * JiM[(size_t)c * ngc + k.off + q];
        JiMJT[(size_t)(k.row + r) * m + c] = JiMJT[(size_t)(k.row + r) * m + c] + acc;
      }
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + k.w[r][q] * v[k.off + q];
      Jv[k.row + r] = Jv[k.row + r] + acc;
    }
    for (int r = 0; r < m; r++) { double acc = 0.0; for (int g = 0; g < ngc; g++) acc = acc + JiM[(size_t)r * ngc + g] * f[g]; JiMf[r] = acc * dt; }
    // the biggest full-rank leading set (Sim:728-755): Cholesky of the selected square, row by row
    std::vector<int> act;
    std::vector<double> L;
    for (int i = 0; i < m; i++) {
      if ((int)act.size() == ngc) break;
      act.push_back(i);
      const int k = (int)act.size();
      L.assign((size_t)k * k, 0.0);
      for (int r = 0; r < k; r++) for (int c = 0; c < k; c++) L[r + (size_t)k * c] = JiMJT[(size_t)act[r] * m + act[c]];
      if (!chol_factor(k, L.data(), k)) act.pop_back();
    }
    const int k = (int)act.size();
    L.assign((size_t)k * k, 0.0);
    for (int r = 0; r < k; r++) for (int c = 0; c < k; c++) L[r + (size_t)k * c] = JiMJT[(size_t)act[r] * m + act[c]];
    if (k > 0) chol_factor(k, L.data(), k);
    std::vector<double> lam(k);
    for (int r = 0; r < k; r++) lam[r] = JiMf[act[r]] + Jv[act[r]];       // JiMf_frr += Jv_frr (Sim:769)
    if (k > 0) chol_solve(k, L.data(), k, lam.data());
    a.assign(ngc, 0.0);
    for (int g = 0; g < ngc; g++) {
      double acc = 0.0;
      for (int r = 0; r < k; r++) acc = acc + JiM[(size_t)act[r] * ngc + g] * lam[r];   // iMJT_frr lambda (Sim:795)
      a[g] = ((-acc) + iMf[g]) / dt;
    }
    return true;
  }
  // Simulator::calc_fwd_dyn (Sim:482-602) + the velocity integration of do_mini_step (TSS:181-192)
  void fwd_dyn_and_integrate(double h) {
    const int nb = sc->nb;
    if (njoints() == 0) {
      for (int b = 0; b < nb; b++) { V3 xdd, wd; fwd_dyn(b, xdd, wd); setV(b, Vl(b) + xdd * h); setW(b, Wa(b) + wd * h); }
      return;
    }
    std::vector<std::vector<int> > islands; find_body_islands(islands);
    std::vector<double> acc(6 * (size_t)nb, 0.0);
    for (const std::vector<int>& isl : islands) {
      std::vector<int> ij;                                           // the island's implicit joints (Sim:506-520)
      for (int j = 0; j < njoints(); j++) {
        const int a = sc->joint_inboard[j], b = sc->joint_outboard[j];
        if ((enabled(a) && std::binary_search(isl.begin(), isl.end(), a)) || (enabled(b) && std::binary_search(isl.begin(), isl.end(), b))) ij.push_back(j);
      }
      if (ij.empty()) {
        for (int b : isl) { V3 xdd, wd; fwd_dyn(b, xdd, wd); double* o = &acc[6 * (size_t)b]; o[0] = xdd.x; o[1] = xdd.y; o[2] = xdd.z; o[3] = wd.x; o[4] = wd.y; o[5] = wd.z; }
        continue;
      }
      // DEVIATION (deviation 9): a mini-step that conservative advancement cut to h = 0 (a resting contact) makes
      // Simulator::solve divide by dt = 0 -- NaN accelerations, then NaN * 0 velocities in the reference.  The velocity change
      // of such a mini-step is a * 0: the island keeps its velocities.
      if (!(h > 0.0)) continue;
      std::vector<double> a;
      if (!solve_kkt(isl, ij, h, a)) { aux->status |= MH_WORLD_UNSUPPORTED; continue; }
      for (size_t i = 0; i < isl.size(); i++) for (int k = 0; k < 6; k++) acc[6 * (size_t)isl[i] + k] = a[6 * i + k];
    }
    for (int b = 0; b < nb; b++) {
      const double* o = &acc[6 * (size_t)b];
      setV(b, Vl(b) + v3(o[0], o[1], o[2]) * h);
      setW(b, Wa(b) + v3(o[3], o[4], o[5]) * h);
    }
  }

  // ---- impact handling ------------------------------------------------------
  struct Island { std::vector<int> contacts; std::vector<int> bodies; };

  // UnilateralConstraint::determine_connected_constraints (UC:940-1194), contacts only
  void find_islands(const std::vector<Contact>& cs, std::vector<Island>& out) const {
    out.clear();
    const int nb = nb_;
    std::vector<char> node(nb, 0), done(cs.size(), 0);
    std::vector<std::vector<int> > adj(nb);           // multimap in insertion order
    for (size_t i = 0; i < cs.size(); i++) {
      const int a = cs[i].g1, b = cs[i].g2;
      if (enabled(a)) node[a] = 1;
      if (enabled(b)) node[b] = 1;
      if (enabled(a) && enabled(b)) { adj[a].push_back(b); adj[b].push_back(a); }
    }
    for (int j = 0; j < njoints(); j++) {                 // the implicit joints' links are nodes too, joined by an edge (UC:993-1008)
      const int a = sc->joint_inboard[j], b = sc->joint_outboard[j];
      if (enabled(a)) node[a] = 1;
      if (enabled(b)) node[b] = 1;
      if (enabled(a) && enabled(b)) { adj[a].push_back(b); adj[b].push_back(a); }
    }
    for (int start = 0; start < nb; start++) {
      if (!node[start]) continue;
      Island isl;
      // The reference pushes a neighbour once per multimap edge while it is not yet *processed* (UC:1104-1109), so a
      // node reached over k parallel edges is queued k times and a chain of such nodes 4, 16, 64, ... times: the queue
      // grows exponentially with the height of a box stack (4 contacts per interface).  Only the FIRST pop of a node
      // adds constraints, and a later pop can only push nodes that were pushed before, so marking nodes when they
      // are first pushed gives the same island (contact order, body set) in linear time.
      std::vector<int> queue; queue.push_back(start);
      std::vector<char> queued(nb, 0); queued[start] = 1;
      for (size_t qi = 0; qi < queue.size(); qi++) {
        const int nd = queue[qi];
        node[nd] = 0;
        isl.bodies.push_back(nd);
        for (int nbr : adj[nd]) if (!queued[nbr]) { queued[nbr] = 1; queue.push_back(nbr); }
        for (size_t i = 0; i < cs.size(); i++)
          if (!done[i] && (cs[i].g1 == nd || cs[i].g2 == nd)) { isl.contacts.push_back((int)i); done[i] = 1; }
      }
      if (!isl.contacts.empty()) out.push_back(isl);
    }
  }

  // problem data of one island: rows [d, r x d] per body (ICH:1847-1895),
  // X = blockdiag(M_i^-1) (ICH:1590-1695 without bilateral rows), the C X C' blocks
  // (ICH:2125-2149 via SparseJacobian::mult, SparseJacobian.cpp:46-82) and C v.
  struct ProblemData {
    int nc, ngc;
    std::vector<int> bodies;                 // sorted unique super bodies
    std::vector<const Contact*> c;
    std::vector<double> J[3];                // Cn, Cs, Ct dense rows: nc x ngc (row-major)
    std::vector<double> XJ[3];               // X * C^T : ngc x nc (col-major, column = contact)
    std::vector<double> G[3][3];             // C_a X C_b^T nc x nc row-major (a<=b used)
    std::vector<double> Cv[3];
    std::vector<double> cn, cs, ct;
    std::vector<double> X;                   // ngc x ngc dense (block diagonal)
  };

  int gc_of(const ProblemData& pd, int body) const {
    for (size_t i = 0; i < pd.bodies.size(); i++) if (pd.bodies[i] == body) return 6 * (int)i;
    return -1;
  }

  void compute_problem_data(const std::vector<Contact>& all, const Island& isl, ProblemData& pd, bool normals_only, const std::vector<double>* Xgen = nullptr) const {
    pd.bodies = isl.bodies;
    std::sort(pd.bodies.begin(), pd.bodies.end());
    pd.bodies.erase(std::unique(pd.bodies.begin(), pd.bodies.end()), pd.bodies.end());
    pd.c.clear();
    for (int ci : isl.contacts) pd.c.push_back(&all[ci]);
    const int nc = pd.nc = (int)pd.c.size();
    const int ngc = pd.ngc = 6 * (int)pd.bodies.size();
    // X
    pd.X.assign((size_t)ngc * ngc, 0.0);
    if (Xgen) pd.X = *Xgen;                                  // compute_X's general case: implicit joints in the island (ICH:1590-1695)
    else for (size_t bi = 0; bi < pd.bodies.size(); bi++) {
      double im, Ji[9]; inv_inertia(pd.bodies[bi], im, Ji);
      const int o = 6 * (int)bi;
      for (int k = 0; k < 3; k++) pd.X[(o + k) * ngc + (o + k)] = im;
      for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) pd.X[(o + 3 + r) * ngc + (o + 3 + cc)] = Ji[3*r+cc];
    }
    const int ndir = normals_only ? 1 : 3;
    // stacked generalized velocity
    std::vector<double> v(ngc, 0.0);
    for (size_t bi = 0; bi < pd.bodies.size(); bi++) {
      const V3 vl = Vl(pd.bodies[bi]), wa = Wa(pd.bodies[bi]);
      double* o = &v[6 * bi]; o[0] = vl.x; o[1] = vl.y; o[2] = vl.z; o[3] = wa.x; o[4] = wa.y; o[5] = wa.z;
    }
    // block lists per row: (gc offset, 6 values) for body g1 then body g2
    struct Blk { int off; double w[6]; };
    std::vector<std::vector<Blk> > blocks[3];
    for (int d = 0; d < ndir; d++) {
      blocks[d].assign(nc, std::vector<Blk>());
      pd.J[d].assign((size_t)nc * ngc, 0.0);
      for (int i = 0; i < nc; i++) {
        const Contact& c = *pd.c[i];
        const V3 dir = (d == 0) ? c.n : (d == 1 ? c.s : c.t);
        const int bodies2[2] = { c.g1, c.g2 };
        for (int k = 0; k < 2; k++) {
          const int b = bodies2[k];
          if (!enabled(b)) continue;
          const V3 dd = (k == 0) ? dir : -dir;
          const V3 r = c.p - X(b);
          const V3 rxd = cross(r, dd);
          Blk blk; blk.off = gc_of(pd, b);
          blk.w[0] = dd.x; blk.w[1] = dd.y; blk.w[2] = dd.z; blk.w[3] = rxd.x; blk.w[4] = rxd.y; blk.w[5] = rxd.z;
          blocks[d][i].push_back(blk);
          for (int q = 0; q < 6; q++) pd.J[d][(size_t)i * ngc + blk.off + q] = blk.w[q];
        }
      }
    }
    // X_C*T = (C X)^T : result(i,col) = sum_blocks [ sum_k block[k] * X[off+k][col] ]
    for (int d = 0; d < ndir; d++) {
      pd.XJ[d].assign((size_t)ngc * nc, 0.0);
      for (int i = 0; i < nc; i++)
        for (int col = 0; col < ngc; col++) {
          double res = 0.0;
          for (const Blk& blk : blocks[d][i]) {
            double tmp = 0.0;
            for (int k = 0; k < 6; k++) tmp = tmp + blk.w[k] * pd.X[(blk.off + k) * ngc + col];
            res = res + tmp;
          }
          pd.XJ[d][(size_t)col + (size_t)ngc * i] = res;
        }
    }
    // C_a X C_b^T (i,j) = sum_blocks of row i [ sum_k block[k] * X_CbT[off+k][j] ]
    for (int a = 0; a < ndir; a++) for (int b = a; b < ndir; b++) {
      pd.G[a][b].assign((size_t)nc * nc, 0.0);
      for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {
        double res = 0.0;
        for (const Blk& blk : blocks[a][i]) {
          double tmp = 0.0;
          for (int k = 0; k < 6; k++) tmp = tmp + blk.w[k] * pd.XJ[b][(size_t)(blk.off + k) + (size_t)ngc * j];
          res = res + tmp;
        }
        pd.G[a][b][(size_t)i * nc + j] = res;
      }
    }
    // C v
    for (int d = 0; d < ndir; d++) {
      pd.Cv[d].assign(nc, 0.0);
      for (int i = 0; i < nc; i++) {
        double res = 0.0;
        for (const Blk& blk : blocks[d][i]) {
          double tmp = 0.0;
          for (int k = 0; k < 6; k++) tmp = tmp + blk.w[k] * v[blk.off + k];
          res = res + tmp;
        }
        pd.Cv[d][i] = res;
      }
    }
    pd.cn.assign(nc, 0.0); pd.cs.assign(nc, 0.0); pd.ct.assign(nc, 0.0);
  }

  // element (r,c) of block a,b with a<=b stored; transposes give the rest (setup_QP :392-401)
  static double Gab(const ProblemData& pd, int a, int b, int i, int j) {
    return (a <= b) ? pd.G[a][b][(size_t)i * pd.nc + j] : pd.G[b][a][(size_t)j * pd.nc + i];
  }

  // ImpactConstraintHandler::solve_qp_work + setup_QP (ICH-QP:94-263, 271-497), no limits.
  // z layout of the LCP: [cn cs ct ncs nct | Cn v+>=0 rows | friction polygon rows]
  void build_impact_lcp(const ProblemData& pd, std::vector<double>& MM, std::vector<double>& qq, int& n) const {
    const int nc = pd.nc;
    const int nvars = 5 * nc;
    int nk_total = 0;
    for (int i = 0; i < nc; i++) nk_total += pd.c[i]->nk / 2;
    const int nineq = nc + nk_total;
    n = nvars + nineq;
    MM.assign((size_t)n * n, 0.0); qq.assign(n, 0.0);   // column-major
    auto at = [&](int r, int c) -> double& { return MM[(size_t)r + (size_t)n * c]; };
    // H: 5x5 blocks over [n, s, t, -s, -t]
    const int dirs[5] = { 0, 1, 2, 1, 2 }; const double sgn[5] = { 1, 1, 1, -1, -1 };
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++)
      for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {
        double g = Gab(pd, dirs[a], dirs[b], i, j);
        if (sgn[a] * sgn[b] < 0) g = -g;
        at(a * nc + i, b * nc + j) = g;
      }
    for (int i = 0; i < nc; i++) at(i, i) = at(i, i) + pd.c[i]->compliance;     // ICH-QP:438-440
    // c
    for (int i = 0; i < nc; i++) {
      qq[i] = pd.Cv[0][i]; qq[nc + i] = pd.Cv[1][i]; qq[2*nc + i] = pd.Cv[2][i];
      qq[3*nc + i] = -pd.Cv[1][i]; qq[4*nc + i] = -pd.Cv[2][i];
    }
    // M rows: Cn v+ >= 0 (copy of H's first block row, with compliance), then friction polygons
    for (int i = 0; i < nc; i++) for (int c = 0; c < nvars; c++) at(nvars + i, c) = at(i, c);
    for (int i = 0; i < nc; i++) qq[nvars + i] = pd.Cv[0][i];
    int row = nvars + nc;
    for (int i = 0; i < nc; i++) {
      const double vel = std::sqrt(pd.Cv[1][i] * pd.Cv[1][i] + pd.Cv[2][i] * pd.Cv[2][i]);
      const int kh = pd.c[i]->nk / 2;
      for (int j = 0; j < kh; j++) {
        const double theta = (double)j / (kh - 1) * M_PI_2;
        const double ct = std::cos(theta), st_ = std::sin(theta);
        at(row, i) = pd.c[i]->mu;
        at(row, nc + i) = -ct; at(row, 3*nc + i) = -ct;
        at(row, 2*nc + i) = -st_; at(row, 4*nc + i) = -st_;
        qq[row] = pd.c[i]->muv * vel;
        row++;
      }
    }
    // upper right = -M^T
    for (int r = nvars; r < n; r++) for (int c = 0; c < nvars; c++) at(c, r) = -at(r, c);
  }

  void lcp_account(int n, unsigned pivots) { aux->lcp_solves++; aux->lcp_rows += (unsigned long long)n; aux->lcp_pivots += pivots; aux->lcp_alg_bytes += 8ull * ((unsigned long long)n * n + 2ull * n); }

  // solve_qp_work's solver chain (ICH-QP:157-233) on the persistent _z / _zlast
  bool solve_impact_lcp(const std::vector<double>& MM, const std::vector<double>& qq, int n, std::vector<double>& zout) {
    Vec z; z.d.assign(zbuf_, zbuf_ + aux->zbuf_cap); z.len = (unsigned)aux->zbuf_size;
    z.resize((unsigned)n);                                   // ICH-QP:158
    if ((int)z.size() == aux->zlast_size) for (int i = 0; i < n; i++) z[i] = zlast_[i];
    oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
    LCP lcp; lcp.rng = &rs;
    Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? trace_cap - trace_len : 0; if (tr.cap < 0) tr.cap = 0;
    lcp.trace = &tr;
    unsigned piv = 0;
    std::vector<double> z_in(z.d.begin(), z.d.begin() + n); oracle_rand_t rs_in = rs;
    bool ok = lcp.lcp_fast_regularized(n, MM.data(), n, qq.data(), z, -20, 4, -8);
    piv += lcp.pivots;
    const unsigned piv_fast = lcp.pivots; const bool ok_fast = ok; unsigned piv_lemke = 0;
    if (!ok) {
      z.set_zero();                                          // ICH-QP:222
      ok = lcp.lcp_lemke_regularized(n, MM.data(), n, qq.data(), z);
      piv += lcp.pivots; piv_lemke = lcp.pivots;
    }
    trace_len += tr.len;
    std::memcpy(aux->rng, &rs, sizeof(rs));
    lcp_account(n, piv);
    if (g_lcp_dump) {                                        // int n, ok_fast, piv_fast, piv_lemke, ok; rng (32 words); MM, qq, z_in
      const int hdr[5] = { n, ok_fast ? 1 : 0, (int)piv_fast, (int)piv_lemke, ok ? 1 : 0 };
      std::fwrite(hdr, sizeof(int), 5, g_lcp_dump); std::fwrite(&rs_in, sizeof(rs_in), 1, g_lcp_dump);
      std::fwrite(MM.data(), 8, (size_t)n * n, g_lcp_dump); std::fwrite(qq.data(), 8, n, g_lcp_dump); std::fwrite(z_in.data(), 8, n, g_lcp_dump);
      std::fflush(g_lcp_dump);
    }
    if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; thrown_ = true; return false; }   // LCPSolverException
    // _zlast = z (ICH-QP:233)
    aux->zlast_size = n;
    for (int i = 0; i < n; i++) zlast_[i] = z[i];
    zout.assign(z.d.begin(), z.d.begin() + n);
    // persist _z's storage: entries [0,n) are rewritten, entries beyond keep what
    // they held (Ravelin keeps a vector's storage when it shrinks).  Growth of
    // the buffer inside lcp_lemke (z.set_zero(2n)) is not modelled: only reads
    // of [0,n) ever happen (documented deviation).
    for (int i = 0; i < n; i++) zbuf_[i] = z[i];
    if (aux->zbuf_cap < n) aux->zbuf_cap = n;
    aux->zbuf_size = n;
    return true;
  }

  // update_from_stacked (ICH:298-410): dv = X_CnT cn + X_CsT cs + X_CtT ct; v += dv
  void apply_impulses(const ProblemData& pd) {
    const int ngc = pd.ngc, nc = pd.nc;
    std::vector<double> dv(ngc, 0.0), tmp(ngc);
    const std::vector<double>* imp[3] = { &pd.cn, &pd.cs, &pd.ct };
    for (int d = 0; d < 3; d++) {
      if (pd.XJ[d].empty()) continue;
      std::fill(tmp.begin(), tmp.end(), 0.0);
      for (int j = 0; j < nc; j++) { const double t = (*imp[d])[j]; for (int r = 0; r < ngc; r++) tmp[r] = tmp[r] + t * pd.XJ[d][(size_t)r + (size_t)ngc * j]; }
      if (d == 0) dv = tmp; else for (int r = 0; r < ngc; r++) dv[r] = dv[r] + tmp[r];
    }
    for (int j = 0; j < nc; j++) {
      pd.c[j]->imp[0] += pd.cn[j];
      if (!pd.XJ[1].empty()) { pd.c[j]->imp[1] += pd.cs[j]; pd.c[j]->imp[2] += pd.ct[j]; }
    }
    for (size_t bi = 0; bi < pd.bodies.size(); bi++) {
      const int b = pd.bodies[bi]; const double* o = &dv[6 * bi];
      setV(b, Vl(b) + v3(o[0], o[1], o[2]));
      setW(b, Wa(b) + v3(o[3], o[4], o[5]));
    }
  }
  // update_constraint_velocities_from_impulses (ICH:427-464)
  void update_constraint_vels(ProblemData& pd) const {
    const int nc = pd.nc;
    auto addmul = [&](std::vector<double>& y, int a, int b, const std::vector<double>& x, bool) {
      // y += G_ab * x (dgemv: tmp = 0; for j: tmp += x_j * col_j; y += tmp); G_ab(i,j) via Gab handles transposes
      std::vector<double> t(nc, 0.0);
      for (int j = 0; j < nc; j++) { const double xj = x[j]; for (int i = 0; i < nc; i++) t[i] = t[i] + xj * Gab(pd, a, b, i, j); }
      for (int i = 0; i < nc; i++) y[i] = y[i] + t[i];
    };
    addmul(pd.Cv[0], 0, 0, pd.cn, false); addmul(pd.Cv[0], 0, 1, pd.cs, false); addmul(pd.Cv[0], 0, 2, pd.ct, false);
    addmul(pd.Cv[1], 1, 0, pd.cn, true);  addmul(pd.Cv[1], 1, 1, pd.cs, false); addmul(pd.Cv[1], 1, 2, pd.ct, false);
    addmul(pd.Cv[2], 2, 0, pd.cn, true);  addmul(pd.Cv[2], 2, 1, pd.cs, true);  addmul(pd.Cv[2], 2, 2, pd.ct, false);
  }
  static void from_stacked_qp(ProblemData& pd, const std::vector<double>& zepd) {   // UCPD:218-228
    const int nc = pd.nc;
    for (int i = 0; i < nc; i++) {
      pd.cn[i] = zepd[i];
      double s = zepd[nc + i];   s = s - zepd[3*nc + i]; pd.cs[i] = s;
      double t = zepd[2*nc + i]; t = t - zepd[4*nc + i]; pd.ct[i] = t;
    }
  }

  // ImpactConstraintHandler::apply_no_slip_model (ICH:1009-1417), contacts only (no limits, no
  // bilateral rows).  Dense products are restated in the order the HIP kernel uses: dot products
  // accumulate from 0 in ascending index order.
  bool apply_no_slip_model(ProblemData& pd) {
    const int nc = pd.nc;
    if (nc > MH_NOSLIP_MAX) { aux->status |= MH_WORLD_UNSUPPORTED; return false; }
    std::vector<int> S, T;
    std::vector<double> Y;
    auto build_Y = [&](bool skew) -> int {
      const int ns = (int)S.size(), nt = (int)T.size(), m = ns + nt;
      Y.assign((size_t)m * m, 0.0);                        // column-major
      for (int a = 0; a < ns; a++) for (int b = 0; b < ns; b++) Y[a + (size_t)m * b] = pd.G[1][1][(size_t)S[a] * nc + S[b]];
      for (int a = 0; a < nt; a++) for (int b = 0; b < nt; b++) Y[(ns + a) + (size_t)m * (ns + b)] = pd.G[2][2][(size_t)T[a] * nc + T[b]];
      for (int a = 0; a < ns; a++) for (int b = 0; b < nt; b++) {
        const double g = pd.G[1][2][(size_t)S[a] * nc + T[b]];
        Y[a + (size_t)m * (ns + b)] = g; Y[(ns + b) + (size_t)m * a] = g;
      }
      if (skew) for (int j = 0; j < m; j++) Y[j + (size_t)m * j] = Y[j + (size_t)m * j] - NEAR_ZERO;   // ICH:1110-1111
      return m;
    };
    // greedy largest non-singular tangent set (ICH:1087-1145)
    for (int i = 0; i < nc; i++) {
      S.push_back(i);
      int m = build_Y(true);
      if (!chol_factor(m, Y.data(), m)) S.pop_back();
      T.push_back(i);
      m = build_Y(true);
      if (!chol_factor(m, Y.data(), m)) T.pop_back();
    }
    const int ns = (int)S.size(), nt = (int)T.size();
    const int m = build_Y(false);                            // ICH:1166-1183
    if (!chol_factor(m, Y.data(), m)) { aux->status |= MH_WORLD_LCP_FAILED; thrown_ = true; return false; }   // assert(success)
    // Q X X^T (nc x m): [Cn X Cs^T(:,S)  Cn X Ct^T(:,T)] (ICH:1198-1203)
    std::vector<double> QX((size_t)nc * m);
    for (int i = 0; i < nc; i++) {
      for (int a = 0; a < ns; a++) QX[(size_t)i * m + a] = pd.G[0][1][(size_t)i * nc + S[a]];
      for (int a = 0; a < nt; a++) QX[(size_t)i * m + ns + a] = pd.G[0][2][(size_t)i * nc + T[a]];
    }
    // W = Y^-1 (Q X X^T)^T, column by column (ICH:1210-1211)
    std::vector<double> W((size_t)m * nc), col(m);
    for (int j = 0; j < nc; j++) {
      for (int a = 0; a < m; a++) col[a] = QX[(size_t)j * m + a];
      chol_solve(m, Y.data(), m, col.data());
      for (int a = 0; a < m; a++) W[a + (size_t)m * j] = col[a];
    }
    // MM = Cn X Cn^T - QX W (ICH:1190-1215)
    std::vector<double> MM((size_t)nc * nc), qq(nc);
    for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {
      double acc = 0.0;
      for (int a = 0; a < m; a++) acc = acc + QX[(size_t)i * m + a] * W[a + (size_t)m * j];
      MM[i + (size_t)nc * j] = pd.G[0][0][(size_t)i * nc + j] - acc;
    }
    // qq = Cn v - QX Y^-1 [Cs v(S); Ct v(T)] (ICH:1217-1236)
    std::vector<double> YXv(m);
    for (int a = 0; a < ns; a++) YXv[a] = pd.Cv[1][S[a]];
    for (int a = 0; a < nt; a++) YXv[ns + a] = pd.Cv[2][T[a]];
    chol_solve(m, Y.data(), m, YXv.data());
    for (int i = 0; i < nc; i++) {
      double acc = 0.0;
      for (int a = 0; a < m; a++) acc = acc + QX[(size_t)i * m + a] * YXv[a];
      qq[i] = pd.Cv[0][i] - acc;
    }
    // lcp_fast on the persistent _v, then the Lemke ladder (ICH:1239, 1281)
    Vec z; z.d.assign(aux->vns, aux->vns + MH_NOSLIP_MAX); z.len = (unsigned)aux->vns_size;
    oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
    LCP lcp; lcp.rng = &rs;
    Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? std::max(0, trace_cap - trace_len) : 0;
    lcp.trace = &tr;
    unsigned piv = 0;
    bool ok = lcp.lcp_fast(nc, MM.data(), nc, qq.data(), z, -1.0);
    piv += lcp.pivots;
    if (!ok) { ok = lcp.lcp_lemke_regularized(nc, MM.data(), nc, qq.data(), z); piv += lcp.pivots; }
    trace_len += tr.len;
    std::memcpy(aux->rng, &rs, sizeof(rs));
    lcp_account(nc, piv);
    if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; thrown_ = true; return false; }   // std::runtime_error("Unable to solve constraint LCP!")
    for (int i = 0; i < nc; i++) aux->vns[i] = z[i];
    aux->vns_size = nc;
    // [cs; ct] = -(Y^-1 X v + Y^-1 (QX)^T v) (ICH:1293-1298)
    std::vector<double> t2(m);
    for (int a = 0; a < m; a++) {
      double acc = 0.0;
      for (int i = 0; i < nc; i++) acc = acc + QX[(size_t)i * m + a] * z[i];
      t2[a] = acc;
    }
    chol_solve(m, Y.data(), m, t2.data());
    for (int i = 0; i < nc; i++) { pd.cn[i] = z[i]; pd.cs[i] = 0.0; pd.ct[i] = 0.0; }
    for (int a = 0; a < ns; a++) pd.cs[S[a]] = -(YXv[a] + t2[a]);
    for (int a = 0; a < nt; a++) pd.ct[T[a]] = -(YXv[ns + a] + t2[ns + a]);
    apply_impulses(pd);                                              // ICH:1351-1385
    return true;
  }
  // apply_no_slip_model_to_connected_constraints (ICH:236-295)
  void apply_no_slip_model_to_island(const std::vector<Contact>& all, const Island& isl) {
    ProblemData pd; compute_problem_data(all, isl, pd, false);
    const int nc = pd.nc;
    if (!apply_no_slip_model(pd)) return;
    update_constraint_vels(pd);                                      // ICH:265
    const double minv = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());
    bool changed = false;                                            // apply_restitution(q) (ICH:497-525)
    for (int i = 0; i < nc; i++) { pd.cn[i] = pd.cn[i] * pd.c[i]->eps; if (!changed && pd.cn[i] > NEAR_ZERO) changed = true; }
    if (changed) {
      for (int i = 0; i < nc; i++) { pd.cs[i] = 0.0; pd.ct[i] = 0.0; }
      apply_impulses(pd);                                            // update_from_stacked(q) (ICH:274)
      update_constraint_vels(pd);
      const double minv_plus = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());
      // ICH:284-291 would re-solve and then read the D-S solver's _z, which this path never
      // sized: undefined in the reference, reported here
      if (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO) aux->status |= MH_WORLD_UNSUPPORTED;
    }
  }

  // ---- Anitescu-Potra model (ImpactConstraintHandlerLCP.cpp; the reference builds it with -DUSE_AP_MODEL) ------------
  // NK_DIRS rows of one contact (ICH-AP:113-118)
  static int ap_rows(int nk) { return (nk > 4) ? (nk + 4) / 4 : 1; }
  // apply_ap_model's LCP (ICH-AP:94-330), contacts only: _MM = [UL UR; LL 0] over z = [cn, cs+, cs-, ct+, ct-, friction rows],
  // UL block (a, b) = +/- C_a X C_b^T, LL = [mu, -cos, -cos, -sin, -sin], UR = friction part of -LL^T; column-major
  void build_ap_lcp(const ProblemData& pd, std::vector<double>& MM, std::vector<double>& qq, int& n) const {
    const int nc = pd.nc, nvars = 5 * nc;
    int nkdirs = 0;
    for (int i = 0; i < nc; i++) nkdirs += ap_rows(pd.c[i]->nk);
    n = nvars + nkdirs;
    MM.assign((size_t)n * n, 0.0); qq.assign(n, 0.0);
    auto at = [&](int r, int c) -> double& { return MM[(size_t)r + (size_t)n * c]; };
    const int dirs[5] = { 0, 1, 1, 2, 2 }; const double sgn[5] = { 1, 1, -1, 1, -1 };
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++)
      for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {
        double g = Gab(pd, dirs[a], dirs[b], i, j);
        if (sgn[a] * sgn[b] < 0) g = -g;
        at(a * nc + i, b * nc + j) = g;
      }
    int r = nvars;
    for (int i = 0; i < nc; i++) {
      const int rows = ap_rows(pd.c[i]->nk);
      for (int k = 0; k < rows; k++) {
        // ICH-AP:250-295: polygon directions cos / sin(pi k / (2 nk4)); a 4-edge cone is the single row of ones
        const double ck = (pd.c[i]->nk > 4) ? std::cos((M_PI * k) / (2.0 * rows)) : 1.0;
        const double sk = (pd.c[i]->nk > 4) ? std::sin((M_PI * k) / (2.0 * rows)) : 1.0;
        at(r + k, i) = pd.c[i]->mu;
        at(r + k, nc + i) = -ck; at(r + k, 2*nc + i) = -ck; at(r + k, 3*nc + i) = -sk; at(r + k, 4*nc + i) = -sk;
        at(nc + i, r + k) = ck;  at(2*nc + i, r + k) = ck;  at(3*nc + i, r + k) = sk;  at(4*nc + i, r + k) = sk;
      }
      r += rows;
    }
    for (int i = 0; i < nc; i++) {                               // ICH-AP:186-187, 305-311
      qq[i] = pd.Cv[0][i]; qq[nc + i] = pd.Cv[1][i]; qq[2*nc + i] = -pd.Cv[1][i];
      qq[3*nc + i] = pd.Cv[2][i]; qq[4*nc + i] = -pd.Cv[2][i];
    }
  }
  // propagate_impulse_data (ICH:643-673): contact_impulse += the wrench of j = n cn + s cs + t ct at the contact point,
  // moved to the global frame (angular part p x j)
  void ap_propagate(const ProblemData& pd, std::vector<double>& wrench) const {
    for (int i = 0; i < pd.nc; i++) {
      const Contact& c = *pd.c[i];
      V3 j = c.n * pd.cn[i]; j = j + c.s * pd.cs[i]; j = j + c.t * pd.ct[i];
      const V3 a = cross(c.p, j);
      double* w = &wrench[6 * (size_t)i];
      w[0] = w[0] + j.x; w[1] = w[1] + j.y; w[2] = w[2] + j.z; w[3] = w[3] + a.x; w[4] = w[4] + a.y; w[5] = w[5] + a.z;
      c.imp[0] += pd.cn[i]; c.imp[1] += pd.cs[i]; c.imp[2] += pd.ct[i];
    }
  }
  bool apply_ap_model(ProblemData& pd, std::vector<double>& wrench) {
    std::vector<double> MM, qq; int n;
    build_ap_lcp(pd, MM, qq, n);
    if (n > lcp_cap_) { aux->status |= MH_WORLD_UNSUPPORTED; return false; }
    Vec z;                                                         // a fresh VectorNd (ICH-AP:331)
    oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
    LCP lcp; lcp.rng = &rs;
    Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? std::max(0, trace_cap - trace_len) : 0;
    lcp.trace = &tr;
    const bool ok = lcp.lcp_lemke_regularized(n, MM.data(), n, qq.data(), z, -20, 1, -2);   // ICH-AP:333
    trace_len += tr.len;
    std::memcpy(aux->rng, &rs, sizeof(rs));
    lcp_account(n, lcp.pivots);
    if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; thrown_ = true; return false; }   // throw std::exception()
    const int nc = pd.nc;
    for (int i = 0; i < nc; i++) {                                   // ICH-AP:336-342
      pd.cn[i] = z[i];
      pd.cs[i] = z[nc + i] - z[2*nc + i];
      pd.ct[i] = z[3*nc + i] - z[4*nc + i];
    }
    ap_propagate(pd, wrench);                                        // ICH-AP:350
    return true;
  }
  // apply_impulses (ICH:676-745) for free bodies: every contact's accumulated wrench w on geom1's body, -w on geom2's,
  // as generalized forces about the body's centre (convert_to_generalized_force), summed per body in contact order;
  // then apply_generalized_impulse: v += M^-1 gj.  (Ravelin arithmetic, unpinned: the order here is the build's.)
  void ap_apply_impulses(const ProblemData& pd, const std::vector<double>& wrench) {
    const size_t nbod = pd.bodies.size();
    std::vector<double> gj(6 * nbod, 0.0); std::vector<char> seen(nbod, 0);
    for (int i = 0; i < pd.nc; i++) {
      const Contact& c = *pd.c[i];
      const int bodies2[2] = { c.g1, c.g2 };
      for (int k = 0; k < 2; k++) {
        const int b = bodies2[k];
        if (!enabled(b)) continue;
        const double sg = (k == 0) ? 1.0 : -1.0;
        const V3 f = v3(sg * wrench[6*(size_t)i], sg * wrench[6*(size_t)i+1], sg * wrench[6*(size_t)i+2]);
        const V3 tq = v3(sg * wrench[6*(size_t)i+3], sg * wrench[6*(size_t)i+4], sg * wrench[6*(size_t)i+5]) - cross(X(b), f);
        const int o = gc_of(pd, b); const size_t bi = (size_t)o / 6;
        const double g[6] = { f.x, f.y, f.z, tq.x, tq.y, tq.z };
        if (!seen[bi]) { for (int q = 0; q < 6; q++) gj[o + q] = g[q]; seen[bi] = 1; }
        else for (int q = 0; q < 6; q++) gj[o + q] = gj[o + q] + g[q];
      }
    }
    for (size_t bi = 0; bi < nbod; bi++) {
      if (!seen[bi]) continue;
      const int b = pd.bodies[bi];
      double im, Ji[9]; inv_inertia(b, im, Ji);
      const double* g = &gj[6 * bi];
      const V3 dv = v3(im * g[0], im * g[1], im * g[2]);
      const V3 dw = v3((Ji[0] * g[3] + Ji[1] * g[4]) + Ji[2] * g[5], (Ji[3] * g[3] + Ji[4] * g[4]) + Ji[5] * g[5], (Ji[6] * g[3] + Ji[7] * g[4]) + Ji[8] * g[5]);
      setV(b, Vl(b) + dv); setW(b, Wa(b) + dw);
    }
  }
  // apply_ap_model_to_connected_constraints (ICH-AP:36-92)
  void apply_ap_model_to_island(const std::vector<Contact>& all, const Island& isl) {
    ProblemData pd; compute_problem_data(all, isl, pd, false);
    const int nc = pd.nc;
    for (int i = 0; i < nc; i++) { pd.c[i]->imp[0] = 0.0; pd.c[i]->imp[1] = 0.0; pd.c[i]->imp[2] = 0.0; }   // ICH-AP:52-55
    std::vector<double> wrench(6 * (size_t)nc, 0.0);
    if (!apply_ap_model(pd, wrench)) return;
    update_constraint_vels(pd);                                      // ICH-AP:61
    const double minv = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());
    bool changed = false;                                            // apply_restitution(q) (ICH:497-525)
    for (int i = 0; i < nc; i++) { pd.cn[i] = pd.cn[i] * pd.c[i]->eps; if (!changed && pd.cn[i] > NEAR_ZERO) changed = true; }
    if (changed) {
      for (int i = 0; i < nc; i++) { pd.cs[i] = 0.0; pd.ct[i] = 0.0; }
      update_constraint_vels(pd);
      const double minv_plus = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());
      if (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO) {        // ICH-AP:78-82: the restitution impulses are never
        if (!apply_ap_model(pd, wrench)) return;                    // propagated on this branch (the reference's behaviour)
      } else ap_propagate(pd, wrench);
    }
    ap_apply_impulses(pd, wrench);                                   // ICH-AP:88
  }

  // apply_model_to_connected_constraints (ICH:530-626), Drumwright-Shell path
  void apply_model(const std::vector<Contact>& all, const Island& isl) {
    ProblemData pd; compute_problem_data(all, isl, pd, false);
    const int nc = pd.nc;
    std::vector<double> MM, qq, z; int n;
    auto solve_and_store = [&]() -> bool {
      build_impact_lcp(pd, MM, qq, n);
      if (n > lcp_cap_) { aux->status |= MH_WORLD_UNSUPPORTED; return false; }
      if (!solve_impact_lcp(MM, qq, n, z)) return false;
      // repack z in the epd layout (ICH-QP:236-250): here identical to the first 5 nc entries
      for (int i = 0; i < 5 * nc; i++) zbuf_[i] = z[i];
      aux->zbuf_size = 5 * nc;
      if (aux->zbuf_cap < 5 * nc) aux->zbuf_cap = 5 * nc;
      return true;
    };
    if (!solve_and_store()) return;
    std::vector<double> zepd(zbuf_, zbuf_ + 5 * nc);
    from_stacked_qp(pd, zepd); apply_impulses(pd);                  // ICH:569
    update_constraint_vels(pd);                                      // ICH:572
    double minv = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());   // ICH:575
    // apply_restitution(_epd, _z) (ICH:470-491): scales z[cn] only
    bool changed = false;
    for (int i = 0; i < nc; i++) { zepd[i] = zepd[i] * pd.c[i]->eps; if (!changed && zepd[i] > NEAR_ZERO) changed = true; }
    for (int i = 0; i < nc; i++) zbuf_[i] = zepd[i];
    if (changed) {
      from_stacked_qp(pd, zepd); apply_impulses(pd);                // ICH:581
      update_constraint_vels(pd);
      const double minv_plus = *std::min_element(pd.Cv[0].begin(), pd.Cv[0].end());
      if (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO) {        // ICH:591
        // second solve reuses the problem data with the updated C*v vectors
        if (!solve_and_store()) return;
        std::vector<double> z2(zbuf_, zbuf_ + 5 * nc);
        from_stacked_qp(pd, z2); apply_impulses(pd);                // ICH:600
      }
    }
  }

  // calc_impacting_unilateral_constraint_forces (CSim:298-355) + apply_model (ICH:96-168)
  // An exception of the impact handler -- LCPSolverException (ICH-QP:225), std::runtime_error("Unable to solve constraint LCP!") (ICH:1282),
  // std::exception (ICH-AP:334), the assert of ICH:1184-1186 -- is caught nowhere between apply_model and main() (CSim:342-350 catches
  // ImpactToleranceException only; programs/driver.cpp, programs/regress.cpp catch nothing): it unwinds process_constraints, do_mini_step and step,
  // and the run of this simulator is over.  thrown_: it has happened in this call; the islands after the failing one are not processed, the
  // tolerance check (ICH:157-167) is not reached, do_mini_step does not advance current_time (TSS:215), step does not stabilise; a world whose
  // status carries MH_WORLD_LCP_FAILED is not stepped again (step() returns at once: the process that owned it has terminated).
  bool thrown_ = false;
  void handle_impacts(const std::vector<Contact>& cs) {
    thrown_ = false;
    if (cs.empty()) return;
    bool none = true;
    for (const Contact& c : cs) if (contact_vel(c, c.n) < -NEAR_ZERO) { none = false; break; }
    if (none) return;
    std::vector<Island> islands; find_islands(cs, islands);
    // remove_inactive_groups (UC:1197-1225)
    std::vector<Island> active;
    for (const Island& isl : islands) {
      bool act = false;
      for (int ci : isl.contacts) if (contact_vel(cs[ci], cs[ci].n) < -NEAR_ZERO) { act = true; break; }
      if (act) active.push_back(isl);
    }
    for (const Island& isl : active) {
      bool all_inf = true;
      for (int ci : isl.contacts) if (cs[ci].mu < 1e2) all_inf = false;
      if (all_inf) apply_no_slip_model_to_island(cs, isl);              // ICH:134-135
      else if (impact_model == MH_IMPACT_MODEL_AP) apply_ap_model_to_island(cs, isl);   // ICH:139-142
      else apply_model(cs, isl);
      if (thrown_) return;
    }
    for (const Island& isl : active)
      for (int ci : isl.contacts) if (contact_vel(cs[ci], cs[ci].n) < -NEAR_ZERO) { aux->status |= MH_WORLD_IMPACT_TOL; }
  }

  // ---- time stepping ------------------------------------------------------------
  std::vector<int> pairs_to_check;            // ConstraintSimulator::_pairs_to_check
  std::vector<PairDist> pairwise;             // _pairwise_distances

  // TimeSteppingSimulator::calc_next_CA_Euler_step (TSS:272-331)
  double next_CA_step() const {
    double t = INF;
    for (const PairDist& d : pairwise) { const double e = CA_step(d); t = (e < t) ? e : t; }
    return t;
  }

  // TimeSteppingSimulator::do_mini_step (TSS:114-222)
  double do_mini_step(double dt) {
    const int nb = sc->nb;
    std::vector<double> qsave_v(7 * (size_t)nb);
    double (*qsave)[7] = reinterpret_cast<double (*)[7]>(qsave_v.data());
    for (int b = 0; b < nb; b++) get_coords(b, qsave[b]);
    double h = 0.0;
    unsigned long ca_guard = 0;
    while (h < dt) {
      g_ca_iters++;
      // a conservative step that no longer advances h (tc below h's ulp) would spin forever in the reference
      if (++ca_guard > MH_CA_HARD_CAP) { aux->status |= MH_WORLD_STALLED; break; }
      broad_phase(dt - h, pairs_to_check);
      calc_pairwise_distances(pairs_to_check, pairwise);
      const double CA = next_CA_step();
      if (CA <= 0.0) break;
      double tc = (sc->min_step_size > CA) ? sc->min_step_size : CA;
      tc = ((dt - h) < tc) ? (dt - h) : tc;
      for (int b = 0; b < nb; b++) {
        set_coords(b, qsave[b]);
        double qd[7]; euler_vel(b, qd);
        double q[7];
        for (int i = 0; i < 7; i++) { q[i] = qd[i] * (h + tc); q[i] = q[i] + qsave[b][i]; }
        set_coords(b, q);
      }
      h += tc;
    }
    // forward dynamics + velocity integration by h (TSS:173-192)
    fwd_dyn_and_integrate(h);
    calc_pairwise_distances(pairs_to_check, pairwise);             // TSS:206
    std::vector<Contact> cs;                                       // find_unilateral_constraints (CSim:488-537)
    for (const PairDist& d : pairwise) if (d.dist < sc->contact_dist_thresh) find_contacts(d.pair, sc->contact_dist_thresh, cs);
    handle_impacts(cs);                                            // TSS:212
    if (thrown_) return h;                                         // (the exception leaves before TSS:215)
    aux->time += h;
    aux->mini_steps++;
    return h;
  }

  // ---- constraint stabilisation (ConstraintStabilization.cpp) ------------------------
  double eval_unilateral(std::vector<double>& uC) {                 // CStab:88-131
    double vio = INF;
    uC.clear();
    calc_pairwise_distances(pairs_to_check, pairwise);
    for (const PairDist& d : pairwise) { uC.push_back(d.dist); vio = (d.dist < vio) ? d.dist : vio; }
    return vio;
  }
  void get_q(std::vector<double>& q) const { q.resize(7 * sc->nb); for (int b = 0; b < sc->nb; b++) get_coords(b, &q[7*b]); }
  void set_q(const std::vector<double>& q) { for (int b = 0; b < sc->nb; b++) set_coords(b, &q[7*b]); }
  double eval_at(double t, unsigned i, const std::vector<double>& dq, const std::vector<double>& q) {   // CStab:1281-1298
    std::vector<double> qs(q.size()), uC;
    for (size_t k = 0; k < q.size(); k++) { qs[k] = dq[k] * t; qs[k] = qs[k] + q[k]; }
    set_q(qs);
    eval_unilateral(uC);
    return uC[i];
  }
  static double sign2(double x, double y) { return (y > 0.0) ? std::fabs(x) : -std::fabs(x); }
  double ridders(double x1, double x2, double fl, double fh, unsigned idx, const std::vector<double>& dq, const std::vector<double>& q) {  // CStab:1322-1379
    const double TOL = 1e-4;
    double ans = INF, fm, fnew, s, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = eval_at(xm, idx, dq, q);
        s = std::sqrt(fm * fm - fl * fh);
        if (s == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s);
        ans = xnew;
        fnew = eval_at(ans, idx, dq, q);
        if (std::fabs(fnew) < TOL && fnew >= 0.0) return xnew;
        if (sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
        else if (sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
        else if (sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
      }
    } else {
      if (fl == 0.0) return x1;
      if (fh == 0.0) return x2;
    }
    return 0.0;
  }
  // evaluate_bilateral_constraints (CStab:133-160): every implicit joint's C, scene order; returns max |C|
  double eval_bilateral(std::vector<double>& C) const {
    C.clear();
    double mx = 0.0;
    for (int j = 0; j < njoints(); j++) {
      double c6[6]; joint_eval(j, c6);
      for (int k = 0; k < joint_rows(sc->joint_type[j]); k++) { C.push_back(c6[k]); const double a = std::fabs(c6[k]); mx = (a > mx) ? a : mx; }
    }
    return mx;
  }
  double eval_bilateral_at(double t, unsigned i, const std::vector<double>& dq, const std::vector<double>& q) {   // CStab:1300-1318
    std::vector<double> qs(q.size()), C;
    for (size_t k = 0; k < q.size(); k++) { qs[k] = dq[k] * t; qs[k] = qs[k] + q[k]; }
    set_q(qs);
    eval_bilateral(C);
    return C[i];
  }
  double ridders_bilateral(double x1, double x2, double fl, double fh, unsigned idx, const std::vector<double>& dq, const std::vector<double>& q) {  // CStab:1382-1439
    const double TOL = 1e-6;
    double ans = INF, fm, fnew, s, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = eval_bilateral_at(xm, idx, dq, q);
        s = std::sqrt(fm * fm - fl * fh);
        if (s == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s);
        ans = xnew;
        fnew = eval_bilateral_at(ans, idx, dq, q);
        if (std::fabs(fnew) < TOL) return ans;
        if (sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
        else if (sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
        else if (sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
      }
    } else {
      if (fl == 0.0) return x1;
      if (fh == 0.0) return x2;
    }
    return 0.0;
  }
  // An island of bodies tied by implicit joints and touched by no unilateral constraint ("remaining island", UC:1158-1191):
  // set_bilateral_only_constraint_data (CStab:531-700: Jfull, get_full_rank_implicit_constraints ICH:1698-1739, the
  // J iM, J iM J' products of compute_X ICH:1657-1660), Jx_v = C on the active rows (CStab:475-486), then determine_dq
  // with an empty LCP (CStab:932-970) and update_from_stacked's bilateral step (ICH:356-374):
  //   (J iM J') lambda = C ,  v = 0 + (0 - iM J' lambda) ,  dq = the bodies' eEuler velocities.
  struct JBlk { int row, off, rows; double w[6][6]; };
  struct BilatData {                       // what compute_problem_data / compute_X keep of an island's implicit joints
    int nbod, ngc, m, k;                   // bodies, coordinates, equations, active (full-rank) equations
    std::vector<int> bodies, act;
    std::vector<JBlk> blocks;
    std::vector<double> Cj, iM, Mg, JiM, A, L, lam;   // C rows; 6x6 blocks of iM and M; J iM (m x ngc); J iM J' on the active rows (col-major k x k) and its factor
  };
  // Jfull, get_full_rank_implicit_constraints (ICH:1698-1739), J iM, J iM J' (ICH:1657-1660), Jx_v = C (CStab:442-453, 475-486)
  bool build_bilateral(const std::vector<int>& island, const std::vector<int>& joints, const std::vector<double>& Call, BilatData& bd) const {
    const int nbod = (int)island.size(), ngc = 6 * nbod;
    int m = 0;
    for (int j : joints) m += joint_rows(sc->joint_type[j]);
    if (nbod > MH_IJOINT_MAX_BODIES || (int)joints.size() > MH_IJOINT_MAX_JOINTS || m > MH_IJOINT_MAX_EQNS) return false;
    bd.nbod = nbod; bd.ngc = ngc; bd.m = m; bd.bodies = island; bd.blocks.clear(); bd.Cj.assign(m, 0.0);
    auto gc_of_body = [&](int b) { for (int i = 0; i < nbod; i++) if (island[i] == b) return 6 * i; return -1; };
    int eq = 0;
    for (int j : joints) {
      const int rows = joint_rows(sc->joint_type[j]);
      int first = 0;                                               // the joint's first row in the scene-wide C vector
      for (int jj = 0; jj < j; jj++) first += joint_rows(sc->joint_type[jj]);
      for (int k = 0; k < rows; k++) bd.Cj[eq + k] = Call[first + k];
      const int sides[2] = { sc->joint_inboard[j], sc->joint_outboard[j] };
      for (int sd = 0; sd < 2; sd++) {
        if (!enabled(sides[sd])) continue;
        JBlk k; k.row = eq; k.off = gc_of_body(sides[sd]); k.rows = rows;
        joint_jac(j, sd == 0, k.w);
        bd.blocks.push_back(k);
      }
      eq += rows;
    }
    const std::vector<JBlk>& blocks = bd.blocks;
    auto covering = [&](int row, std::vector<const JBlk*>& out) { out.clear(); for (const JBlk& k : blocks) if (row >= k.row && row < k.row + k.rows) out.push_back(&k); };
    // J J' and the greedy full-rank set on J J' - sqrt(eps) I
    std::vector<double> JJT((size_t)m * m, 0.0);
    std::vector<const JBlk*> br, bc;
    for (int r = 0; r < m; r++) { covering(r, br); for (int c = 0; c < m; c++) { covering(c, bc);
      double tot = 0.0;
      for (const JBlk* kr : br) for (const JBlk* kc : bc) {
        if (kr->off != kc->off) continue;
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + kr->w[r - kr->row][q] * kc->w[c - kc->row][q];
        tot = tot + acc;
      }
      JJT[(size_t)r * m + c] = tot; } }
    std::vector<int>& act = bd.act; act.clear();
    std::vector<double> L;
    for (int i = 0; i < m; i++) {
      if ((int)act.size() == ngc) break;
      act.push_back(i);
      const int k = (int)act.size();
      L.assign((size_t)k * k, 0.0);
      for (int r = 0; r < k; r++) for (int c = 0; c < k; c++) L[r + (size_t)k * c] = JJT[(size_t)act[r] * m + act[c]];
      for (int r = 0; r < k; r++) L[r + (size_t)k * r] = L[r + (size_t)k * r] - NEAR_ZERO;
      if (!chol_factor(k, L.data(), k)) act.pop_back();
    }
    const int k = bd.k = (int)act.size();
    bd.iM.assign((size_t)nbod * 36, 0.0); bd.Mg.assign((size_t)nbod * 36, 0.0);
    for (int i = 0; i < nbod; i++) {
      double im, Ji[9]; inv_inertia(island[i], im, Ji);
      double Jw[9]; inertia_world(island[i], Jw);
      double* B = &bd.iM[(size_t)i * 36]; double* M = &bd.Mg[(size_t)i * 36];
      for (int q = 0; q < 3; q++) { B[7 * q] = im; M[7 * q] = mass_[island[i]]; }
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { B[6 * (3 + r) + 3 + c] = Ji[3 * r + c]; M[6 * (3 + r) + 3 + c] = Jw[3 * r + c]; }
    }
    bd.JiM.assign((size_t)m * ngc, 0.0);
    for (const JBlk& kb : blocks) {
      const double* B = &bd.iM[(size_t)(kb.off / 6) * 36];
      for (int r = 0; r < kb.rows; r++) for (int c = 0; c < 6; c++) {
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + kb.w[r][q] * B[6 * q + c];
        bd.JiM[(size_t)(kb.row + r) * ngc + kb.off + c] = acc;
      }
    }
    bd.A.assign((size_t)k * k, 0.0);
    for (int r = 0; r < k; r++) { covering(act[r], br); for (int c = 0; c < k; c++) {
      double tot = 0.0;
      for (const JBlk* kr : br) {
        double acc = 0.0;
        for (int q = 0; q < 6; q++) acc = acc + kr->w[act[r] - kr->row][q] * bd.JiM[(size_t)act[c] * ngc + kr->off + q];
        tot = tot + acc;
      }
      bd.A[r + (size_t)k * c] = tot; } }
    bd.L = bd.A;
    if (k > 0 && !chol_factor(k, bd.L.data(), k)) return false;
    bd.lam.assign(k, 0.0);
    for (int r = 0; r < k; r++) bd.lam[r] = bd.Cj[act[r]];           // update_from_stacked (ICH:356-367): (J iM J') lambda = Jx_v
    if (k > 0) chol_solve(k, bd.L.data(), k, bd.lam.data());
    return true;
  }
  // iM J' lambda of coordinate g (ICH:370)
  static double bilateral_dv(const BilatData& bd, int g) {
    double acc = 0.0;
    for (int r = 0; r < bd.k; r++) acc = acc + bd.JiM[(size_t)bd.act[r] * bd.ngc + g] * bd.lam[r];
    return acc;
  }
  // An island of bodies tied by implicit joints and touched by no unilateral constraint ("remaining island", UC:1158-1191):
  // set_bilateral_only_constraint_data (CStab:531-700), then determine_dq with an empty LCP (CStab:932-970) and
  // update_from_stacked's bilateral step (ICH:356-374): v = 0 + (0 - iM J' lambda), dq = the bodies' eEuler velocities.
  bool bilateral_only_dq(const std::vector<int>& island, const std::vector<int>& joints, const std::vector<double>& Call, std::vector<double>& dq) {
    BilatData bd;
    if (!build_bilateral(island, joints, Call, bd)) return false;
    for (int i = 0; i < bd.nbod; i++) {
      double dv[6];
      for (int q = 0; q < 6; q++) dv[q] = 0.0 + (0.0 - bilateral_dv(bd, 6 * i + q));
      const int b = island[i];
      setV(b, v3(dv[0], dv[1], dv[2])); setW(b, v3(dv[3], dv[4], dv[5]));
      double qd[7]; euler_vel(b, qd);
      for (int q = 0; q < 7; q++) dq[7 * b + q] = qd[q];
    }
    return true;
  }
  // ImpactConstraintHandler::compute_X (ICH:1590-1695): X = iM - 2 G + G' M G with G = iM H' J iM, H' = J' (J iM J')^-1, over the
  // active rows -- the inverse inertia of the island projected on the joints' null space; dense ngc x ngc, row-major.
  // Every product accumulates from 0 over ascending indices (Ravelin's gemm order is not in the tree: parity unpinned).
  void compute_X_general(const BilatData& bd, std::vector<double>& X) const {
    const int ngc = bd.ngc, k = bd.k, nbod = bd.nbod;
    std::vector<double> Ainv = bd.A;
    if (k > 0) inverse_spd(k, Ainv.data(), k);
    // H' = J' Ainv (ngc x k): the blocks' rows in block order
    std::vector<double> HT((size_t)ngc * k, 0.0);
    std::vector<int> pos(bd.m, -1);
    for (int r = 0; r < k; r++) pos[bd.act[r]] = r;
    for (int g = 0; g < ngc; g++) for (int c = 0; c < k; c++) {
      double acc = 0.0;
      for (const JBlk& kb : bd.blocks) {
        if (g < kb.off || g >= kb.off + 6) continue;
        for (int r = 0; r < kb.rows; r++) { const int pr = pos[kb.row + r]; if (pr >= 0) acc = acc + kb.w[r][g - kb.off] * Ainv[pr + (size_t)k * c]; }
      }
      HT[(size_t)g * k + c] = acc;
    }
    std::vector<double> HTJiM((size_t)ngc * ngc), G((size_t)ngc * ngc), MG((size_t)ngc * ngc);
    for (int g = 0; g < ngc; g++) for (int h = 0; h < ngc; h++) {
      double acc = 0.0;
      for (int c = 0; c < k; c++) acc = acc + HT[(size_t)g * k + c] * bd.JiM[(size_t)bd.act[c] * ngc + h];
      HTJiM[(size_t)g * ngc + h] = acc;
    }
    for (int i = 0; i < nbod; i++) for (int r = 0; r < 6; r++) for (int h = 0; h < ngc; h++) {
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + bd.iM[(size_t)i * 36 + 6 * r + q] * HTJiM[(size_t)(6 * i + q) * ngc + h];
      G[(size_t)(6 * i + r) * ngc + h] = acc;
    }
    for (int i = 0; i < nbod; i++) for (int r = 0; r < 6; r++) for (int h = 0; h < ngc; h++) {
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + bd.Mg[(size_t)i * 36 + 6 * r + q] * G[(size_t)(6 * i + q) * ngc + h];
      MG[(size_t)(6 * i + r) * ngc + h] = acc;
    }
    X.assign((size_t)ngc * ngc, 0.0);
    for (int i = 0; i < nbod; i++) for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) X[(size_t)(6 * i + r) * ngc + 6 * i + c] = bd.iM[(size_t)i * 36 + 6 * r + c];
    for (int g = 0; g < ngc; g++) for (int h = 0; h < ngc; h++) {
      double acc = 0.0;
      for (int p2 = 0; p2 < ngc; p2++) acc = acc + G[(size_t)p2 * ngc + g] * MG[(size_t)p2 * ngc + h];
      X[(size_t)g * ngc + h] = (X[(size_t)g * ngc + h] - 2.0 * G[(size_t)g * ngc + h]) + acc;
    }
  }
  // update_q (CStab:1056-1216)
  bool update_q(const std::vector<double>& dq, std::vector<double>& q) {
    std::vector<double> uC, uC_old, qstar(q.size()), C, C_old;
    eval_unilateral(uC_old);
    eval_bilateral(C_old);
    double old_cvio = 0.0;
    for (double c : C_old) old_cvio = old_cvio + c * c;
    old_cvio = std::sqrt(old_cvio);
    for (size_t k = 0; k < q.size(); k++) { qstar[k] = dq[k]; qstar[k] = qstar[k] + q[k]; }
    set_q(qstar);
    eval_unilateral(uC);
    eval_bilateral(C);
    std::vector<char> bbr(C.size(), 0);
    for (size_t i = 0; i < C.size(); i++) bbr[i] = ((C[i] < 0.0 && C_old[i] > 0.0) || (C[i] > 0.0 && C_old[i] < 0.0)) ? 1 : 0;
    std::vector<char> br(uC.size(), 0);
    for (size_t i = 0; i < uC.size(); i++)
      br[i] = ((uC_old[i] < 0.0 && uC[i] > 0.0) || (uC_old[i] > 0.0 && uC[i] < 0.0)) ? 1 : 0;
    double t = 1.0;
    for (size_t i = 0; i < br.size(); i++) {
      if (!br[i]) continue;
      const double root = ridders(0, t, uC_old[i], uC[i], (unsigned)i, dq, q);
      if (root > 0.0 && root < 1.0) t = (root < t) ? root : t;
    }
    for (size_t i = 0; i < bbr.size(); i++) {                       // CStab:1132-1145
      if (!bbr[i]) continue;
      const double root = ridders_bilateral(0, t, C_old[i], C[i], (unsigned)i, dq, q);
      if (root > 0.0 && root < 1.0) t = (root < t) ? root : t;
    }
    for (size_t k = 0; k < q.size(); k++) { qstar[k] = dq[k] * t; qstar[k] = qstar[k] + q[k]; }
    set_q(qstar);
    eval_unilateral(uC);
    eval_bilateral(C);
    const double BETA = 0.6;
    while (true) {
      bool stop = true;
      for (size_t i = 0; i < br.size(); i++) if (!br[i] && uC[i] < 0.0 && uC_old[i] > uC[i]) { stop = false; break; }
      if (stop) {                                                   // CStab:1180-1192 (no joints: cvio = 0 < bilateral_eps)
        double cvio = 0.0;
        for (double c : C) cvio = cvio + c * c;
        cvio = std::sqrt(cvio);
        if (cvio < BILATERAL_EPS || cvio < old_cvio) break;
      }
      t *= BETA;
      if (t < NEAR_ZERO) return false;
      for (size_t k = 0; k < q.size(); k++) { qstar[k] = dq[k] * t; qstar[k] = qstar[k] + q[k]; }
      set_q(qstar);
      eval_bilateral(C);
      eval_unilateral(uC);
    }
    q = qstar;
    return true;
  }
  // ConstraintStabilization::stabilize (CStab:167-254)
  void stabilize() {
    if (sc->cstab_max_iterations == 0) return;
    const int nb = sc->nb;
    // the bodies of the jointed islands (Simulator::find_islands' groups that hold a joint): static per scene
    std::vector<std::vector<int> > jisl; std::vector<std::vector<int> > jisl_joints; std::vector<char> jointed(nb, 0);
    if (njoints() > 0) {
      std::vector<std::vector<int> > all; find_body_islands(all);
      for (const std::vector<int>& isl : all) {
        std::vector<int> ij;
        for (int j = 0; j < njoints(); j++) {
          const int a = sc->joint_inboard[j], b = sc->joint_outboard[j];
          if ((enabled(a) && std::binary_search(isl.begin(), isl.end(), a)) || (enabled(b) && std::binary_search(isl.begin(), isl.end(), b))) ij.push_back(j);
        }
        if (ij.empty()) continue;
        jisl.push_back(isl); jisl_joints.push_back(ij);
        for (int b : isl) jointed[b] = 1;
      }
    }
    std::vector<double> Cb;
    double max_bvio = eval_bilateral(Cb);
    std::vector<double> vsave_v(6 * (size_t)nb);
    double (*vsave)[6] = reinterpret_cast<double (*)[6]>(vsave_v.data());
    for (int b = 0; b < nb; b++) for (int k = 0; k < 6; k++) vsave[b][k] = st[13*b + 7 + k];
    std::vector<double> q; get_q(q);
    std::vector<double> uC;
    double max_uvio = eval_unilateral(uC);
    unsigned iterations = 0;
    while (max_uvio < sc->cstab_eps || max_bvio > BILATERAL_EPS) {
      if (iterations == sc->cstab_max_iterations) break;
      // the reference's default cap is UINT_MAX: a cycling stabiliser would never return; stop and say so
      if (iterations == MH_CSTAB_HARD_CAP) { aux->status |= MH_WORLD_STALLED; break; }
      for (int b = 0; b < nb; b++) for (int k = 0; k < 6; k++) st[13*b + 7 + k] = 0.0;
      // compute_problem_data (CStab:347-492): own broad phase with dt = 0, one contact per pair
      std::vector<int> cpairs; broad_phase(0.0, cpairs);
      std::vector<Contact> cs;
      for (int p : cpairs) {
        const PairDist d = signed_dist(p);
        if (d.dist >= NEAR_ZERO) {                                 // separated: synthetic contact (CStab:316-331)
          Contact c; c.pair = p; c.g1 = d.a; c.g2 = d.b; c.p = d.pa;
          const V3 nn = d.pb - d.pa;
          c.n = nn / norm(nn);
          c.dist = d.dist;
          orthonormal_basis(c.n, c.s, c.t); fill_params(c);
          cs.push_back(c);
        } else find_contacts(p, NEAR_ZERO, cs);                    // CStab:337
      }
      std::vector<Island> islands; find_islands(cs, islands);
      std::vector<double> dq(q.size(), 0.0);
      // islands tied by joints and free of unilateral constraints (UC:1158-1191): the bilateral step alone.  A jointed
      // island that the stabiliser's contact list touches would need compute_X's general case (ICH:1590-1695): not built,
      // the world is flagged and neither that contact island nor the jointed island moves.
      for (size_t ji = 0; ji < jisl.size(); ji++) {
        bool touched = false;
        for (const Contact& c : cs) if ((enabled(c.g1) && std::binary_search(jisl[ji].begin(), jisl[ji].end(), c.g1)) ||
                                        (enabled(c.g2) && std::binary_search(jisl[ji].begin(), jisl[ji].end(), c.g2))) { touched = true; break; }
        if (touched) continue;                                      // part of a contact island: below
        if (!bilateral_only_dq(jisl[ji], jisl_joints[ji], Cb, dq)) aux->status |= MH_WORLD_STAB_FAILED;
      }
      for (const Island& isl : islands) {
        bool mixed = false;
        for (int b : isl.bodies) if (jointed[b]) mixed = true;
        ProblemData pd;
        BilatData bd;
        if (mixed) {
          // set_unilateral_constraint_data with implicit joints (CStab:705-904): the island's joints, their full-rank rows,
          // compute_X's general case; Jx_v = C on the active rows (CStab:442-453)
          std::vector<int> bodies = isl.bodies;
          std::sort(bodies.begin(), bodies.end());
          bodies.erase(std::unique(bodies.begin(), bodies.end()), bodies.end());
          std::vector<int> ij;
          for (int j = 0; j < njoints(); j++) {
            const int a = sc->joint_inboard[j], b = sc->joint_outboard[j];
            if ((enabled(a) && std::binary_search(bodies.begin(), bodies.end(), a)) || (enabled(b) && std::binary_search(bodies.begin(), bodies.end(), b))) ij.push_back(j);
          }
          if (!build_bilateral(bodies, ij, Cb, bd)) { aux->status |= MH_WORLD_UNSUPPORTED; continue; }
          std::vector<double> Xg; compute_X_general(bd, Xg);
          compute_problem_data(cs, isl, pd, true, &Xg);
        } else
        compute_problem_data(cs, isl, pd, true);
        const int nc = pd.nc;
        for (int i = 0; i < nc; i++) pd.Cv[0][i] = pd.c[i]->dist - std::fabs(sc->cstab_eps) - NEAR_ZERO;   // CStab:431
        // determine_dq (CStab:932-970): MM = Cn X Cn', cold lcp_fast then Lemke ladder
        if (nc > lcp_cap_) { aux->status |= MH_WORLD_UNSUPPORTED; continue; }
        std::vector<double> MM((size_t)nc * nc);
        for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) MM[(size_t)i + (size_t)nc * j] = pd.G[0][0][(size_t)i * nc + j];
        Vec z;                                                       // fresh local: size 0 -> cold start
        oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
        LCP lcp; lcp.rng = &rs;
        Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? std::max(0, trace_cap - trace_len) : 0;
        lcp.trace = &tr;
        unsigned piv = 0;
        bool ok = lcp.lcp_fast(nc, MM.data(), nc, pd.Cv[0].data(), z, -1.0);
        piv += lcp.pivots;
        if (!ok) { ok = lcp.lcp_lemke_regularized(nc, MM.data(), nc, pd.Cv[0].data(), z); piv += lcp.pivots; }
        trace_len += tr.len;
        std::memcpy(aux->rng, &rs, sizeof(rs));
        lcp_account(nc, piv); aux->stab_rows += (unsigned long long)nc;
        // update_from_stacked(pd, z): cn = z (whatever z holds, even after a failed solve)
        for (int i = 0; i < nc; i++) pd.cn[i] = (i < (int)z.size()) ? z[i] : 0.0;
        pd.XJ[1].clear(); pd.XJ[2].clear();
        apply_impulses(pd);
        if (mixed) for (size_t bi = 0; bi < pd.bodies.size(); bi++) {    // dv -= iM J' lambda (ICH:370)
          const int b = pd.bodies[bi];
          double s6[6];
          for (int q = 0; q < 6; q++) s6[q] = st[13 * b + 7 + q] - bilateral_dv(bd, 6 * (int)bi + q);
          for (int q = 0; q < 6; q++) st[13 * b + 7 + q] = s6[q];
        }
        for (int b : pd.bodies) { double qd[7]; euler_vel(b, qd); for (int k = 0; k < 7; k++) dq[7*b + k] = qd[k]; }
      }
      if (!update_q(dq, q)) { aux->status |= MH_WORLD_STAB_FAILED; break; }
      max_uvio = eval_unilateral(uC);
      max_bvio = eval_bilateral(Cb);
      iterations++;
      aux->stab_iters++;
    }
    for (int b = 0; b < nb; b++) for (int k = 0; k < 6; k++) st[13*b + 7 + k] = vsave[b][k];
  }

  // TimeSteppingSimulator::step (TSS:52-111)
  void step(double dt) {
    if (aux->status & MH_WORLD_LCP_FAILED) return;                 // the run ended with an exception of the impact handler (see handle_impacts)
    broad_phase(dt, pairs_to_check);
    calc_pairwise_distances(pairs_to_check, pairwise);
    double h = 0.0;
    unsigned guard = 0;
    while (h < dt) {
      h += do_mini_step(dt - h);
      if (thrown_) return;
      if (++guard > 100000u) { aux->status |= MH_WORLD_STALLED; break; }   // the reference would spin forever
    }
    stabilize();
    aux->steps++;
  }
};

} // namespace oracle
#endif
