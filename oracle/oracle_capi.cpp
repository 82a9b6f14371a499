// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// extern "C" surface of the CPU oracle so tests/ and bench.py's cpu_baseline
// leg can drive it through ctypes.  Nothing under moby_amd/ may link this.
#include <cstring>
#include <ctime>
#include "lcp.hpp"

using namespace oracle;

extern "C" {

void oracle_srand_state(uint32_t* state32, uint32_t seed)
{
  oracle_rand_t s; oracle_srand(&s, seed);
  std::memcpy(state32, &s, sizeof(s));
}

int oracle_rand_next(uint32_t* state32)
{
  oracle_rand_t s; std::memcpy(&s, state32, sizeof(s));
  int v = oracle_rand(&s);
  std::memcpy(state32, &s, sizeof(s));
  return v;
}

// dgesv-semantics solve (oracle/linalg.hpp); returns LAPACK info
int oracle_lu_solve(int n, double* A, int ld, double* b) { return lu_solve(n, A, ld, b); }

// kind: 0 lcp_fast, 1 lcp_fast_regularized, 2 lcp_lemke, 3 lcp_lemke_regularized
// z: buffer of at least max(2n, *z_size) doubles; *z_size is z.size() on entry
// (== n requests lcp_fast's warm start) and on exit.
// rng: 32 uint32 words (oracle_rand_t), in/out.
// returns 1 = solver returned true, 0 = false
int oracle_lcp_solve(int kind, int n, const double* M, int ld, const double* q,
                     double* z, int* z_size,
                     int min_exp, unsigned step_exp, int max_exp,
                     double piv_tol, double zero_tol,
                     uint32_t* rng, unsigned* pivots,
                     int32_t* trace, int trace_cap, int* trace_len)
{
  oracle_rand_t rs; std::memcpy(&rs, rng, sizeof(rs));
  LCP lcp; lcp.rng = &rs;
  Trace tr; tr.buf = trace; tr.cap = trace_cap;
  lcp.trace = &tr;
  Vec zz;
  const unsigned cap = std::max<unsigned>(2u * (unsigned)n, (unsigned)*z_size);
  zz.d.assign(z, z + cap); zz.len = (unsigned)*z_size;
  bool ok = false;
  switch (kind) {
    case 0: ok = lcp.lcp_fast(n, M, ld, q, zz, zero_tol); break;
    case 1: ok = lcp.lcp_fast_regularized(n, M, ld, q, zz, min_exp, step_exp, max_exp, piv_tol, zero_tol); break;
    case 2: ok = lcp.lcp_lemke(n, M, ld, q, zz, piv_tol, zero_tol); break;
    case 3: ok = lcp.lcp_lemke_regularized(n, M, ld, q, zz, min_exp, step_exp, max_exp, piv_tol, zero_tol); break;
    default: return -1;
  }
  const unsigned ncopy = std::min<unsigned>(cap, (unsigned)zz.d.size());
  std::memcpy(z, zz.d.data(), sizeof(double) * ncopy);
  *z_size = (int)zz.len;
  if (pivots) *pivots = lcp.pivots;
  if (trace_len) *trace_len = tr.len;
  std::memcpy(rng, &rs, sizeof(rs));
  return ok ? 1 : 0;
}

// Batch driver used for the CPU baseline: solves B independent problems
// (problem b at M + b*strideM, q + b*n, z + b*2n, rng + b*32) sequentially on
// one thread and returns elapsed CPU-wall seconds.
double oracle_lcp_solve_batch(int kind, int B, int n, const double* M, int ld, long strideM,
                              const double* q, double* z, int* z_size,
                              int min_exp, unsigned step_exp, int max_exp,
                              double piv_tol, double zero_tol, uint32_t* rng,
                              int* status, unsigned* pivots)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int b = 0; b < B; b++) {
    int zs = z_size ? z_size[b] : n;
    unsigned piv = 0;
    int st = oracle_lcp_solve(kind, n, M + (size_t)b*strideM, ld, q + (size_t)b*n, z + (size_t)b*2*n, &zs,
                              min_exp, step_exp, max_exp, piv_tol, zero_tol, rng + (size_t)b*32, &piv,
                              nullptr, 0, nullptr);
    if (z_size) z_size[b] = zs;
    if (status) status[b] = st;
    if (pivots) pivots[b] = piv;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

} // extern "C"

// ---------------------------------------------------------------------------
// world stepper (oracle/world.hpp)
#include "world.hpp"

extern "C" {

void oracle_world_aux_init(mh_world_aux* a, uint32_t seed)
{
  std::memset(a, 0, sizeof(*a));
  oracle_srand_state(a->rng, seed);
}

// Steps ONE world nsteps times; traj (nsteps x nb x 7) and trace optional.
// Returns elapsed seconds.
double oracle_world_step(const mh_scene* sc, double dt, int nsteps, double* state, mh_world_aux* aux,
                         double* traj, int32_t* trace, int trace_cap, int* trace_len)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  World w(sc, state, aux);
  w.trace = trace; w.trace_cap = trace_cap;
  for (int s = 0; s < nsteps; s++) {
    w.step(dt);
    if (traj) for (int b = 0; b < sc->nb; b++) for (int k = 0; k < 7; k++) traj[((size_t)s * sc->nb + b) * 7 + k] = state[13*b + k];
  }
  if (trace_len) *trace_len = w.trace_len;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

unsigned long long oracle_dbg_ca_iters(void) { return g_ca_iters; }
int oracle_dbg_lemke_exit(void) { return g_lemke_exit; }
// lcp_fast's iteration statistics (lcp.hpp, g_fast_stats): on != 0 switches the counting on and clears the counters; out (13 values) may be null
void oracle_dbg_fast_repeats(int on, unsigned long long* out) { if (out) for (int i = 0; i < 13; i++) out[i] = g_fast_stats[i]; g_fast_diag = on; if (on) for (int i = 0; i < 13; i++) g_fast_stats[i] = 0; }
// diagnostic: write every impact LCP solve_impact_lcp sees (inputs, rand() state, pivot counts) to `path`; NULL stops
void oracle_dbg_lcp_dump(const char* path) { if (g_lcp_dump) { std::fclose(g_lcp_dump); g_lcp_dump = nullptr; } if (path) g_lcp_dump = std::fopen(path, "wb"); }
// the model of the device's structure-exploiting LU (compact_lu.hpp): nb = panel width of the check (0 = off)
void oracle_dbg_lemke_compact(int nb) { g_lemke_compact = nb; }
void oracle_dbg_lu_fma(int on) { g_lu_fma = on; }     // EXPERIMENT: fused multiply-subtracts in dgesv for LCPs of more than 64 rows (linalg.hpp)
void oracle_dbg_compact_check(int nb) { g_compact_check = nb; for (auto& v : g_compact_stats) v = 0; }
void oracle_dbg_compact_stats(unsigned long long* out) { for (int i = 0; i < 8; i++) out[i] = g_compact_stats[i]; }
// kind / idx: n entries (CL_UNIT: -e_idx, CL_DENSE: column idx of dense (n x *, ld)); b in/out; returns info or CL_FALLBACK
int oracle_lu_solve_compact(int n, const int* kind, const int* idx, const double* dense, int ld, double* b, int nb, int* rows_out)
{ return lu_solve_compact(n, kind, idx, dense, ld, b, nb, rows_out, nullptr); }
void oracle_dbg_lu_hist(unsigned long long* out) { for (int i = 0; i < 130; i++) out[i] = g_lu_hist[i]; }

// B worlds sequentially on one thread (CPU baseline); returns elapsed seconds
double oracle_world_step_batch(const mh_scene* sc, int B, double dt, int nsteps, double* state, mh_world_aux* aux)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int b = 0; b < B; b++) {
    World w(sc, state + (size_t)b * sc->nb * 13, aux + b);
    for (int s = 0; s < nsteps; s++) w.step(dt);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

// finds the contacts of the current state and handles impacts in place, without integrating
// (what ConstraintSimulator::find_unilateral_constraints + calc_impacting_unilateral_constraint_forces
// do at the end of a mini-step, TSS:206-212); used to start a run from a post-impact state
void oracle_world_handle_impacts(const mh_scene* sc, double* state, mh_world_aux* aux)
{
  World w(sc, state, aux);
  std::vector<int> pairs; w.broad_phase(0.0, pairs);
  std::vector<PairDist> pd; w.calc_pairwise_distances(pairs, pd);
  std::vector<Contact> cs;
  for (const PairDist& d : pd) if (d.dist < sc->contact_dist_thresh) w.find_contacts(d.pair, sc->contact_dist_thresh, cs);
  w.handle_impacts(cs);
}

// the impact LCP (_MM column-major n x n, _qq) the handler would assemble for the
// current state's contacts (first active island), for cross-checks; returns n
int oracle_world_impact_lcp(const mh_scene* sc, double* state, mh_world_aux* aux, double* MM, double* qq, int cap)
{
  World w(sc, state, aux);
  std::vector<int> pairs; w.broad_phase(0.0, pairs);
  std::vector<PairDist> pd; w.calc_pairwise_distances(pairs, pd);
  std::vector<Contact> cs;
  for (const PairDist& d : pd) if (d.dist < sc->contact_dist_thresh) w.find_contacts(d.pair, sc->contact_dist_thresh, cs);
  std::vector<World::Island> isl; w.find_islands(cs, isl);
  if (isl.empty()) return 0;
  World::ProblemData p; w.compute_problem_data(cs, isl[0], p, false);
  std::vector<double> M, q; int n;
  if (w.impact_model == MH_IMPACT_MODEL_AP) w.build_ap_lcp(p, M, q, n); else w.build_impact_lcp(p, M, q, n);
  if (n > cap) return -n;
  std::memcpy(MM, M.data(), sizeof(double) * (size_t)n * n);
  std::memcpy(qq, q.data(), sizeof(double) * n);
  return n;
}

unsigned long long oracle_lu_nan_pivots() { return g_lu_nan_pivots; }

// the impact model of every World built from now on (MH_IMPACT_MODEL_DS / _AP: the reference's USE_AP build option)
void oracle_set_impact_model(int model) { g_impact_model = model; }

// ImpactConstraintHandler::process_constraints (ICH:75-168) on an explicit contact list: the checker of
// mh_impact_batch_process (include/moby_hip_impact.h).  Bodies / buffers of any size; aux carries the rand()
// stream, status bits, counters and the _z / _zlast sizes.  order_out (nc ints or NULL) receives the island
// order of the contacts (position in the LCP -> caller index) of the first active island.
void oracle_impact_process(int nb, int nc, const double* mass, const double* inertia, double* state,
                           const mh_contact* contacts, double* impulses, mh_world_aux* aux,
                           double* zlast, double* zbuf, int lcp_cap, int* order_out)
{
  World w(nb, mass, reinterpret_cast<const double (*)[3]>(inertia), state, aux, zlast, zbuf, lcp_cap);
  std::vector<Contact> cs((size_t)nc);
  for (int i = 0; i < nc; i++) {
    const mh_contact& m = contacts[i]; Contact& c = cs[i];
    c.g1 = (m.body1 >= 0 && m.body1 < nb) ? m.body1 : nb; c.g2 = (m.body2 >= 0 && m.body2 < nb) ? m.body2 : nb;
    c.pair = 0; c.dist = 0.0;
    c.p = v3(m.point[0], m.point[1], m.point[2]); c.n = v3(m.normal[0], m.normal[1], m.normal[2]);
    World::orthonormal_basis(c.n, c.s, c.t);
    c.mu = m.mu_coulomb; c.muv = m.mu_viscous; c.eps = m.epsilon; c.compliance = m.compliance; c.nk = m.nk;
  }
  if (order_out) {
    std::vector<World::Island> isl; w.find_islands(cs, isl);
    for (int i = 0; i < nc; i++) order_out[i] = -1;
    if (!isl.empty()) for (size_t k = 0; k < isl[0].contacts.size() && (int)k < nc; k++) order_out[k] = isl[0].contacts[k];
  }
  w.handle_impacts(cs);
  if (impulses) for (int i = 0; i < nc; i++) for (int d = 0; d < 3; d++) impulses[3 * i + d] = cs[i].imp[d];
}

// _MM / _qq of the island the contact list forms (first island), column-major; returns n or -n if n > cap
int oracle_impact_lcp(int nb, int nc, const double* mass, const double* inertia, double* state,
                      const mh_contact* contacts, double* MM, double* qq, int cap)
{
  mh_world_aux aux; std::memset(&aux, 0, sizeof(aux));
  World w(nb, mass, reinterpret_cast<const double (*)[3]>(inertia), state, &aux, nullptr, nullptr, cap);
  std::vector<Contact> cs((size_t)nc);
  for (int i = 0; i < nc; i++) {
    const mh_contact& m = contacts[i]; Contact& c = cs[i];
    c.g1 = (m.body1 >= 0 && m.body1 < nb) ? m.body1 : nb; c.g2 = (m.body2 >= 0 && m.body2 < nb) ? m.body2 : nb;
    c.pair = 0; c.dist = 0.0;
    c.p = v3(m.point[0], m.point[1], m.point[2]); c.n = v3(m.normal[0], m.normal[1], m.normal[2]);
    World::orthonormal_basis(c.n, c.s, c.t);
    c.mu = m.mu_coulomb; c.muv = m.mu_viscous; c.eps = m.epsilon; c.compliance = m.compliance; c.nk = m.nk;
  }
  std::vector<World::Island> isl; w.find_islands(cs, isl);
  if (isl.empty()) return 0;
  World::ProblemData p; w.compute_problem_data(cs, isl[0], p, false);
  std::vector<double> M, q; int n;
  if (w.impact_model == MH_IMPACT_MODEL_AP) w.build_ap_lcp(p, M, q, n); else w.build_impact_lcp(p, M, q, n);
  if (n > cap) return -n;
  std::memcpy(MM, M.data(), sizeof(double) * (size_t)n * n);
  std::memcpy(qq, q.data(), sizeof(double) * n);
  return n;
}


// ---- big scenes (include/moby_hip_stack.h) ----------------------------------------------------------
static SceneView view_of(const mh_big_scene* s, const std::vector<int>& enabled_all)
{
  return SceneView{ s->nb, s->has_ground, s->geom_type, reinterpret_cast<const double (*)[3]>(s->geom_dim), s->mass,
                    reinterpret_cast<const double (*)[3]>(s->inertia), s->plane_R, s->plane_o, s->gravity,
                    s->npairs, s->pair_a, s->pair_b, s->pair_model, enabled_all.data(),
                    s->cp_epsilon, s->cp_mu_coulomb, s->cp_mu_viscous, s->cp_compliance, nullptr,
                    s->min_step_size, s->contact_dist_thresh, s->cstab_eps, s->cstab_max_iterations,
                    s->njoints, s->joint_type, s->joint_inboard, s->joint_outboard,
                    reinterpret_cast<const double (*)[3]>(s->joint_anchor_in), reinterpret_cast<const double (*)[3]>(s->joint_anchor_out),
                    reinterpret_cast<const double (*)[9]>(s->joint_vec_in), reinterpret_cast<const double (*)[9]>(s->joint_vec_out) };
}

// implicit joint j of a big scene at `state`: its constraint values C (6) and the two Jacobian blocks (6 x 6 each, row-major,
// inboard then outboard) -- the tests differentiate one against the other
void oracle_joint_eval(const mh_big_scene* s, double* state, int j, double* C, double* Cq_in, double* Cq_out)
{
  std::vector<int> en((size_t)s->npairs, 1);
  SceneView v = view_of(s, en);
  mh_world_aux aux; std::memset(&aux, 0, sizeof(aux));
  World w(v, state, &aux, nullptr, nullptr, 0);
  for (int k = 0; k < 6; k++) C[k] = 0.0;
  w.joint_eval(j, C);
  double A[6][6], B[6][6];
  std::memset(A, 0, sizeof(A)); std::memset(B, 0, sizeof(B));
  w.joint_jac(j, true, A); w.joint_jac(j, false, B);
  std::memcpy(Cq_in, A, sizeof(A)); std::memcpy(Cq_out, B, sizeof(B));
}

// nsteps x TimeSteppingSimulator::step (mode 0) or one ConstraintStabilization::stabilize (mode 1) of ONE big world.
// zlast / zbuf: the handler's _zlast / _z storage (lcp_cap doubles each); their sizes live in aux (zlast_size, zbuf_size,
// zbuf_cap).  Returns elapsed seconds.
double oracle_big_step(const mh_big_scene* s, double dt, int nsteps, double* state, mh_world_aux* aux,
                       double* zlast, double* zbuf, int lcp_cap, int mode)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  std::vector<int> en((size_t)s->npairs, 1), nk((size_t)s->npairs, s->nk);
  SceneView v = view_of(s, en);
  v.cp_nk = nk.data();
  World w(v, state, aux, zlast, zbuf, lcp_cap);
  w.impact_model = s->impact_model;
  if (mode == 1) {
    w.broad_phase(0.0, w.pairs_to_check);              // the simulator's pair list of the step that just ended
    w.calc_pairwise_distances(w.pairs_to_check, w.pairwise);
    w.stabilize();
  } else for (int k = 0; k < nsteps; k++) w.step(dt);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}


} // extern "C"

// ---- articulated bodies (include/moby_hip_artic.h) ----------------------------------------------------
#include "artic.hpp"

extern "C" {

// B worlds x nsteps of TimeSteppingSimulator::step, sequentially on one thread; returns elapsed seconds
double oracle_artic_step(const mh_artic_model* m, int B, double dt, int nsteps, double* q, double* qd, mh_world_aux* aux)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int b = 0; b < B; b++) {
    Artic w(m, q + (size_t)b * m->nj, qd + (size_t)b * m->nj, aux + b);
    for (int s = 0; s < nsteps; s++) w.step(dt);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

// the same through do_mini_step / handle_impacts even without spheres (tests: must agree bit for bit with oracle_artic_step)
double oracle_artic_step_general(const mh_artic_model* m, int B, double dt, int nsteps, double* q, double* qd, mh_world_aux* aux)
{
  for (int b = 0; b < B; b++) {
    Artic w(m, q + (size_t)b * m->nj, qd + (size_t)b * m->nj, aux + b);
    w.force_general = true;
    for (int s = 0; s < nsteps; s++) w.step(dt);
  }
  return 0.0;
}

// qdd = H^-1 (tau - C) of one state; H (nj x nj, row-major), C (nj) and the link poses (nj x 12) optional.  Returns 1 / 0 (H not PD).
int oracle_artic_fwd_dyn(const mh_artic_model* m, const double* q, const double* qd, const double* tau, double* qdd, double* H, double* C, double* poses)
{
  mh_world_aux aux; std::memset(&aux, 0, sizeof(aux));
  std::vector<double> qq(q, q + m->nj), qv(qd, qd + m->nj);
  Artic w(m, qq.data(), qv.data(), &aux);
  bool ok;
  if (m->algorithm == MH_ARTIC_FSAB) { ok = w.fwd_dyn_aba(tau, qdd); w.crba(); w.bias(); }   // H / C as get_generalized_inertia would give them
  else ok = w.fwd_dyn(tau, qdd);
  if (H) for (int e = 0; e < m->nj * m->nj; e++) H[e] = w.H[e];
  if (C) for (int i = 0; i < m->nj; i++) C[i] = w.C[i];
  if (poses) for (int i = 0; i < m->nj; i++) { for (int k = 0; k < 9; k++) poses[12 * i + k] = w.R[i][k]; for (int k = 0; k < 3; k++) poses[12 * i + 9 + k] = w.x[i][k]; }
  return ok ? 1 : 0;
}

void oracle_artic_jacobian(const mh_artic_model* m, const double* q, int link, const double* p, double* J)
{
  mh_world_aux aux; std::memset(&aux, 0, sizeof(aux));
  std::vector<double> qq(q, q + m->nj), qv(m->nj, 0.0);
  Artic w(m, qq.data(), qv.data(), &aux);
  w.kinematics();
  w.jacobian(link, p, J);
}

void oracle_sincos(double x, double* s, double* c) { sincos_kernel(x, *s, *c); }

} // extern "C"
