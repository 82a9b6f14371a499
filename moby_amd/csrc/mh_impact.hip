// mh_impact.inc -- batched ImpactConstraintHandler::process_constraints on explicit contact lists
// (include/moby_hip_impact.h; seam B2 of SURVEY.md 8b).
//
// One world = one island of up to IMP_MAXC contacts over up to IMP_MAXB free bodies.  Everything a world
// needs lives in HBM (the LCP matrix alone is n^2 doubles: 33.5 MB at n = 2048); the pipeline of one call is
//   mh_k_imp_prep    1 workgroup / world   impacting? one island? island order (the reference's BFS), X blocks,
//                                          Jacobian rows [d, r x d] and X C^T rows per contact side, C v
//   mh_k_imp_gram    1 thread / (i, j)     the six C_a X C_b^T blocks (ICH:2125-2149)
//   mh_k_imp_mm      1 workgroup / column  _MM = [H, -M^T; M, 0], _qq (ICH-QP:271-497), warm start z
//   LCP entry        lcp_fast_regularized(-20, 4, -8), then for the worlds it failed on z = 0 +
//                    lcp_lemke_regularized (ICH-QP:219-224) -- the block solver above n = 64
//   mh_k_imp_post    1 workgroup / world   update_from_stacked, update_constraint_velocities_from_impulses,
//                                          apply_restitution, the second-solve test (ICH:569-600), impact tolerance
// and, only when some contact has epsilon > 0, a second round (mm with new _qq, LCP, post) for the worlds that asked.
// Arithmetic order follows oracle/world.hpp (compute_problem_data, build_impact_lcp, apply_impulses,
// update_constraint_vels), which restates the reference's SparseJacobian products.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/moby_hip_impact.h"
#include "mh_host.h"
#define MH_DEV __device__ __forceinline__

namespace mh { namespace imp {

constexpr int T = 256;
constexpr int MAXC = 512;      // contacts per world (n = 6 nc + nc nk/2 <= 4096 with nk >= 4)
constexpr int MAXB = 256;      // bodies per world
constexpr double NEAR_ZERO_ = 1.4901161193847656e-08;   // Constants.h:21

struct P3 { double x, y, z; };
MH_DEV P3 p3(double x, double y, double z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
MH_DEV P3 operator+(P3 a, P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
MH_DEV P3 operator-(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
MH_DEV P3 operator-(P3 a) { return p3(-a.x, -a.y, -a.z); }
MH_DEV P3 operator/(P3 a, double s) { return p3(a.x / s, a.y / s, a.z / s); }
MH_DEV double dot3(P3 a, P3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MH_DEV P3 cross3(P3 a, P3 b) { return p3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

struct Dev {
  int B, nb, nc, nk, kh, n;
  const double* mass; const double* inertia;      // nb, nb x 3
  double* state; const mh_contact* contacts;      // B x nb x 13, B x nc
  int* order;        // B x nc      island position -> caller index
  int* cbody;        // B x nc x 2  body of side 0 / 1 in island order, -1 = static
  double* cpar;      // B x nc x 4  mu, mu_viscous, epsilon, compliance (island order)
  double* W;         // B x nc x 36 Jacobian rows [dir][side][6]
  double* XJ;        // B x nc x 36 (X C^T) rows   [dir][side][6]
  double* Cv;        // B x 3 x nc  Cn v, Cs v, Ct v
  double* G;         // B x 6 x nc x nc   nn ns nt ss st tt, row-major
  double* MM; double* qq; double* z; double* zlast; int* zlast_size; int* zsz;
  uint32_t* rng;
  int* need; int* need2; int* again; int* lst1; int* lst2; unsigned* piv1; unsigned* piv2;   // again: second solve asked for (ICH:591)
  int* status; unsigned* pivots; int* solves; double* imp;   // imp: B x nc x 3, caller order
  const double* fcos; const double* fsin;                    // kh each (host libm)
};

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
MH_DEV void inv_inertia(const double* st, const double* J, double m, double* out /*10*/) {
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

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(T)
void mh_k_imp_prep(Dev d)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const int nc = d.nc, nb = d.nb;
  __shared__ int s_b1[MAXC], s_b2[MAXC], s_order[MAXC];
  __shared__ unsigned char s_done[MAXC], s_node[MAXB], s_queued[MAXB];
  __shared__ int s_queue[MAXB];
  __shared__ double s_xinv[MAXB * 10];
  __shared__ int s_flag[5];   // 0: some contact impacting, 1: some mu < 100, 2: single island, 3: need, 4: malformed contact
  const mh_contact* C = d.contacts + (size_t)b * nc;
  const double* st = d.state + (size_t)b * nb * 13;
  if (t < 5) s_flag[t] = 0;
  for (int i = t; i < nb; i += T) { s_node[i] = 0; s_queued[i] = 0; }
  __syncthreads();
  for (int i = t; i < nc; i += T) {
    int g1 = C[i].body1, g2 = C[i].body2;
    if (g1 < 0 || g1 >= nb) g1 = -1;
    if (g2 < 0 || g2 >= nb) g2 = -1;
    s_b1[i] = g1; s_b2[i] = g2; s_done[i] = 0;
    if (g1 >= 0) s_node[g1] = 1;
    if (g2 >= 0) s_node[g2] = 1;
    const P3 p = ld3(C[i].point), n = ld3(C[i].normal);
    const double vn = dot3(n, point_vel(st, g1, p) - point_vel(st, g2, p));     // calc_constraint_vel (UC:695-747)
    if (vn < -NEAR_ZERO_) s_flag[0] = 1;
    if (C[i].mu_coulomb < 1e2) s_flag[1] = 1;                                    // ICH:127
    // contacts may have been written through mh_impact_batch_device_ptrs, past upload()'s host checks: the same
    // checks here (nk of the batch, unit normal, two different bodies of which one is dynamic)
    const double nn = dot3(n, n);
    if (C[i].nk != d.nk || !(nn > 0.25 && nn < 4.0) || g1 == g2) s_flag[4] = 1;
  }
  for (int i = t; i < nb; i += T) inv_inertia(st + 13 * i, d.inertia + 3 * i, d.mass[i], s_xinv + 10 * i);
  __syncthreads();
  if (t == 0) {
    // determine_connected_constraints (UC:940-1194): breadth-first from the lowest body; a node's contacts join the
    // island, in list order, when the node is first popped; neighbours are pushed in edge-insertion (= list) order
    int cnt = 0, start = -1;
    for (int i = 0; i < nb; i++) if (s_node[i]) { start = i; break; }
    if (start >= 0) {
      int qlen = 0;
      s_queue[qlen++] = start; s_queued[start] = 1;
      for (int qi = 0; qi < qlen; qi++) {
        const int nd = s_queue[qi];
        for (int i = 0; i < nc; i++) {
          const int g1 = s_b1[i], g2 = s_b2[i];
          if (g1 < 0 || g2 < 0) continue;
          int other = -1;
          if (g1 == nd) other = g2; else if (g2 == nd) other = g1;
          if (other >= 0 && !s_queued[other]) { s_queued[other] = 1; s_queue[qlen++] = other; }
        }
        for (int i = 0; i < nc; i++)
          if (!s_done[i] && (s_b1[i] == nd || s_b2[i] == nd)) { s_order[cnt++] = i; s_done[i] = 1; }
      }
    }
    const int single = (cnt == nc);
    s_flag[2] = single;
    const int need = s_flag[0] && single && s_flag[1] && !s_flag[4];
    s_flag[3] = need;
    d.need[b] = need; d.need2[b] = 0; d.again[b] = 0; d.lst1[b] = 1; d.lst2[b] = 1; d.piv1[b] = 0u; d.piv2[b] = 0u;
    d.pivots[b] = 0u; d.solves[b] = 0;
    if ((s_flag[0] && !need) || s_flag[4]) d.status[b] |= MH_WORLD_UNSUPPORTED;     // several islands / the no-slip model: not built here; malformed contact
  }
  __syncthreads();
  for (int i = t; i < nc * 3; i += T) d.imp[(size_t)b * nc * 3 + i] = 0.0;
  if (!s_flag[3]) return;
  // rows of the island's contacts (ICH:1847-1895), X C^T (SparseJacobian::mult, SparseJacobian.cpp:46-82) and C v
  for (int k = t; k < nc; k += T) {
    const int i = s_order[k];
    const size_t ck = (size_t)b * nc + k;
    d.order[ck] = i;
    const int bod[2] = { s_b1[i], s_b2[i] };
    d.cbody[2 * ck] = bod[0]; d.cbody[2 * ck + 1] = bod[1];
    d.cpar[4 * ck] = C[i].mu_coulomb; d.cpar[4 * ck + 1] = C[i].mu_viscous; d.cpar[4 * ck + 2] = C[i].epsilon; d.cpar[4 * ck + 3] = C[i].compliance;
    const P3 p = ld3(C[i].point), n = ld3(C[i].normal);
    P3 s, tt; basis(n, s, tt);
    for (int dd = 0; dd < 3; dd++) {
      const P3 dir = (dd == 0) ? n : (dd == 1 ? s : tt);
      double res = 0.0;
      for (int sd = 0; sd < 2; sd++) {
        double* w = d.W + ck * 36 + dd * 12 + sd * 6;
        double* xj = d.XJ + ck * 36 + dd * 12 + sd * 6;
        const int bb = bod[sd];
        if (bb < 0) { for (int q = 0; q < 6; q++) { w[q] = 0.0; xj[q] = 0.0; } continue; }
        const P3 dv = (sd == 0) ? dir : -dir;
        const P3 r = p - ld3(st + 13 * bb);
        const P3 rxd = cross3(r, dv);
        const double ww[6] = { dv.x, dv.y, dv.z, rxd.x, rxd.y, rxd.z };
        const double* X = s_xinv + 10 * bb;
        for (int q = 0; q < 3; q++) xj[q] = 0.0 + ww[q] * X[0];
        for (int c = 0; c < 3; c++) xj[3 + c] = ((0.0 + ww[3] * X[1 + c]) + ww[4] * X[1 + 3 + c]) + ww[5] * X[1 + 6 + c];
        double tmp = 0.0;
        for (int q = 0; q < 6; q++) { w[q] = ww[q]; tmp = tmp + ww[q] * st[13 * bb + 7 + q]; }
        res = res + tmp;
      }
      d.Cv[((size_t)b * 3 + dd) * nc + k] = res;
    }
  }
}

// the six blocks C_a X C_b^T, a <= b (oracle compute_problem_data: sum over the row's blocks of 6-term dots)
__global__ __launch_bounds__(T)
void mh_k_imp_gram(Dev d)
{
  const int b = blockIdx.y;
  if (!d.need[b]) return;
  const int nc = d.nc;
  const int e = blockIdx.x * T + threadIdx.x;
  if (e >= nc * nc) return;
  const int i = e / nc, j = e - i * nc;
  const size_t ci = (size_t)b * nc + i, cj = (size_t)b * nc + j;
  const int bi[2] = { d.cbody[2 * ci], d.cbody[2 * ci + 1] };
  const int bj[2] = { d.cbody[2 * cj], d.cbody[2 * cj + 1] };
  int match[2];
  for (int s = 0; s < 2; s++) match[s] = (bi[s] < 0) ? -1 : (bi[s] == bj[0] ? 0 : (bi[s] == bj[1] ? 1 : -1));
  const double* Wi = d.W + ci * 36;
  const double* Xj = d.XJ + cj * 36;
  const int ba[6] = { 0, 0, 0, 1, 1, 2 }, bbk[6] = { 0, 1, 2, 1, 2, 2 };
  for (int blk = 0; blk < 6; blk++) {
    double res = 0.0;
    for (int s = 0; s < 2; s++) {
      if (bi[s] < 0) continue;                 // a static side contributes no block
      double tmp = 0.0;
      if (match[s] >= 0) {
        const double* w = Wi + ba[blk] * 12 + s * 6;
        const double* x = Xj + bbk[blk] * 12 + match[s] * 6;
        for (int q = 0; q < 6; q++) tmp = tmp + w[q] * x[q];
      }
      res = res + tmp;
    }
    d.G[(((size_t)b * 6 + blk) * nc + i) * nc + j] = res;
  }
}

struct MMv {
  const double* G; const double* cpar; const double* fcos; const double* fsin; int nc, kh, nvars;
  MH_DEV double gab(int a, int b, int i, int j) const {
    // block index of (a <= b): 00->0 01->1 02->2 11->3 12->4 22->5
    if (a <= b) { const int blk = (a == 0) ? b : (a == 1 ? 2 + b : 5); return G[((size_t)blk * nc + i) * nc + j]; }
    const int blk = (b == 0) ? a : (b == 1 ? 2 + a : 5);
    return G[((size_t)blk * nc + j) * nc + i];
  }
  MH_DEV double H(int r, int c) const {            // 5 x 5 blocks over [n, s, t, -s, -t] (ICH-QP:392-440)
    const int a = r / nc, i = r - a * nc, bb = c / nc, j = c - bb * nc;
    const int da = (a == 0) ? 0 : ((a == 1 || a == 3) ? 1 : 2), db = (bb == 0) ? 0 : ((bb == 1 || bb == 3) ? 1 : 2);
    double g = gab(da, db, i, j);
    if ((a >= 3) != (bb >= 3)) g = -g;
    if (r == c && r < nc) g = g + cpar[4 * i + 3];
    return g;
  }
  MH_DEV double lower(int rr, int c) const {        // rows below H: Cn v+ >= 0, then the friction polygons
    if (rr < nc) return H(rr, c);
    const int fr = rr - nc, i = fr / kh, j = fr - i * kh;
    const int bc = c / nc, ci = c - bc * nc;
    if (ci != i) return 0.0;
    if (bc == 0) return cpar[4 * i];
    return (bc == 1 || bc == 3) ? -fcos[j] : -fsin[j];
  }
  MH_DEV double at(int r, int c) const {
    if (r < nvars) return (c < nvars) ? H(r, c) : -lower(c - nvars, r);
    return (c < nvars) ? lower(r - nvars, c) : 0.0;
  }
};

// _MM (column c = blockIdx.x < n) and, in the extra workgroup c == n, _qq and the start z
__global__ __launch_bounds__(T)
void mh_k_imp_mm(Dev d, const int* __restrict__ run_if, int only_q)
{
  const int b = blockIdx.y;
  if (!run_if[b]) return;
  const int nc = d.nc, n = d.n, nvars = 5 * nc, t = threadIdx.x;
  const int c = only_q ? n : (int)blockIdx.x;      // only_q: launched with one workgroup per world
  if (c < n) {
    MMv m; m.G = d.G + (size_t)b * 6 * nc * nc; m.cpar = d.cpar + (size_t)b * nc * 4; m.fcos = d.fcos; m.fsin = d.fsin;
    m.nc = nc; m.kh = d.kh; m.nvars = nvars;
    double* col = d.MM + ((size_t)b * n + c) * n;
    for (int r = t; r < n; r += T) col[r] = m.at(r, c);
    return;
  }
  const double* Cv = d.Cv + (size_t)b * 3 * nc;
  double* q = d.qq + (size_t)b * n;
  for (int r = t; r < n; r += T) {
    double v;
    if (r < nvars) {
      const int a = r / nc, i = r - a * nc;
      v = (a == 0) ? Cv[i] : (a == 1 ? Cv[nc + i] : (a == 2 ? Cv[2 * nc + i] : (a == 3 ? -Cv[nc + i] : -Cv[2 * nc + i])));
    } else if (r < nvars + nc) v = Cv[r - nvars];
    else {
      const int i = (r - nvars - nc) / d.kh;
      const double vel = sqrt(Cv[nc + i] * Cv[nc + i] + Cv[2 * nc + i] * Cv[2 * nc + i]);
      v = d.cpar[((size_t)b * nc + i) * 4 + 1] * vel;
    }
    q[r] = v;
  }
  // _z.resize(n); if (_zlast.size() == n) _z = _zlast (ICH-QP:158-162): with one n per batch _z's own storage is
  // only ever read as zeros (first call) -- see DESIGN.md
  const bool warm = d.zlast_size[b] == n;
  for (int r = t; r < n; r += T) d.z[(size_t)b * n + r] = warm ? d.zlast[(size_t)b * n + r] : 0.0;
  if (t == 0) d.zsz[b] = n;
}

// worlds whose lcp_fast ladder failed: z.set_zero() and the Lemke ladder next (ICH-QP:222-224)
__global__ __launch_bounds__(T)
void mh_k_imp_lemke_prep(Dev d, const int* __restrict__ run_if)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const bool go = run_if[b] && d.lst1[b] == 0;
  if (t == 0) d.need2[b] = go ? 1 : 0;
  if (!go) return;
  for (int r = t; r < d.n; r += T) d.z[(size_t)b * d.n + r] = 0.0;
}

// after a solve.  phase 0: ICH:569-598 (first solve); phase 1: ICH:600 (second solve)
__global__ __launch_bounds__(T)
void mh_k_imp_post(Dev d, const int* __restrict__ run_if, int* __restrict__ again, int phase)
{
  const int b = blockIdx.x, t = threadIdx.x;
  if (!run_if[b]) return;
  const int nc = d.nc, nb = d.nb, n = d.n;
  __shared__ double s_c[3][MAXC], s_cv[3][MAXC];
  __shared__ int s_b[2][MAXC];
  __shared__ double s_red[T];
  __shared__ int s_any;
  double* st = d.state + (size_t)b * nb * 13;
  const double* z = d.z + (size_t)b * n;
  const bool ok = d.lst1[b] != 0 || d.lst2[b] != 0;
  if (t == 0) {
    d.pivots[b] += d.piv1[b] + (d.lst1[b] == 0 ? d.piv2[b] : 0u);
    d.solves[b] += 1;
    if (!ok) d.status[b] |= MH_WORLD_LCP_FAILED;        // LCPSolverException (ICH-QP:225)
    if (phase == 0) again[b] = 0;
    s_any = 0;
  }
  bool second = false;
  if (ok) {
  for (int r = t; r < n; r += T) d.zlast[(size_t)b * n + r] = z[r];          // _zlast = _z (ICH-QP:233)
  if (t == 0) d.zlast_size[b] = n;
  const double* G = d.G + (size_t)b * 6 * nc * nc;
  const double* cpar = d.cpar + (size_t)b * nc * 4;
  MMv m; m.G = G; m.nc = nc;
  for (int i = t; i < nc; i += T) {
    s_b[0][i] = d.cbody[2 * ((size_t)b * nc + i)]; s_b[1][i] = d.cbody[2 * ((size_t)b * nc + i) + 1];
    for (int a = 0; a < 3; a++) s_cv[a][i] = d.Cv[((size_t)b * 3 + a) * nc + i];
  }
  __syncthreads();
  // one application of impulses: from_stacked (UCPD:218-228) with cn optionally scaled by epsilon, then
  // update_from_stacked (ICH:298-410) and update_constraint_velocities_from_impulses (ICH:427-464)
  auto apply = [&](bool restitution) {
    for (int i = t; i < nc; i += T) {
      double cn = z[i];
      if (restitution) cn = cn * cpar[4 * i + 2];
      double s = z[nc + i]; s = s - z[3 * nc + i];
      double tt = z[2 * nc + i]; tt = tt - z[4 * nc + i];
      s_c[0][i] = cn; s_c[1][i] = s; s_c[2][i] = tt;
      double* im = d.imp + ((size_t)b * nc + d.order[(size_t)b * nc + i]) * 3;
      im[0] += cn; im[1] += s; im[2] += tt;
    }
    __syncthreads();
    // dv(r) = sum_d sum_j c_d[j] * (X C_d^T)(r, j), j ascending; v += dv
    for (int e = t; e < nb * 6; e += T) {
      const int bb = e / 6, q = e - bb * 6;
      double dv = 0.0;
      bool touched = false;
      for (int dd = 0; dd < 3; dd++) {
        double tmp = 0.0;
        for (int j = 0; j < nc; j++) {
          const int sd = (s_b[0][j] == bb) ? 0 : ((s_b[1][j] == bb) ? 1 : -1);
          if (sd < 0) continue;
          touched = true;
          tmp = tmp + s_c[dd][j] * d.XJ[((size_t)b * nc + j) * 36 + dd * 12 + sd * 6 + q];
        }
        dv = (dd == 0) ? tmp : dv + tmp;
      }
      if (touched) st[13 * bb + 7 + q] = st[13 * bb + 7 + q] + dv;
    }
    // C_a v += sum_b (C_a X C_b^T) c_b, each product accumulated from 0 over ascending j
    for (int i = t; i < nc; i += T) {
      for (int a = 0; a < 3; a++) {
        double y = s_cv[a][i];
        for (int bb = 0; bb < 3; bb++) {
          double acc = 0.0;
          for (int j = 0; j < nc; j++) acc = acc + s_c[bb][j] * m.gab(a, bb, i, j);
          y = y + acc;
        }
        s_cv[a][i] = y;
      }
    }
    __syncthreads();
  };
  auto min_cn_v = [&]() -> double {
    double mn = 1.7976931348623157e308;
    for (int i = t; i < nc; i += T) mn = (s_cv[0][i] < mn) ? s_cv[0][i] : mn;
    s_red[t] = mn;
    __syncthreads();
    for (int w = T / 2; w > 0; w >>= 1) { if (t < w) s_red[t] = (s_red[t + w] < s_red[t]) ? s_red[t + w] : s_red[t]; __syncthreads(); }
    const double r = s_red[0];
    __syncthreads();
    return r;
  };
  apply(false);
  if (phase == 0) {
    const double minv = min_cn_v();                                             // ICH:575
    for (int i = t; i < nc; i += T) if (z[i] * cpar[4 * i + 2] > NEAR_ZERO_) s_any = 1;   // apply_restitution (ICH:470-491)
    __syncthreads();
    if (s_any) {
      apply(true);                                                              // ICH:581
      const double minv_plus = min_cn_v();
      second = (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO_);              // ICH:591
    }
  }
  for (int i = t; i < nc; i += T) for (int a = 0; a < 3; a++) d.Cv[((size_t)b * 3 + a) * nc + i] = s_cv[a][i];
  if (t == 0 && phase == 0) again[b] = second ? 1 : 0;
  }   // ok
  if (second) return;
  // ImpactToleranceException test (ICH:159-168): any contact still approaching faster than NEAR_ZERO
  __syncthreads();
  const mh_contact* C = d.contacts + (size_t)b * nc;
  bool bad = false;
  for (int i = t; i < nc; i += T) {
    int g1 = C[i].body1, g2 = C[i].body2;
    if (g1 < 0 || g1 >= nb) g1 = -1;
    if (g2 < 0 || g2 >= nb) g2 = -1;
    const P3 p = ld3(C[i].point), nn = ld3(C[i].normal);
    if (dot3(nn, point_vel(st, g1, p) - point_vel(st, g2, p)) < -NEAR_ZERO_) bad = true;
  }
  if (bad) atomicOr(&d.status[b], MH_WORLD_IMPACT_TOL);
}

}} // namespace mh::imp

struct mh_impact_batch {
  int B, nb, nc, nk, n;
  mh::imp::Dev d;
  std::vector<void*> allocs;
  double* ws_d; int* ws_i;
};

extern "C" {

int mh_impact_batch_destroy(mh_impact_batch* ib)
{
  if (!ib) return MH_OK;
  (void)hipDeviceSynchronize();
  for (void* p : ib->allocs) if (p) (void)hipFree(p);
  delete ib;
  return MH_OK;
}

int mh_impact_batch_create(int B, int nb, int nc, int nk, const double* mass, const double* inertia, mh_impact_batch** out)
{
  namespace im = mh::imp;
  if (!out) return fail(MH_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (B <= 0 || nb <= 0 || nc <= 0) return fail(MH_ERR_INVALID_ARG, "B, nb, nc must be positive");
  if (!mass || !inertia) return fail(MH_ERR_INVALID_ARG, "null mass/inertia");
  if (nk < 4 || (nk & 1)) return fail(MH_ERR_INVALID_ARG, "nk must be even and >= 4 (ContactParameters.cpp:128-135), got %d", nk);
  if (nb > im::MAXB) return fail(MH_ERR_INVALID_ARG, "nb = %d > %d bodies per world", nb, im::MAXB);
  if (nc > im::MAXC) return fail(MH_ERR_INVALID_ARG, "nc = %d > %d contacts per world", nc, im::MAXC);
  const long n = 6L * nc + (long)nc * (nk / 2);
  if (n > MH_LCP_MAX_N_BLOCK) return fail(MH_ERR_UNSUPPORTED_N, "impact LCP n = %ld > %d", n, MH_LCP_MAX_N_BLOCK);
  for (int i = 0; i < nb; i++) {
    if (!(mass[i] > 0.0) || !(inertia[3*i] > 0.0) || !(inertia[3*i+1] > 0.0) || !(inertia[3*i+2] > 0.0))
      return fail(MH_ERR_INVALID_ARG, "body %d: mass and principal inertias must be positive", i);
  }
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  mh_impact_batch* ib = new mh_impact_batch();
  ib->B = B; ib->nb = nb; ib->nc = nc; ib->nk = nk; ib->n = (int)n; ib->ws_d = nullptr; ib->ws_i = nullptr;
  im::Dev& d = ib->d;
  std::memset(&d, 0, sizeof(d));
  d.B = B; d.nb = nb; d.nc = nc; d.nk = nk; d.kh = nk / 2; d.n = (int)n;
  bool okall = true;
  auto A = [&](size_t bytes, bool zero) -> void* {
    void* p = nullptr;
    if (!okall) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { okall = false; return nullptr; }
    ib->allocs.push_back(p);
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) okall = false;
    return p;
  };
  const size_t sB = (size_t)B, sn = (size_t)n, snc = (size_t)nc;
  double* dmass = (double*)A(nb * 8, false); double* dJ = (double*)A(nb * 24, false);
  d.state = (double*)A(sB * nb * 13 * 8, true);
  mh_contact* dcon = (mh_contact*)A(sB * snc * sizeof(mh_contact), true);
  d.order = (int*)A(sB * snc * 4, true); d.cbody = (int*)A(sB * snc * 8, true); d.cpar = (double*)A(sB * snc * 32, true);
  d.W = (double*)A(sB * snc * 36 * 8, true); d.XJ = (double*)A(sB * snc * 36 * 8, true);
  d.Cv = (double*)A(sB * 3 * snc * 8, true); d.G = (double*)A(sB * 6 * snc * snc * 8, true);
  d.MM = (double*)A(sB * sn * sn * 8, false); d.qq = (double*)A(sB * sn * 8, true);
  d.z = (double*)A(sB * sn * 8, true); d.zlast = (double*)A(sB * sn * 8, true);
  d.zlast_size = (int*)A(sB * 4, true); d.zsz = (int*)A(sB * 4, true);
  d.rng = (uint32_t*)A(sB * MH_RAND_WORDS * 4, false);
  d.need = (int*)A(sB * 4, true); d.need2 = (int*)A(sB * 4, true); d.lst1 = (int*)A(sB * 4, true); d.lst2 = (int*)A(sB * 4, true);
  d.piv1 = (unsigned*)A(sB * 4, true); d.piv2 = (unsigned*)A(sB * 4, true);
  d.status = (int*)A(sB * 4, true); d.pivots = (unsigned*)A(sB * 4, true); d.solves = (int*)A(sB * 4, true);
  d.imp = (double*)A(sB * snc * 3 * 8, true);
  d.again = (int*)A(sB * 4, true);
  double* dcos = (double*)A(d.kh * 8, false); double* dsin = (double*)A(d.kh * 8, false);
  if (n > MH_LCP_MAX_N_WAVE) {
    ib->ws_d = (double*)A(sB * (sn * sn + 5 * sn) * 8, false);
    ib->ws_i = (int*)A(sB * 4 * sn * 4, false);
  }
  if (!okall) { mh_impact_batch_destroy(ib); return fail(MH_ERR_HIP, "device allocation failed (B = %d, n = %ld: %.1f GB for _MM + LU workspace)", B, n, 2.0 * sB * sn * sn * 8 / 1e9); }
  d.mass = dmass; d.inertia = dJ; d.contacts = dcon; d.fcos = dcos; d.fsin = dsin;
  // friction polygon directions (ICH-QP:452-470) with the host libm, like the world kernel's table
  std::vector<double> hc(d.kh), hs(d.kh);
  for (int j = 0; j < d.kh; j++) { const double theta = (double)j / (d.kh - 1) * M_PI_2; hc[j] = std::cos(theta); hs[j] = std::sin(theta); }
  std::vector<uint32_t> hr((size_t)B * MH_RAND_WORDS);
  mh_rand_seed(hr.data(), 1u);
  for (int b = 1; b < B; b++) std::memcpy(&hr[(size_t)b * MH_RAND_WORDS], hr.data(), MH_RAND_WORDS * 4);
  bool ok = hipMemcpy(dmass, mass, nb * 8, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(dJ, inertia, nb * 24, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(dcos, hc.data(), d.kh * 8, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(dsin, hs.data(), d.kh * 8, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(d.rng, hr.data(), hr.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mh_impact_batch_destroy(ib); return fail(MH_ERR_HIP, "upload of body tables failed"); }
  *out = ib;
  return MH_OK;
}

int mh_impact_batch_lcp_size(const mh_impact_batch* ib) { return ib ? ib->n : 0; }

int mh_impact_batch_upload(mh_impact_batch* ib, const double* state, const mh_contact* contacts)
{
  if (!ib || !state || !contacts) return fail(MH_ERR_INVALID_ARG, "null batch/state/contacts");
  const size_t ncon = (size_t)ib->B * ib->nc;
  for (size_t i = 0; i < ncon; i++) {
    const mh_contact& c = contacts[i];
    if (c.nk != ib->nk) return fail(MH_ERR_INVALID_ARG, "contact %zu has nk = %d, the batch was created for nk = %d", i, c.nk, ib->nk);
    const double nn = c.normal[0] * c.normal[0] + c.normal[1] * c.normal[1] + c.normal[2] * c.normal[2];
    if (!(nn > 0.25 && nn < 4.0)) return fail(MH_ERR_INVALID_ARG, "contact %zu: normal is not a unit vector", i);
    const bool s1 = c.body1 < 0 || c.body1 >= ib->nb, s2 = c.body2 < 0 || c.body2 >= ib->nb;
    if (!s1 && c.body1 == c.body2) return fail(MH_ERR_INVALID_ARG, "contact %zu: body1 == body2", i);
    if (s1 && s2) return fail(MH_ERR_INVALID_ARG, "contact %zu joins two static bodies", i);
  }
  MH_HIP(hipMemcpy(ib->d.state, state, (size_t)ib->B * ib->nb * 13 * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(const_cast<mh_contact*>(ib->d.contacts), contacts, ncon * sizeof(mh_contact), hipMemcpyHostToDevice));
  return MH_OK;
}

static int impact_solve_round(mh_impact_batch* ib, hipStream_t s, const int* run_if)
{
  namespace im = mh::imp;
  im::Dev& d = ib->d;
  const int B = ib->B, n = ib->n;
  mh_lcp_opts o1; o1.min_exp = -20; o1.step_exp = 4u; o1.max_exp = -8; o1.piv_tol = -1.0; o1.zero_tol = -1.0;   // ICH-QP:219
  int rc = mh_lcp_solve_dev_masked(s, MH_LCP_FAST_REG, B, n, d.MM, n, (long)n * n, d.qq, d.z, d.zsz, nullptr, d.rng, d.lst1, d.piv1,
                                nullptr, 0, nullptr, &o1, run_if, ib->ws_d, ib->ws_i);
  if (rc != MH_OK) return rc;
  hipLaunchKernelGGL(im::mh_k_imp_lemke_prep, dim3(B), dim3(im::T), 0, s, d, run_if);
  MH_HIP(hipGetLastError());
  rc = mh_lcp_solve_dev_masked(s, MH_LCP_LEMKE_REG, B, n, d.MM, n, (long)n * n, d.qq, d.z, d.zsz, nullptr, d.rng, d.lst2, d.piv2,
                            nullptr, 0, nullptr, nullptr, d.need2, ib->ws_d, ib->ws_i);                                    // ICH-QP:224
  return rc;
}

int mh_impact_batch_process(mh_impact_batch* ib, void* stream)
{
  namespace im = mh::imp;
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  hipStream_t s = (hipStream_t)stream;
  im::Dev& d = ib->d;
  const int B = ib->B, n = ib->n, nc = ib->nc;
  hipLaunchKernelGGL(im::mh_k_imp_prep, dim3(B), dim3(im::T), 0, s, d);
  hipLaunchKernelGGL(im::mh_k_imp_gram, dim3((nc * nc + im::T - 1) / im::T, B), dim3(im::T), 0, s, d);
  hipLaunchKernelGGL(im::mh_k_imp_mm, dim3(n + 1, B), dim3(im::T), 0, s, d, (const int*)d.need, 0);
  MH_HIP(hipGetLastError());
  int rc = impact_solve_round(ib, s, d.need);
  if (rc != MH_OK) return rc;
  hipLaunchKernelGGL(im::mh_k_imp_post, dim3(B), dim3(im::T), 0, s, d, (const int*)d.need, d.again, 0);
  MH_HIP(hipGetLastError());
  {
    // second solve for the worlds whose restitution impulses left a contact approaching (ICH:591-600): same _MM, new _qq.
    // The mask `again` is computed on the device (mh_k_imp_post, phase 0) and the round is always enqueued -- masked
    // workgroups exit at once -- so contacts written through device_ptrs() get it too (no host-side epsilon scan).
    hipLaunchKernelGGL(im::mh_k_imp_mm, dim3(1, B), dim3(im::T), 0, s, d, (const int*)d.again, 1);
    MH_HIP(hipGetLastError());
    rc = impact_solve_round(ib, s, d.again);
    if (rc != MH_OK) return rc;
    hipLaunchKernelGGL(im::mh_k_imp_post, dim3(B), dim3(im::T), 0, s, d, (const int*)d.again, d.again, 1);
    MH_HIP(hipGetLastError());
  }
  return MH_OK;
}

int mh_impact_batch_download(mh_impact_batch* ib, double* state, double* impulses, int* status, unsigned* pivots, int* solves)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  if (state) MH_HIP(hipMemcpy(state, ib->d.state, B * ib->nb * 13 * 8, hipMemcpyDeviceToHost));
  if (impulses) MH_HIP(hipMemcpy(impulses, ib->d.imp, B * ib->nc * 3 * 8, hipMemcpyDeviceToHost));
  if (status) MH_HIP(hipMemcpy(status, ib->d.status, B * 4, hipMemcpyDeviceToHost));
  if (pivots) MH_HIP(hipMemcpy(pivots, ib->d.pivots, B * 4, hipMemcpyDeviceToHost));
  if (solves) MH_HIP(hipMemcpy(solves, ib->d.solves, B * 4, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_debug_lcp(mh_impact_batch* ib, double* MM, double* qq)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B, n = (size_t)ib->n;
  if (MM) MH_HIP(hipMemcpy(MM, ib->d.MM, B * n * n * 8, hipMemcpyDeviceToHost));
  if (qq) MH_HIP(hipMemcpy(qq, ib->d.qq, B * n * 8, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_save_solver_state(mh_impact_batch* ib, double* zlast, int* zlast_size, uint32_t* rng, int* status)
{
  if (!ib || !zlast || !zlast_size || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  MH_HIP(hipMemcpy(zlast, ib->d.zlast, B * ib->n * 8, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(zlast_size, ib->d.zlast_size, B * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(rng, ib->d.rng, B * MH_RAND_WORDS * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(status, ib->d.status, B * 4, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_load_solver_state(mh_impact_batch* ib, const double* zlast, const int* zlast_size, const uint32_t* rng,
                                      const int* status)
{
  if (!ib || !zlast || !zlast_size || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  for (int b = 0; b < ib->B; b++)
    if (zlast_size[b] != 0 && zlast_size[b] != ib->n) return fail(MH_ERR_INVALID_ARG, "world %d: _zlast of size %d in a batch of n = %d", b, zlast_size[b], ib->n);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  MH_HIP(hipMemcpy(ib->d.zlast, zlast, B * ib->n * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d.zlast_size, zlast_size, B * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d.rng, rng, B * MH_RAND_WORDS * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d.status, status, B * 4, hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_impact_batch_device_ptrs(mh_impact_batch* ib, double** state_dev, mh_contact** contacts_dev)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  if (state_dev) *state_dev = ib->d.state;
  if (contacts_dev) *contacts_dev = const_cast<mh_contact*>(ib->d.contacts);
  return MH_OK;
}

int mh_impact_process_batch(int B, int nb, int nc, int nk, const double* mass, const double* inertia,
                            double* state, const mh_contact* contacts, double* impulses,
                            int* status, unsigned* pivots, int* solves)
{
  if (B == 0) return MH_OK;
  mh_impact_batch* ib = nullptr;
  int rc = mh_impact_batch_create(B, nb, nc, nk, mass, inertia, &ib);
  if (rc != MH_OK) return rc;
  rc = mh_impact_batch_upload(ib, state, contacts);
  if (rc == MH_OK) rc = mh_impact_batch_process(ib, nullptr);
  if (rc == MH_OK) rc = mh_impact_batch_download(ib, state, impulses, status, pivots, solves);
  mh_impact_batch_destroy(ib);
  return rc;
}

} // extern "C"
