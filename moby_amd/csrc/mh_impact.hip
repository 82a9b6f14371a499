// mh_impact.hip -- the island pipeline (mh_imp_core.h) and, on top of it, the batched
// ImpactConstraintHandler::process_constraints on explicit contact lists (include/moby_hip_impact.h; seam B2).
//
// One world = up to MAXC contacts over up to MAXB free bodies, in any number of islands.  Everything a world needs
// lives in HBM (the LCP matrix alone is n^2 doubles: 33.5 MB at n = 2048); the pipeline of one call is
//   k_prep        1 workgroup / world   contact checks, impacting?, the reference's island search (UC:940-1194) over
//                                       ALL islands, remove_inactive_groups (UC:1197-1225), X blocks, Jacobian rows
//                                       [d, r x d] and X C^T rows per contact side, C v -- in island order
//   then per ROUND r (= the r-th island of every world that has one; ICH:113 loops islands sequentially, and islands
//   share no body, so round r of all worlds runs as one set of launches):
//   k_gram        1 thread / (i, j)     the six C_a X C_b^T blocks (ICH:2125-2149) of the island
//   k_mm          1 workgroup / column  _MM = [H, -M^T; M, 0], _qq (ICH-QP:271-497), the start z (_z / _zlast rules)
//   LCP entry     lcp_fast_regularized(-20, 4, -8), then for the worlds it failed on z = 0 + lcp_lemke_regularized
//                 (ICH-QP:219-224), per-island sizes -- wave solver up to 64 rows, block solver above
//   k_post        1 workgroup / world   update_from_stacked, update_constraint_velocities_from_impulses,
//                                       apply_restitution, the second-solve test (ICH:569-600)
//   and a second solve (mm with new _qq, LCP, post) for the worlds that asked;
//   k_finish      1 workgroup / world   ImpactToleranceException test (ICH:157-167).
// Mode MH_CORE_STAB runs ConstraintStabilization::compute_problem_data / determine_dq (CStab:347-492, 932-970) through the
// same kernels: normal rows only, MM = Cn X Cn^T, qq = signed distance - |eps| - NEAR_ZERO, cold lcp_fast then the Lemke
// ladder, velocities := X Cn^T z.
// Arithmetic order follows oracle/world.hpp (compute_problem_data, build_impact_lcp, apply_impulses,
// update_constraint_vels), which restates the reference's SparseJacobian products.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/moby_hip_impact.h"
#include "../../include/moby_hip_stack.h"
#include "mh_host.h"
#include "mh_imp_core.h"
#include "mh_imp_dev.h"
#include "mh_lcp_wave.h"

namespace mh { namespace imp {

constexpr int T = 256;
constexpr int MAXC = 512;      // contacts per world (n = 6 nc + nc nk/2 <= 4096 with nk >= 4)
constexpr int MAXB = 256;      // bodies per world
constexpr double NEAR_ZERO_ = 1.4901161193847656e-08;   // Constants.h:21

typedef mh_imp_core Dev;

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(T)
void k_prep(Dev d, int mode)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const int nb = d.nb, ncmax = d.ncmax;
  const int nc = d.ncount ? d.ncount[b] : ncmax;
  __shared__ int s_b1[MAXC], s_b2[MAXC], s_order[MAXC];
  __shared__ unsigned char s_done[MAXC], s_impf[MAXC], s_inf[MAXC], s_node[MAXB], s_queued[MAXB];
  __shared__ int s_queue[MAXB];
  __shared__ int s_flag[4];   // 0: some contact impacting, 1: malformed contact, 2: contacts kept, 3: islands kept
  const mh_contact* C = d.contacts + (size_t)b * ncmax;
  const double* st = d.state + (size_t)b * nb * 13;
  double* xinv = d.xinv + (size_t)b * nb * 10;
  if (t < 4) s_flag[t] = 0;
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
    const bool impacting = vn < -NEAR_ZERO_;
    s_impf[i] = impacting ? 1 : 0;
    if (impacting) s_flag[0] = 1;
    s_inf[i] = (C[i].mu_coulomb < 1e2) ? 0 : 1;                                  // ICH:127
    // contacts may have been written through mh_impact_batch_device_ptrs, past upload()'s host checks: the same
    // checks here (nk of the batch, unit normal, two different bodies of which one is dynamic)
    const double nn = dot3(n, n);
    if ((mode == MH_CORE_IMPACT && C[i].nk != d.nk) || !(nn > 0.25 && nn < 4.0) || g1 == g2) s_flag[1] = 1;
  }
  for (int i = t; i < nb; i += T) inv_inertia(st + 13 * i, d.inertia + 3 * i, d.mass[i], xinv + 10 * i);
  for (int j = t; j < d.nj; j += T) {                    // the joints' dynamic links are nodes as well (UC:993-1008)
    const int a = d.jin[j], c = d.jout[j];
    if (a >= 0 && a < nb) s_node[a] = 1;
    if (c >= 0 && c < nb) s_node[c] = 1;
  }
  __syncthreads();
  if (t == 0) {
    int cnt = 0, nisl = 0;
    const bool go = !s_flag[1] && (mode == MH_CORE_STAB || s_flag[0]);           // CSim:313-323: nothing impacting => nothing to do
    if (s_flag[1]) d.status[b] |= MH_WORLD_UNSUPPORTED;
    if (go) {
      // determine_connected_constraints (UC:940-1194): breadth-first from the lowest body of each island; a node's
      // contacts join the island, in list order, when the node is first popped; neighbours are pushed in
      // edge-insertion (= list) order.  (Nodes are marked when first pushed: same islands, linear time -- DESIGN 2.7.)
      for (int start = 0; start < nb; start++) {
        if (!s_node[start]) continue;
        const int begin = cnt;
        bool active = false, all_inf = true, has_jointed = false;
        int qlen = 0;
        s_queue[qlen++] = start; s_queued[start] = 1;
        for (int qi = 0; qi < qlen; qi++) {
          const int nd = s_queue[qi];
          s_node[nd] = 0;
          if (d.jointed && d.jointed[nd]) has_jointed = true;
          for (int i = 0; i < nc; i++) {
            const int g1 = s_b1[i], g2 = s_b2[i];
            if (g1 < 0 || g2 < 0) continue;
            int other = -1;
            if (g1 == nd) other = g2; else if (g2 == nd) other = g1;
            if (other >= 0 && !s_queued[other]) { s_queued[other] = 1; s_queue[qlen++] = other; }
          }
          for (int j = 0; j < d.nj; j++) {               // joint edges come after the contact edges in the multimap
            const int a = d.jin[j], c = d.jout[j];
            if (a < 0 || a >= nb || c < 0 || c >= nb) continue;
            const int other = (a == nd) ? c : ((c == nd) ? a : -1);
            if (other >= 0 && !s_queued[other]) { s_queued[other] = 1; s_queue[qlen++] = other; }
          }
          for (int i = 0; i < nc; i++)
            if (!s_done[i] && (s_b1[i] == nd || s_b2[i] == nd)) {
              s_order[cnt++] = i; s_done[i] = 1;
              if (s_impf[i]) active = true;
              if (!s_inf[i]) all_inf = false;
            }
        }
        if (cnt == begin) continue;
        bool keep = true;
        if (mode == MH_CORE_IMPACT) {
          if (!active) keep = false;                                               // remove_inactive_groups (UC:1197-1225)
        }
        // stabilisation of a contact island that holds jointed bodies: compute_X's general case (k_bilat_X), on islands of the built size
        const bool bilat = keep && mode == MH_CORE_STAB && has_jointed;
        if (bilat && (qlen > MH_IJOINT_MAX_BODIES || !d.bT1)) { keep = false; d.status[b] |= MH_WORLD_UNSUPPORTED; }
        if (keep && nisl >= d.islmax) { keep = false; d.status[b] |= MH_WORLD_UNSUPPORTED; }
        if (!keep) { cnt = begin; continue; }
        d.isl_start[(size_t)b * d.islmax + nisl] = begin; d.isl_len[(size_t)b * d.islmax + nisl] = cnt - begin;
        // every mu >= 100: the no-slip model (ICH:134-135); otherwise Drumwright-Shell, or Anitescu-Potra in a USE_AP batch (ICH:139-146)
        d.isl_model[(size_t)b * d.islmax + nisl] = (mode != MH_CORE_IMPACT) ? (bilat ? 3 : 0) : (all_inf ? 1 : (d.ap ? 2 : 0));
        if (bilat) {                                        // the island's super bodies, sorted (CStab:718-722)
          int* ib = d.isl_bod + ((size_t)b * d.islmax + nisl) * MH_IJOINT_MAX_BODIES;
          for (int i = 0; i < qlen; i++) { int v = s_queue[i], k = i; while (k > 0 && ib[k - 1] > v) { ib[k] = ib[k - 1]; k--; } ib[k] = v; }
          d.isl_nbod[(size_t)b * d.islmax + nisl] = qlen;
        }
        nisl++;
      }
    }
    s_flag[2] = cnt; s_flag[3] = nisl;
    d.nisl[b] = nisl; d.run[b] = 0; d.need2[b] = 0; d.again[b] = 0; d.ncur[b] = 0; d.thrown[b] = 0;
    d.lst1[b] = 1; d.lst2[b] = 1; d.piv1[b] = 0u; d.piv2[b] = 0u;
    atomicMax(d.maxisl, nisl);
  }
  __syncthreads();
  for (int i = t; i < ncmax * 3; i += T) d.imp[(size_t)b * ncmax * 3 + i] = 0.0;
  const int kept = s_flag[2];
  // rows of the kept contacts (ICH:1847-1895), X C^T (SparseJacobian::mult, SparseJacobian.cpp:46-82) and C v
  const int ndir = (mode == MH_CORE_STAB) ? 1 : 3;
  for (int k = t; k < kept; k += T) {
    const int i = s_order[k];
    const size_t ck = (size_t)b * ncmax + k;
    d.order[ck] = i;
    const int bod[2] = { s_b1[i], s_b2[i] };
    d.cbody[2 * ck] = bod[0]; d.cbody[2 * ck + 1] = bod[1];
    d.cpar[4 * ck] = C[i].mu_coulomb; d.cpar[4 * ck + 1] = C[i].mu_viscous; d.cpar[4 * ck + 2] = C[i].epsilon; d.cpar[4 * ck + 3] = C[i].compliance;
    const P3 p = ld3(C[i].point), n = ld3(C[i].normal);
    P3 s, tt; basis(n, s, tt);
    for (int dd = 0; dd < ndir; dd++) {
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
        const double* X = xinv + 10 * bb;
        for (int q = 0; q < 3; q++) xj[q] = 0.0 + ww[q] * X[0];
        for (int c = 0; c < 3; c++) xj[3 + c] = ((0.0 + ww[3] * X[1 + c]) + ww[4] * X[1 + 3 + c]) + ww[5] * X[1 + 6 + c];
        double tmp = 0.0;
        for (int q = 0; q < 6; q++) { w[q] = ww[q]; tmp = tmp + ww[q] * st[13 * bb + 7 + q]; }
        res = res + tmp;
      }
      // CStab:431: Cn_v = signed_violation - |eps| - NEAR_ZERO
      if (mode == MH_CORE_STAB) res = d.cdist[(size_t)b * ncmax + i] - fabs(d.stab_eps) - NEAR_ZERO_;
      d.Cv[((size_t)b * 3 + dd) * ncmax + k] = res;
    }
  }
}

// island r of world b: first position and contact count; false when the world has no r-th island
MH_DEV bool island(const Dev& d, int b, int r, int& start, int& nc) {
  if (r >= d.nisl[b]) return false;
  start = d.isl_start[(size_t)b * d.islmax + r]; nc = d.isl_len[(size_t)b * d.islmax + r];
  return true;
}

// the six blocks C_a X C_b^T, a <= b (oracle compute_problem_data: sum over the row's blocks of 6-term dots)
__global__ __launch_bounds__(T)
void k_gram(Dev d, int r, int mode)
{
  const int b = blockIdx.y;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  const int e = blockIdx.x * T + threadIdx.x;
  if (e >= nc * nc) return;
  const int i = e / nc, j = e - i * nc;
  const size_t ci = (size_t)b * d.ncmax + start + i, cj = (size_t)b * d.ncmax + start + j;
  const int bi[2] = { d.cbody[2 * ci], d.cbody[2 * ci + 1] };
  const int bj[2] = { d.cbody[2 * cj], d.cbody[2 * cj + 1] };
  int match[2];
  for (int s = 0; s < 2; s++) match[s] = (bi[s] < 0) ? -1 : (bi[s] == bj[0] ? 0 : (bi[s] == bj[1] ? 1 : -1));
  const double* Wi = d.W + ci * 36;
  const double* Xj = d.XJ + cj * 36;
  const int ba[6] = { 0, 0, 0, 1, 1, 2 }, bbk[6] = { 0, 1, 2, 1, 2, 2 };
  double* G = d.G + (size_t)b * 6 * d.ncmax * d.ncmax;
  const int nblk = (mode == MH_CORE_STAB) ? 1 : 6;
  for (int blk = 0; blk < nblk; blk++) {
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
    G[((size_t)blk * nc + i) * nc + j] = res;
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


// apply_ap_model's _MM = [UL UR; LL 0] (ICH-AP:120-303) over z = [cn, cs+, cs-, ct+, ct-, friction rows]
struct APv {
  MMv g; const double* apcos; const double* apsin; int nc, nk4, nvars;
  MH_DEV double at(int r, int c) const {
    if (r < nvars && c < nvars) {
      const int a = r / nc, i = r - a * nc, bb = c / nc, j = c - bb * nc;
      double v = g.gab((a + 1) >> 1, (bb + 1) >> 1, i, j);            // directions n, s, s, t, t
      if ((a == 2 || a == 4) != (bb == 2 || bb == 4)) v = -v;
      return v;
    }
    if (r >= nvars && c < nvars) {                                      // LL: mu, -cos, -cos, -sin, -sin
      const int fr = r - nvars, i = fr / nk4, k = fr - i * nk4, bc = c / nc, ci = c - bc * nc;
      if (ci != i) return 0.0;
      if (bc == 0) return g.cpar[4 * i];
      return (bc <= 2) ? -apcos[k] : -apsin[k];
    }
    if (r < nvars) {                                                    // UR: the friction part of -LL^T
      const int fc = c - nvars, i = fc / nk4, k = fc - i * nk4, a = r / nc, ri = r - a * nc;
      if (ri != i || a == 0) return 0.0;
      return (a <= 2) ? apcos[k] : apsin[k];
    }
    return 0.0;
  }
};

// _MM (column c = blockIdx.x < n, leading dimension n) and, in the LAST workgroup of a world, _qq, the start z and the
// round's mask.  phase 1 (the second solve of ICH:591-600: same _MM, new _qq) launches only that last workgroup.
__global__ __launch_bounds__(T)
void k_mm(Dev d, int r, int mode, int phase)
{
  const int b = blockIdx.y, t = threadIdx.x;
  const bool last = (blockIdx.x == gridDim.x - 1);
  int start, nc;
  if (!island(d, b, r, start, nc)) {                  // this world has no r-th island: it sits the round out
    if (last && t == 0 && phase == 0) { d.run[b] = 0; d.ncur[b] = 0; d.again[b] = 0; }
    return;
  }
  if (phase == 1 && !d.again[b]) return;
  if (d.isl_model[(size_t)b * d.islmax + r] == 1 || d.isl_model[(size_t)b * d.islmax + r] == 4) {   // the no-slip model builds its own (nc x nc) LCP: k_noslip; 4: dropped by k_bilat_X
    if (last && t == 0 && phase == 0) { d.run[b] = 0; d.ncur[b] = 0; d.again[b] = 0; }
    return;
  }
  const int nvars = 5 * nc;
  const bool apm = d.isl_model[(size_t)b * d.islmax + r] == 2;
  const int n = (mode == MH_CORE_STAB) ? nc : (apm ? nvars + nc * d.nk4 : nvars + nc + nc * d.kh);
  if (n > d.nmax) {                                   // beyond the capacity the batch was created with
    if (last && t == 0) { d.status[b] |= MH_WORLD_UNSUPPORTED; d.run[b] = 0; d.ncur[b] = 0; }
    return;
  }
  const double* G = d.G + (size_t)b * 6 * d.ncmax * d.ncmax;
  const double* cpar = d.cpar + ((size_t)b * d.ncmax + start) * 4;
  if (!last) {
    const int c = (int)blockIdx.x;
    if (c >= n) return;
    double* col = d.MM + (size_t)b * d.nmax * d.nmax + (size_t)c * n;
    if (mode == MH_CORE_STAB) { for (int rr = t; rr < n; rr += T) col[rr] = G[(size_t)rr * nc + c]; return; }   // MM = Cn X Cn^T (CStab:940)
    MMv m; m.G = G; m.cpar = cpar; m.fcos = d.fcos; m.fsin = d.fsin; m.nc = nc; m.kh = d.kh; m.nvars = nvars;
    if (apm) {
      APv a; a.g = m; a.apcos = d.apcos; a.apsin = d.apsin; a.nc = nc; a.nk4 = d.nk4; a.nvars = nvars;
      for (int rr = t; rr < n; rr += T) col[rr] = a.at(rr, c);
      return;
    }
    for (int rr = t; rr < n; rr += T) col[rr] = m.at(rr, c);
    return;
  }
  const double* Cv = d.Cv + (size_t)b * 3 * d.ncmax + start;     // direction a at Cv + a * ncmax
  const int ncm = d.ncmax;
  double* q = d.qq + (size_t)b * d.nmax;
  double* z = d.z + (size_t)b * d.nmax;
  if (mode == MH_CORE_STAB) {
    for (int rr = t; rr < n; rr += T) { q[rr] = Cv[rr]; z[rr] = 0.0; }
    if (t == 0) { d.zsz[b] = 0; d.ncur[b] = n; d.run[b] = 1; }      // VectorNd z: fresh, size 0 => cold start (CStab:934)
    return;
  }
  if (apm) {                                            // ICH-AP:186-187, 305-311; VectorNd z: fresh (ICH-AP:331)
    for (int rr = t; rr < n; rr += T) {
      double v = 0.0;
      if (rr < nvars) {
        const int a = rr / nc, i = rr - a * nc;
        v = (a == 0) ? Cv[i] : (a == 1 ? Cv[ncm + i] : (a == 2 ? -Cv[ncm + i] : (a == 3 ? Cv[2 * ncm + i] : -Cv[2 * ncm + i])));
      }
      q[rr] = v; z[rr] = 0.0;
    }
    if (t == 0) { d.zsz[b] = 0; d.ncur[b] = n; d.run[b] = 1; }
    return;
  }
  for (int rr = t; rr < n; rr += T) {
    double v;
    if (rr < nvars) {
      const int a = rr / nc, i = rr - a * nc;
      v = (a == 0) ? Cv[i] : (a == 1 ? Cv[ncm + i] : (a == 2 ? Cv[2 * ncm + i] : (a == 3 ? -Cv[ncm + i] : -Cv[2 * ncm + i])));
    } else if (rr < nvars + nc) v = Cv[rr - nvars];
    else {
      const int i = (rr - nvars - nc) / d.kh;
      const double vel = sqrt(Cv[ncm + i] * Cv[ncm + i] + Cv[2 * ncm + i] * Cv[2 * ncm + i]);
      v = cpar[4 * i + 1] * vel;
    }
    q[rr] = v;
  }
  // _z.resize(n) keeps its storage when it fits and comes back zeroed when it has to grow; then
  // if (_zlast.size() == n) _z = _zlast (ICH-QP:158-162).  lcp_fast then warm-starts from whatever _z holds (LCP.cpp:65).
  const bool warm = d.zlast_size[b] == n;
  const bool keep = n <= d.zbuf_cap[b];
  const double* zl = d.zlast + (size_t)b * d.nmax; const double* zb = d.zbuf + (size_t)b * d.nmax;
  for (int rr = t; rr < n; rr += T) z[rr] = warm ? zl[rr] : (keep ? zb[rr] : 0.0);
  if (t == 0) { d.zsz[b] = n; d.ncur[b] = n; d.run[b] = 1; }
}

// worlds whose first solver failed: the Lemke ladder next.  Impact: z.set_zero() first (ICH-QP:222-224, the size stays);
// stabilisation: z as lcp_fast left it (CStab:954-955).
__global__ __launch_bounds__(T)
void k_lemke_prep(Dev d, const int* __restrict__ run_if, int mode)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const bool go = run_if[b] && d.lst1[b] == 0;
  if (t == 0) d.need2[b] = go ? 1 : 0;
  if (!go || mode == MH_CORE_STAB) return;
  const int n = d.ncur[b];
  for (int rr = t; rr < n; rr += T) d.z[(size_t)b * d.nmax + rr] = 0.0;
}

MH_DEV void account(const Dev& d, int b, int n) {      // World::lcp_account
  unsigned long long* c = d.cnt + (size_t)b * 5;
  c[0] += 1ull; c[1] += (unsigned long long)n; c[2] += (unsigned long long)(d.piv1[b] + (d.lst1[b] == 0 ? d.piv2[b] : 0u));
  c[3] += 8ull * ((unsigned long long)n * n + 2ull * n);
}

// after a solve.  phase 0: ICH:569-598 (first solve); phase 1: ICH:600 (second solve)
__global__ __launch_bounds__(T)
void k_post(Dev d, int r, int phase)
{
  const int b = blockIdx.x, t = threadIdx.x;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (!d.run[b]) return;
  if (phase == 1 && !d.again[b]) return;
  if (d.isl_model[(size_t)b * d.islmax + r] != 0) return;
  const int nb = d.nb, n = d.ncur[b], ncm = d.ncmax;
  __shared__ double s_c[3][MAXC], s_cv[3][MAXC];
  __shared__ int s_b[2][MAXC];
  __shared__ double s_red[1];
  __shared__ int s_any;
  double* st = d.state + (size_t)b * nb * 13;
  const double* z = d.z + (size_t)b * d.nmax;
  const bool ok = d.lst1[b] != 0 || d.lst2[b] != 0;
  if (t == 0) {
    account(d, b, n);
    if (!ok) { d.status[b] |= MH_WORLD_LCP_FAILED; d.thrown[b] = 1; }        // LCPSolverException (ICH-QP:225)
    if (phase == 0) d.again[b] = 0;
    s_any = 0;
  }
  bool second = false;
  if (ok) {
  double* zb = d.zbuf + (size_t)b * d.nmax;
  const double* cpar = d.cpar + ((size_t)b * ncm + start) * 4;
  // _zlast = _z (ICH-QP:233); _z's storage keeps the solution, then the epd repack resizes it to 5 nc (ICH-QP:236-250)
  for (int rr = t; rr < n; rr += T) { d.zlast[(size_t)b * d.nmax + rr] = z[rr]; zb[rr] = z[rr]; }
  if (t == 0) { d.zlast_size[b] = n; if (d.zbuf_cap[b] < n) d.zbuf_cap[b] = n; d.zbuf_size[b] = 5 * nc; }
  const double* G = d.G + (size_t)b * 6 * ncm * ncm;
  MMv m; m.G = G; m.nc = nc;
  for (int i = t; i < nc; i += T) {
    const size_t ck = (size_t)b * ncm + start + i;
    s_b[0][i] = d.cbody[2 * ck]; s_b[1][i] = d.cbody[2 * ck + 1];
    for (int a = 0; a < 3; a++) s_cv[a][i] = d.Cv[((size_t)b * 3 + a) * ncm + start + i];
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
      double* im = d.imp + ((size_t)b * ncm + d.order[(size_t)b * ncm + start + i]) * 3;
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
          tmp = tmp + s_c[dd][j] * d.XJ[((size_t)b * ncm + start + j) * 36 + dd * 12 + sd * 6 + q];
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
  auto min_cn_v = [&]() -> double {                     // one thread, at most MAXC LDS reads (a tree reduction with barriers in
    if (t == 0) { double mn = s_cv[0][0]; for (int i = 1; i < nc; i++) mn = (s_cv[0][i] < mn) ? s_cv[0][i] : mn; s_red[0] = mn; }   // its loop made k_post_ap miscompile)
    __syncthreads();
    const double rr = s_red[0];
    __syncthreads();
    return rr;
  };
  apply(false);
  if (phase == 0) {
    const double minv = min_cn_v();                                             // ICH:575
    // apply_restitution (ICH:470-491) scales the cn entries of _z in place
    for (int i = t; i < nc; i += T) { const double e = z[i] * cpar[4 * i + 2]; zb[i] = e; if (e > NEAR_ZERO_) s_any = 1; }
    __syncthreads();
    if (s_any) {
      apply(true);                                                              // ICH:581
      const double minv_plus = min_cn_v();
      second = (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO_);              // ICH:591
    }
  }
  for (int i = t; i < nc; i += T) for (int a = 0; a < 3; a++) d.Cv[((size_t)b * 3 + a) * ncm + start + i] = s_cv[a][i];
  if (t == 0 && phase == 0) d.again[b] = second ? 1 : 0;
  }   // ok
}

// apply_ap_model_to_connected_constraints around its LCPs (ICH-AP:36-92, 336-350) for the islands of model 2, in two
// small kernels (a single one needed spilled SGPRs, and came out of the compiler with a wrong execution mask after its
// min-reduction: tests/tools/ap_case.py found it).
// k_post_ap, phase 0: after the first solve -- impulses from z, propagate_impulse_data, constraint velocities,
// restitution, the second-solve test; phase 1: after the second solve.  It leaves need2[b] = 1 when the island's wrenches
// are final; k_apply_ap then does apply_impulses (ICH:676-745) and clears the flag.
__global__ __launch_bounds__(T)
void k_post_ap(Dev d, int r, int phase)
{
  const int b = blockIdx.x, t = threadIdx.x;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (!d.run[b]) return;
  if (phase == 1 && !d.again[b]) return;
  if (d.isl_model[(size_t)b * d.islmax + r] != 2) return;
  const int ncm = d.ncmax;
  __shared__ double s_c[3][MAXC], s_cv[3][MAXC];
  __shared__ double s_min;
  __shared__ int s_any;
  const double* z = d.z + (size_t)b * d.nmax;
  const bool ok = d.lst1[b] != 0;
  if (t == 0) {
    const int n = d.ncur[b];
    unsigned long long* c = d.cnt + (size_t)b * 5;
    c[0] += 1ull; c[1] += (unsigned long long)n; c[2] += (unsigned long long)d.piv1[b]; c[3] += 8ull * ((unsigned long long)n * n + 2ull * n);
    if (!ok) { d.status[b] |= MH_WORLD_LCP_FAILED; d.thrown[b] = 1; }        // throw std::exception() (ICH-AP:334): nothing is applied
    if (phase == 0) d.again[b] = 0;
    d.need2[b] = 0;
    s_any = 0;
  }
  if (!ok) return;
  const double* cpar = d.cpar + ((size_t)b * ncm + start) * 4;
  double* apw = d.apw + ((size_t)b * ncm + start) * 6;
  MMv m; m.G = d.G + (size_t)b * 6 * ncm * ncm; m.nc = nc;
  for (int i = t; i < nc; i += T) {
    s_cv[0][i] = d.Cv[((size_t)b * 3 + 0) * ncm + start + i];
    s_cv[1][i] = d.Cv[((size_t)b * 3 + 1) * ncm + start + i];
    s_cv[2][i] = d.Cv[((size_t)b * 3 + 2) * ncm + start + i];
    if (phase == 0) { double* w = apw + 6 * i; w[0] = 0.0; w[1] = 0.0; w[2] = 0.0; w[3] = 0.0; w[4] = 0.0; w[5] = 0.0; }   // ICH-AP:52-55
    s_c[0][i] = z[i];                                                         // ICH-AP:336-342
    s_c[1][i] = z[nc + i] - z[2 * nc + i];
    s_c[2][i] = z[3 * nc + i] - z[4 * nc + i];
  }
  __syncthreads();
  auto propagate = [&]() {                              // propagate_impulse_data (ICH:643-673)
    const mh_contact* C = d.contacts + (size_t)b * ncm;
    for (int i = t; i < nc; i += T) {
      const int ci = d.order[(size_t)b * ncm + start + i];
      const P3 p = ld3(C[ci].point), nn = ld3(C[ci].normal);
      P3 s, tt; basis(nn, s, tt);
      P3 j = nn * s_c[0][i]; j = j + s * s_c[1][i]; j = j + tt * s_c[2][i];
      const P3 a = cross3(p, j);
      double* w = apw + 6 * i;
      w[0] = w[0] + j.x; w[1] = w[1] + j.y; w[2] = w[2] + j.z; w[3] = w[3] + a.x; w[4] = w[4] + a.y; w[5] = w[5] + a.z;
      double* im = d.imp + ((size_t)b * ncm + ci) * 3;
      im[0] += s_c[0][i]; im[1] += s_c[1][i]; im[2] += s_c[2][i];
    }
    __syncthreads();
  };
  auto row_update = [&](int a, int i) -> double {       // C_a v(i) + sum_b (C_a X C_b^T c_b)(i), each product from 0 over ascending j
    double y = s_cv[a][i];
    for (int bb = 0; bb < 3; bb++) {
      double acc = 0.0;
      for (int j = 0; j < nc; j++) acc = acc + s_c[bb][j] * m.gab(a, bb, i, j);
      y = y + acc;
    }
    return y;
  };
  auto update_cv = [&]() {                              // update_constraint_velocities_from_impulses (ICH:427-464)
    for (int i = t; i < nc; i += T) {
      const double y0 = row_update(0, i), y1 = row_update(1, i), y2 = row_update(2, i);
      s_cv[0][i] = y0; s_cv[1][i] = y1; s_cv[2][i] = y2;
    }
    __syncthreads();
  };
  auto min_cn_v = [&]() -> double {                     // calc_min_constraint_velocity (ICH:413-424): one thread, at most MAXC reads
    if (t == 0) { double mn = s_cv[0][0]; for (int i = 1; i < nc; i++) mn = (s_cv[0][i] < mn) ? s_cv[0][i] : mn; s_min = mn; }
    __syncthreads();
    const double rr = s_min;
    __syncthreads();
    return rr;
  };
  propagate();
  bool second = false;
  if (phase == 0) {
    update_cv();                                                                // ICH-AP:61
    const double minv = min_cn_v();
    for (int i = t; i < nc; i += T) { const double e = s_c[0][i] * cpar[4 * i + 2]; s_c[0][i] = e; if (e > NEAR_ZERO_) s_any = 1; }   // apply_restitution(q) (ICH:497-525)
    __syncthreads();
    if (s_any) {
      for (int i = t; i < nc; i += T) { s_c[1][i] = 0.0; s_c[2][i] = 0.0; }
      __syncthreads();
      update_cv();
      const double minv_plus = min_cn_v();
      second = (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO_);              // ICH-AP:78
      if (!second) propagate();                                                 // ICH-AP:84 (on the other branch the restitution impulses are never propagated)
    }
    for (int i = t; i < nc; i += T) {
      d.Cv[((size_t)b * 3 + 0) * ncm + start + i] = s_cv[0][i];
      d.Cv[((size_t)b * 3 + 1) * ncm + start + i] = s_cv[1][i];
      d.Cv[((size_t)b * 3 + 2) * ncm + start + i] = s_cv[2][i];
    }
  }
  if (t == 0) { if (phase == 0) d.again[b] = second ? 1 : 0; d.need2[b] = second ? 0 : 1; }
}

// apply_impulses (ICH:676-745) for the A-P islands k_post_ap has finished: the accumulated wrench w of every contact on
// geom1's body, -w on geom2's, as generalized forces about the body centre, summed in contact order; v += M^-1 gj
__global__ __launch_bounds__(T)
void k_apply_ap(Dev d, int r)
{
  const int b = blockIdx.x, t = threadIdx.x;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (d.isl_model[(size_t)b * d.islmax + r] != 2 || !d.run[b] || !d.need2[b]) return;
  const int nb = d.nb, ncm = d.ncmax;
  __shared__ int s_b[2][MAXC];
  for (int i = t; i < nc; i += T) {
    const size_t ck = (size_t)b * ncm + start + i;
    s_b[0][i] = d.cbody[2 * ck]; s_b[1][i] = d.cbody[2 * ck + 1];
  }
  __syncthreads();
  if (t == 0) d.need2[b] = 0;
  double* st = d.state + (size_t)b * nb * 13;
  const double* apw = d.apw + ((size_t)b * ncm + start) * 6;
  const double* xinv = d.xinv + (size_t)b * nb * 10;
  for (int bb = t; bb < nb; bb += T) {
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0;
    bool seen = false;
    const P3 x = ld3(st + 13 * bb);
    for (int j = 0; j < nc; j++) {
      const int sd = (s_b[0][j] == bb) ? 0 : ((s_b[1][j] == bb) ? 1 : -1);
      if (sd < 0) continue;
      const double sg = (sd == 0) ? 1.0 : -1.0;
      const double* w = apw + 6 * j;
      P3 f; f.x = sg * w[0]; f.y = sg * w[1]; f.z = sg * w[2];
      P3 tq; tq.x = sg * w[3]; tq.y = sg * w[4]; tq.z = sg * w[5];
      tq = tq - cross3(x, f);
      if (!seen) { g0 = f.x; g1 = f.y; g2 = f.z; g3 = tq.x; g4 = tq.y; g5 = tq.z; seen = true; }
      else { g0 = g0 + f.x; g1 = g1 + f.y; g2 = g2 + f.z; g3 = g3 + tq.x; g4 = g4 + tq.y; g5 = g5 + tq.z; }
    }
    if (!seen) continue;
    const double* X = xinv + 10 * bb;
    double* v = st + 13 * bb + 7;
    v[0] = v[0] + X[0] * g0; v[1] = v[1] + X[0] * g1; v[2] = v[2] + X[0] * g2;
    v[3] = v[3] + ((X[1] * g3 + X[2] * g4) + X[3] * g5);
    v[4] = v[4] + ((X[4] * g3 + X[5] * g4) + X[6] * g5);
    v[5] = v[5] + ((X[7] * g3 + X[8] * g4) + X[9] * g5);
  }
}

// MH_CORE_STAB, islands of model 3 (contacts over bodies tied by implicit joints): set_unilateral_constraint_data's joint part
// (CStab:725-745, 826-880: the island's joints, Jfull, get_full_rank_implicit_constraints ICH:1698-1739) and
// ImpactConstraintHandler::compute_X (ICH:1590-1695): X = iM - 2G + G'MG, G = iM H' J iM, H' = J'(J iM J')^-1 -- then the rows
// X Cn' of the island's contacts and Cn X Cn' over them, replacing what k_prep / k_gram computed with the block-diagonal X.
// One workgroup per world; operation order: oracle World::build_bilateral / compute_X_general / compute_problem_data.
constexpr int KB = MH_IJOINT_MAX_BODIES, KM = MH_IJOINT_MAX_EQNS, KJ = MH_IJOINT_MAX_JOINTS, KG = 6 * MH_IJOINT_MAX_BODIES;
__global__ __launch_bounds__(T)
void k_bilat_X(Dev d, int r)
{
  const int b = blockIdx.x, t = threadIdx.x;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (d.isl_model[(size_t)b * d.islmax + r] != 3) return;
  const int nb = d.nb, ncm = d.ncmax;
  const double* st = d.state + (size_t)b * nb * 13;
  const int nbod = d.isl_nbod[(size_t)b * d.islmax + r], ngc = 6 * nbod;
  const int* bodies = d.isl_bod + ((size_t)b * d.islmax + r) * KB;
  JointTab jt; jt.nj = d.nj; jt.type = d.jtype; jt.in = d.jin; jt.out = d.jout; jt.anchor_in = d.janchor_in; jt.anchor_out = d.janchor_out;
  jt.vec_in = d.jvec_in; jt.vec_out = d.jvec_out;
  __shared__ double s_iM[KB * 36], s_Mg[KB * 36], s_w[2 * KJ * 36];
  __shared__ int s_jl[KJ], s_brow[2 * KJ], s_boff[2 * KJ], s_brows[2 * KJ], s_pos[KM], s_act[KM];
  __shared__ double s_A[KM * KM], s_L[KM * KM], s_C[KM], s_lam[KM];
  __shared__ int s_n[4];                                  // njl, m, k, drop
  double* T1 = d.bT1 + (size_t)b * KG * KG; double* T2 = d.bT2 + (size_t)b * KG * KG; double* T3 = d.bT3 + (size_t)b * KG * KG;
  double* T4 = d.bT4 + (size_t)b * KG * KG;
  double* JiM = d.bJiM + (size_t)b * KM * KG;
  double* XCn = d.bXCn + (size_t)b * ncm * KG;
  auto gc_of = [&](int body) { for (int i = 0; i < nbod; i++) if (bodies[i] == body) return 6 * i; return -1; };
  if (t == 0) {
    int njl = 0, m = 0, drop = 0;
    for (int j = 0; j < d.nj; j++) {
      const int a = d.jin[j], c = d.jout[j];
      const bool inisl = (a >= 0 && a < nb && gc_of(a) >= 0) || (c >= 0 && c < nb && gc_of(c) >= 0);
      if (!inisl) continue;
      const int rows = jt_rows(d.jtype[j]);
      if (njl == KJ || m + rows > KM) { drop = 1; break; }
      const int sides[2] = { a, c };
      for (int sd = 0; sd < 2; sd++) { s_brow[2 * njl + sd] = m; s_boff[2 * njl + sd] = (sides[sd] >= 0 && sides[sd] < nb) ? gc_of(sides[sd]) : -1; s_brows[2 * njl + sd] = rows; }
      s_jl[njl++] = j; m += rows;
    }
    s_n[0] = njl; s_n[1] = m; s_n[3] = drop;
    if (drop) { d.status[b] |= MH_WORLD_UNSUPPORTED; d.isl_model[(size_t)b * d.islmax + r] = 4; }
  }
  __syncthreads();
  if (s_n[3]) return;
  const int njl = s_n[0], m = s_n[1];
  for (int i = t; i < nbod; i += T) {
    const int bb = bodies[i];
    double xi[10], Jw[9];
    inv_inertia(st + 13 * bb, d.inertia + 3 * bb, d.mass[bb], xi, Jw);
    double* Bm = s_iM + 36 * i; double* Mm = s_Mg + 36 * i;
    for (int k = 0; k < 36; k++) { Bm[k] = 0.0; Mm[k] = 0.0; }
    for (int k = 0; k < 3; k++) { Bm[7 * k] = xi[0]; Mm[7 * k] = d.mass[bb]; }
    for (int rr = 0; rr < 3; rr++) for (int c = 0; c < 3; c++) { Bm[6 * (3 + rr) + 3 + c] = xi[1 + 3 * rr + c]; Mm[6 * (3 + rr) + 3 + c] = Jw[3 * rr + c]; }
  }
  for (int jl = t; jl < njl; jl += T) {
    double c6[6]; jt_eval(jt, st, nb, s_jl[jl], c6);
    for (int k = 0; k < s_brows[2 * jl]; k++) s_C[s_brow[2 * jl] + k] = c6[k];
  }
  for (int k = t; k < 2 * njl; k += T) if (s_boff[k] >= 0) jt_jac(jt, st, nb, s_jl[k >> 1], (k & 1) == 0, s_w + 36 * k);
  for (int e = t; e < m * ngc; e += T) JiM[e] = 0.0;
  __syncthreads();
  for (int e = t; e < m * m; e += T) {                    // J J'
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
    for (int e = t; e < s_brows[k] * 6; e += T) {
      const int rr = e / 6, c = e - 6 * rr;
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * k + 6 * rr + q] * Bm[6 * q + c];
      JiM[(size_t)(s_brow[k] + rr) * ngc + s_boff[k] + c] = acc;
    }
  }
  __syncthreads();
  if (t == 0) {                                          // greedy full-rank rows: incremental Cholesky of J J' - sqrt(eps) I
    int k = 0;
    for (int i = 0; i < m; i++) s_pos[i] = -1;
    for (int i = 0; i < m; i++) {
      if (k == ngc) break;
      for (int j = 0; j < k; j++) {
        double tv = s_A[i * m + s_act[j]];
        for (int p = 0; p < j; p++) tv = tv - s_L[k * m + p] * s_L[j * m + p];
        s_L[k * m + j] = tv / s_L[j * m + j];
      }
      double ajj = s_A[i * m + i] - NEAR_ZERO_;
      for (int p = 0; p < k; p++) ajj = ajj - s_L[k * m + p] * s_L[k * m + p];
      if (ajj > 0.0) { s_L[k * m + k] = sqrt(ajj); s_act[k] = i; s_pos[i] = k; k++; }
    }
    s_n[2] = k;
  }
  __syncthreads();
  const int k = s_n[2];
  for (int e = t; e < k * k; e += T) {                    // J iM J' on the active rows -> s_L (row r, column c at r * m + c)
    const int rr = e / k, c = e - rr * k, row = s_act[rr];
    double tot = 0.0;
    for (int kb = 0; kb < 2 * njl; kb++) {
      if (s_boff[kb] < 0 || row < s_brow[kb] || row >= s_brow[kb] + s_brows[kb]) continue;
      double acc = 0.0;
      for (int q = 0; q < 6; q++) acc = acc + s_w[36 * kb + 6 * (row - s_brow[kb]) + q] * JiM[(size_t)s_act[c] * ngc + s_boff[kb] + q];
      tot = tot + acc;
    }
    s_L[rr * m + c] = tot;
  }
  __syncthreads();
  if (t == 0) {                                          // dpotf2 'L'; lambda = (J iM J')^-1 C
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
    if (!ok) { s_n[3] = 1; d.status[b] |= MH_WORLD_UNSUPPORTED; d.isl_model[(size_t)b * d.islmax + r] = 4; }
    else {
      for (int rr = 0; rr < k; rr++) s_lam[rr] = s_C[s_act[rr]];
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
  if (s_n[3]) return;
  for (int i = t; i < k; i += T) { d.bact[(size_t)b * KM + i] = s_act[i]; d.blam[(size_t)b * KM + i] = s_lam[i]; }
  if (t == 0) d.bk[b] = k;
  // inverse_SPD(J iM J') (ICH:1664-1665): column c = the solve of L L' x = e_c, then the lower triangle mirrored -> s_A (i + m c)
  for (int c = t; c < k; c += T) {
    double* e = s_A + (size_t)m * c;
    for (int i = 0; i < k; i++) e[i] = (i == c) ? 1.0 : 0.0;
    for (int kk = 0; kk < k; kk++) {
      e[kk] = e[kk] / s_L[kk * m + kk];
      const double bk = e[kk];
      for (int i = kk + 1; i < k; i++) e[i] = e[i] - bk * s_L[i * m + kk];
    }
    for (int kk = k - 1; kk >= 0; kk--) {
      double tv = e[kk];
      for (int i = kk + 1; i < k; i++) tv = tv - s_L[i * m + kk] * e[i];
      e[kk] = tv / s_L[kk * m + kk];
    }
  }
  __syncthreads();
  for (int e = t; e < k * k; e += T) { const int c = e / k, i = e - c * k; if (i > c) s_A[c + m * i] = s_A[i + m * c]; }
  __syncthreads();
  // H' = J' Ainv (ngc x k) -> T4
  for (int e = t; e < ngc * k; e += T) {
    const int g = e / k, c = e - g * k;
    double acc = 0.0;
    for (int kb = 0; kb < 2 * njl; kb++) {
      if (s_boff[kb] < 0 || g < s_boff[kb] || g >= s_boff[kb] + 6) continue;
      for (int rr = 0; rr < s_brows[kb]; rr++) { const int pr = s_pos[s_brow[kb] + rr]; if (pr >= 0) acc = acc + s_w[36 * kb + 6 * rr + (g - s_boff[kb])] * s_A[pr + m * c]; }
    }
    T4[(size_t)g * k + c] = acc;
  }
  __syncthreads();
  for (int e = t; e < ngc * ngc; e += T) {                // H' J iM -> T1
    const int g = e / ngc, h = e - g * ngc;
    double acc = 0.0;
    for (int c = 0; c < k; c++) acc = acc + T4[(size_t)g * k + c] * JiM[(size_t)s_act[c] * ngc + h];
    T1[(size_t)g * ngc + h] = acc;
  }
  __syncthreads();
  for (int e = t; e < ngc * ngc; e += T) {                // G = iM (H' J iM) -> T2
    const int g = e / ngc, h = e - g * ngc, i = g / 6, rr = g - 6 * i;
    double acc = 0.0;
    for (int q = 0; q < 6; q++) acc = acc + s_iM[36 * i + 6 * rr + q] * T1[(size_t)(6 * i + q) * ngc + h];
    T2[(size_t)g * ngc + h] = acc;
  }
  __syncthreads();
  for (int e = t; e < ngc * ngc; e += T) {                // M G -> T3
    const int g = e / ngc, h = e - g * ngc, i = g / 6, rr = g - 6 * i;
    double acc = 0.0;
    for (int q = 0; q < 6; q++) acc = acc + s_Mg[36 * i + 6 * rr + q] * T2[(size_t)(6 * i + q) * ngc + h];
    T3[(size_t)g * ngc + h] = acc;
  }
  __syncthreads();
  for (int e = t; e < ngc * ngc; e += T) {                // X = (iM - 2 G) + G' M G -> T1
    const int g = e / ngc, h = e - g * ngc;
    double acc = 0.0;
    for (int p = 0; p < ngc; p++) acc = acc + T2[(size_t)p * ngc + g] * T3[(size_t)p * ngc + h];
    const double x0 = (g / 6 == h / 6) ? s_iM[36 * (g / 6) + 6 * (g - 6 * (g / 6)) + (h - 6 * (h / 6))] : 0.0;
    T1[(size_t)g * ngc + h] = (x0 - 2.0 * T2[(size_t)g * ngc + h]) + acc;
  }
  __syncthreads();
  // X Cn' of the island's contacts (compute_problem_data: sum over the row's blocks of 6-term products), then Cn X Cn'
  for (int e = t; e < nc * ngc; e += T) {
    const int i = e / ngc, g = e - i * ngc;
    const size_t ck = (size_t)b * ncm + start + i;
    double res = 0.0;
    for (int sd = 0; sd < 2; sd++) {
      const int body = d.cbody[2 * ck + sd];
      if (body < 0) continue;
      const int off = gc_of(body);
      const double* w = d.W + ck * 36 + sd * 6;
      double tmp = 0.0;
      for (int q = 0; q < 6; q++) tmp = tmp + w[q] * T1[(size_t)(off + q) * ngc + g];
      res = res + tmp;
    }
    XCn[(size_t)i * KG + g] = res;
  }
  __syncthreads();
  double* G = d.G + (size_t)b * 6 * ncm * ncm;
  for (int e = t; e < nc * nc; e += T) {
    const int i = e / nc, j = e - i * nc;
    const size_t ck = (size_t)b * ncm + start + i;
    double res = 0.0;
    for (int sd = 0; sd < 2; sd++) {
      const int body = d.cbody[2 * ck + sd];
      if (body < 0) continue;
      const int off = gc_of(body);
      const double* w = d.W + ck * 36 + sd * 6;
      double tmp = 0.0;
      for (int q = 0; q < 6; q++) tmp = tmp + w[q] * XCn[(size_t)j * KG + off + q];
      res = res + tmp;
    }
    G[(size_t)i * nc + j] = res;
  }
}

// determine_dq's tail (CStab:958-969): update_from_stacked with cn = z -- the bodies' velocities become X Cn^T z
__global__ __launch_bounds__(T)
void k_stab_apply(Dev d, int r)
{
  const int b = blockIdx.x, t = threadIdx.x;
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (!d.run[b]) return;
  const int nb = d.nb, ncm = d.ncmax;
  __shared__ double s_c[MAXC];
  __shared__ int s_b[2][MAXC];
  double* st = d.state + (size_t)b * nb * 13;
  const double* z = d.z + (size_t)b * d.nmax;
  if (t == 0) { account(d, b, nc); d.cnt[(size_t)b * 5 + 4] += (unsigned long long)nc; }
  // cn = z, whatever z holds (even after a failed solve)
  for (int i = t; i < nc; i += T) {
    const size_t ck = (size_t)b * ncm + start + i;
    s_b[0][i] = d.cbody[2 * ck]; s_b[1][i] = d.cbody[2 * ck + 1];
    s_c[i] = z[i];
    d.imp[((size_t)b * ncm + d.order[ck]) * 3] += z[i];
  }
  __syncthreads();
  if (d.isl_model[(size_t)b * d.islmax + r] == 3) {
    // implicit joints in the island: dv = X Cn' cn with the general X, then dv -= iM J' lambda (ICH:343-374)
    const int nbod = d.isl_nbod[(size_t)b * d.islmax + r], ngc = 6 * nbod, k = d.bk[b];
    const int* bodies = d.isl_bod + ((size_t)b * d.islmax + r) * MH_IJOINT_MAX_BODIES;
    const double* XCn = d.bXCn + (size_t)b * ncm * 96;
    const double* JiM = d.bJiM + (size_t)b * 48 * 96;
    for (int g = t; g < ngc; g += T) {
      double tmp = 0.0;
      for (int j = 0; j < nc; j++) tmp = tmp + s_c[j] * XCn[(size_t)j * 96 + g];
      double acc = 0.0;
      for (int rr = 0; rr < k; rr++) acc = acc + JiM[(size_t)d.bact[(size_t)b * 48 + rr] * ngc + g] * d.blam[(size_t)b * 48 + rr];
      double* v = st + 13 * bodies[g / 6] + 7 + (g - 6 * (g / 6));
      double x = *v + tmp;
      x = x - acc;
      *v = x;
    }
    return;
  }
  for (int e = t; e < nb * 6; e += T) {
    const int bb = e / 6, q = e - bb * 6;
    double tmp = 0.0;
    bool touched = false;
    for (int j = 0; j < nc; j++) {
      const int sd = (s_b[0][j] == bb) ? 0 : ((s_b[1][j] == bb) ? 1 : -1);
      if (sd < 0) continue;
      touched = true;
      tmp = tmp + s_c[j] * d.XJ[((size_t)b * ncm + start + j) * 36 + sd * 6 + q];
    }
    if (touched) st[13 * bb + 7 + q] = st[13 * bb + 7 + q] + tmp;
  }
}

// ImpactConstraintHandler::apply_no_slip_model[_to_connected_constraints] (ICH:236-295, 1009-1417), contacts only, for the
// islands k_prep marked (every mu >= 100).  One wavefront per world: the dense algebra is tiny (at most 2 x MH_NOSLIP_MAX
// tangent rows) and sequential by construction (greedy Cholesky tests of a growing matrix), lane 0 does it serially over
// LDS in the operation order of oracle/world.hpp::apply_no_slip_model; the (nc x nc) LCP runs wave-wide on the persistent
// _v (lcp_fast, then the Lemke ladder: ICH:1239, 1281), impulses / constraint velocities / restitution lane-parallel.
constexpr int NSC = MH_NOSLIP_MAX, NSM = 2 * MH_NOSLIP_MAX;
MH_DEV bool ns_chol_factor(int n, double* A) {                      // oracle/linalg.hpp chol_factor (dpotf2 'L')
  for (int j = 0; j < n; j++) {
    double ajj = A[j + n*j];
    for (int k = 0; k < j; k++) ajj = ajj - A[j + n*k] * A[j + n*k];
    if (!(ajj > 0.0)) return false;
    ajj = sqrt(ajj);
    A[j + n*j] = ajj;
    for (int i = j+1; i < n; i++) {
      double t = A[i + n*j];
      for (int k = 0; k < j; k++) t = t - A[i + n*k] * A[j + n*k];
      A[i + n*j] = t / ajj;
    }
  }
  return true;
}
MH_DEV void ns_chol_solve(int n, const double* L, double* b) {      // oracle/linalg.hpp chol_solve
  for (int k = 0; k < n; k++) {
    b[k] = b[k] / L[k + n*k];
    const double bk = b[k];
    for (int i = k+1; i < n; i++) b[i] = b[i] - bk * L[i + n*k];
  }
  for (int k = n-1; k >= 0; k--) {
    double t = b[k];
    for (int i = k+1; i < n; i++) t = t - L[i + n*k] * b[i];
    b[k] = t / L[k + n*k];
  }
}

__global__ __launch_bounds__(64)
void k_noslip(Dev d, int r, mh::Pow10Table p10)
{
  const int b = blockIdx.x, lane = mh::lane_id();
  int start, nc;
  if (!island(d, b, r, start, nc)) return;
  if (d.isl_model[(size_t)b * d.islmax + r] != 1) return;
  if (nc > NSC) { if (lane == 0) d.status[b] |= MH_WORLD_UNSUPPORTED; return; }    // _v's warm start is kept for up to MH_NOSLIP_MAX contacts
  __shared__ double Y[NSM * NSM], QX[NSC * NSM], Wm[NSM * NSC], MM[NSC * NSC], Alu[NSC * NSC], art[NSC], YXv[NSM], t2[NSM], col[NSM], qq[NSC];
  __shared__ double s_c[3][NSC], s_cv[3][NSC];
  __shared__ int S[NSC], Tt[NSC], s_b[2][NSC], s_m[2];
  const int nb = d.nb, ncm = d.ncmax;
  double* st = d.state + (size_t)b * nb * 13;
  const double* G = d.G + (size_t)b * 6 * ncm * ncm;                 // blocks nn ns nt ss st tt, nc x nc row-major each
  const double* cpar = d.cpar + ((size_t)b * ncm + start) * 4;
  auto g = [&](int blk, int i, int j) { return G[((size_t)blk * nc + i) * nc + j]; };
  if (lane < nc) {
    const size_t ck = (size_t)b * ncm + start + lane;
    s_b[0][lane] = d.cbody[2 * ck]; s_b[1][lane] = d.cbody[2 * ck + 1];
    for (int a = 0; a < 3; a++) s_cv[a][lane] = d.Cv[((size_t)b * 3 + a) * ncm + start + lane];
  }
  __syncthreads();
  if (lane == 0) {
    int ns = 0, nt = 0;
    auto build_Y = [&](bool skew) -> int {                            // ICH:1098-1111
      const int m = ns + nt;
      for (int a = 0; a < ns; a++) for (int c = 0; c < ns; c++) Y[a + m*c] = g(3, S[a], S[c]);
      for (int a = 0; a < nt; a++) for (int c = 0; c < nt; c++) Y[(ns + a) + m*(ns + c)] = g(5, Tt[a], Tt[c]);
      for (int a = 0; a < ns; a++) for (int c = 0; c < nt; c++) { const double v = g(4, S[a], Tt[c]); Y[a + m*(ns + c)] = v; Y[(ns + c) + m*a] = v; }
      if (skew) for (int j = 0; j < m; j++) Y[j + m*j] = Y[j + m*j] - NEAR_ZERO_;
      return m;
    };
    for (int i = 0; i < nc; i++) {                                    // greedy largest non-singular tangent set (ICH:1087-1145)
      S[ns] = i; ns++;
      int m = build_Y(true);
      if (!ns_chol_factor(m, Y)) ns--;
      Tt[nt] = i; nt++;
      m = build_Y(true);
      if (!ns_chol_factor(m, Y)) nt--;
    }
    const int m = build_Y(false);                                     // ICH:1166-1183
    const bool ok = ns_chol_factor(m, Y);
    if (ok) {
      for (int i = 0; i < nc; i++) {                                  // Q X X' (ICH:1198-1203)
        for (int a = 0; a < ns; a++) QX[i*m + a] = g(1, i, S[a]);
        for (int a = 0; a < nt; a++) QX[i*m + ns + a] = g(2, i, Tt[a]);
      }
      for (int j = 0; j < nc; j++) {                                  // W = Y^-1 (Q X X')' (ICH:1210-1211)
        for (int a = 0; a < m; a++) col[a] = QX[j*m + a];
        ns_chol_solve(m, Y, col);
        for (int a = 0; a < m; a++) Wm[a + m*j] = col[a];
      }
      for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {     // MM = Cn X Cn' - QX W (ICH:1190-1215), column-major
        double acc = 0.0;
        for (int a = 0; a < m; a++) acc = acc + QX[i*m + a] * Wm[a + m*j];
        MM[i + nc*j] = g(0, i, j) - acc;
      }
      for (int a = 0; a < ns; a++) YXv[a] = s_cv[1][S[a]];            // Y^-1 [Cs v(S); Ct v(T)] (ICH:1222-1232)
      for (int a = 0; a < nt; a++) YXv[ns + a] = s_cv[2][Tt[a]];
      ns_chol_solve(m, Y, YXv);
      for (int i = 0; i < nc; i++) {                                  // qq = Cn v - QX YXv (ICH:1217-1236)
        double acc = 0.0;
        for (int a = 0; a < m; a++) acc = acc + QX[i*m + a] * YXv[a];
        qq[i] = s_cv[0][i] - acc;
      }
    }
    s_m[0] = ok ? m : -1; s_m[1] = ns;
  }
  __syncthreads();
  const int m = s_m[0], ns = s_m[1], nt = m - ns;
  if (m < 0) { if (lane == 0) { d.status[b] |= MH_WORLD_LCP_FAILED; d.thrown[b] = 1; } return; }       // assert(success), ICH:1184-1186
  // lcp_fast(_MM, _qq, _v), then the Lemke ladder
  const bool valid = lane < nc;
  double nrm0 = 0.0;
  for (int e = lane; e < nc * nc; e += 64) { const double a = fabs(MM[e]); nrm0 = (a > nrm0) ? a : nrm0; }
  nrm0 = mh::wave_max(nrm0);
  const double dii = valid ? MM[lane + nc * lane] : 0.0;
  const double qi = valid ? qq[lane] : 0.0;
  int zsize = mh::uni(d.vns_size[b]);
  double zi = (valid && zsize == nc) ? d.vns[(size_t)b * NSC + lane] : 0.0;
  if (lane == 0) mh::g_lcp_prof_on = 0;
  __syncthreads();
  mh::WaveRand rng; rng.load(d.rng + (size_t)b * MH_RAND_WORDS);
  mh::DenseLds Md; Md.M = MM; Md.n = nc;
  mh::LuScratch Sc; Sc.small = Alu; Sc.ka = nc; Sc.big = Alu;
  mh::Trace tr; tr.buf = nullptr; tr.cap = 0; tr.len = 0;
  mh::LcpParams P; P.kind = MH_LCP_FAST; P.min_exp = -20; P.step_exp = 1u; P.max_exp = 1; P.piv_tol = -1.0; P.zero_tol = -1.0;
  unsigned piv = 0, total = 0;
  bool ok = mh::lcp_solve_wave(P, p10, nc, Md, Sc, art, nrm0, dii, qi, zi, zsize, rng, piv, tr);
  total += piv;
  if (!ok) { P.kind = MH_LCP_LEMKE_REG; ok = mh::lcp_solve_wave(P, p10, nc, Md, Sc, art, nrm0, dii, qi, zi, zsize, rng, piv, tr); total += piv; }
  rng.store(d.rng + (size_t)b * MH_RAND_WORDS);
  if (lane == 0) {
    unsigned long long* c = d.cnt + (size_t)b * 5;
    c[0] += 1ull; c[1] += (unsigned long long)nc; c[2] += (unsigned long long)total; c[3] += 8ull * ((unsigned long long)nc * nc + 2ull * nc);
    if (!ok) { d.status[b] |= MH_WORLD_LCP_FAILED; d.thrown[b] = 1; }          // std::runtime_error("Unable to solve constraint LCP!")
  }
  if (!ok) return;
  if (valid) { d.vns[(size_t)b * NSC + lane] = zi; s_c[0][lane] = zi; s_c[1][lane] = 0.0; s_c[2][lane] = 0.0; }
  if (lane == 0) d.vns_size[b] = nc;
  __syncthreads();
  if (lane == 0) {                                                   // [cs; ct] = -(Y^-1 X v + Y^-1 (QX)' cn) (ICH:1293-1308)
    for (int a = 0; a < m; a++) { double acc = 0.0; for (int i = 0; i < nc; i++) acc = acc + QX[i*m + a] * s_c[0][i]; t2[a] = acc; }
    ns_chol_solve(m, Y, t2);
    for (int a = 0; a < ns; a++) s_c[1][S[a]] = -(YXv[a] + t2[a]);
    for (int a = 0; a < nt; a++) s_c[2][Tt[a]] = -(YXv[ns + a] + t2[ns + a]);
  }
  __syncthreads();
  MMv mg; mg.G = G; mg.nc = nc;
  // update_from_stacked (ICH:298-410) + update_constraint_velocities_from_impulses (ICH:427-464) for the impulses in s_c
  auto apply = [&]() {
    if (valid) { double* im = d.imp + ((size_t)b * ncm + d.order[(size_t)b * ncm + start + lane]) * 3; im[0] += s_c[0][lane]; im[1] += s_c[1][lane]; im[2] += s_c[2][lane]; }
    for (int e = lane; e < nb * 6; e += 64) {
      const int bb = e / 6, q = e - bb * 6;
      double dv = 0.0;
      bool touched = false;
      for (int dd = 0; dd < 3; dd++) {
        double tmp = 0.0;
        for (int j = 0; j < nc; j++) {
          const int sd = (s_b[0][j] == bb) ? 0 : ((s_b[1][j] == bb) ? 1 : -1);
          if (sd < 0) continue;
          touched = true;
          tmp = tmp + s_c[dd][j] * d.XJ[((size_t)b * ncm + start + j) * 36 + dd * 12 + sd * 6 + q];
        }
        dv = (dd == 0) ? tmp : dv + tmp;
      }
      if (touched) st[13 * bb + 7 + q] = st[13 * bb + 7 + q] + dv;
    }
    double y3[3] = { 0.0, 0.0, 0.0 };
    if (valid) for (int a = 0; a < 3; a++) {
      double y = s_cv[a][lane];
      for (int bb = 0; bb < 3; bb++) { double acc = 0.0; for (int j = 0; j < nc; j++) acc = acc + s_c[bb][j] * mg.gab(a, bb, lane, j); y = y + acc; }
      y3[a] = y;
    }
    __syncthreads();
    if (valid) for (int a = 0; a < 3; a++) s_cv[a][lane] = y3[a];
    __syncthreads();
  };
  auto min_cn_v = [&]() -> double { double mn = s_cv[0][0]; for (int i = 1; i < nc; i++) mn = (s_cv[0][i] < mn) ? s_cv[0][i] : mn; return mn; };
  apply();                                                            // ICH:1351-1385, :265
  const double minv = min_cn_v();
  // apply_restitution(q) (ICH:497-525)
  double cn = 0.0; bool ch = false;
  if (valid) { cn = s_c[0][lane] * cpar[4 * lane + 2]; ch = cn > NEAR_ZERO_; }
  const bool changed = mh::ballot(ch) != 0ull;
  __syncthreads();
  if (valid) s_c[0][lane] = cn;
  if (changed) {
    if (valid) { s_c[1][lane] = 0.0; s_c[2][lane] = 0.0; }
    __syncthreads();
    apply();                                                          // update_from_stacked(q) (ICH:274)
    const double minv_plus = min_cn_v();
    // ICH:284-291 would re-solve and then read the Drumwright-Shell solver's _z, which this path never sized
    if (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO_ && lane == 0) d.status[b] |= MH_WORLD_UNSUPPORTED;
  }
  if (valid) for (int a = 0; a < 3; a++) d.Cv[((size_t)b * 3 + a) * ncm + start + lane] = s_cv[a][lane];
}

// ImpactToleranceException test (ICH:157-167) over the contacts of the processed islands: any still approaching
// faster than NEAR_ZERO
// the exception unwinds process_constraints (mh_imp_core.h: thrown): the islands after the failing one are not processed, the tolerance check is not reached
__global__ void k_unwind(Dev d)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < d.B && d.thrown[b]) d.nisl[b] = 0;
}

__global__ __launch_bounds__(T)
void k_finish(Dev d)
{
  const int b = blockIdx.x, t = threadIdx.x;
  const int nisl = d.nisl[b];
  if (nisl == 0) return;
  const int nb = d.nb, ncm = d.ncmax;
  const int kept = d.isl_start[(size_t)b * d.islmax + nisl - 1] + d.isl_len[(size_t)b * d.islmax + nisl - 1];
  const double* st = d.state + (size_t)b * nb * 13;
  const mh_contact* C = d.contacts + (size_t)b * ncm;
  bool bad = false;
  for (int k = t; k < kept; k += T) {
    const int i = d.order[(size_t)b * ncm + k];
    int g1 = C[i].body1, g2 = C[i].body2;
    if (g1 < 0 || g1 >= nb) g1 = -1;
    if (g2 < 0 || g2 >= nb) g2 = -1;
    const P3 p = ld3(C[i].point), nn = ld3(C[i].normal);
    if (dot3(nn, point_vel(st, g1, p) - point_vel(st, g2, p)) < -NEAR_ZERO_) bad = true;
  }
  if (bad) atomicOr(&d.status[b], MH_WORLD_IMPACT_TOL);
}

}} // namespace mh::imp

// ---------------------------------------------------------------------------------------------------------
extern "C" {

void mh_imp_core_destroy(mh_imp_core* c)
{
  if (!c) return;
  for (int i = 0; i < c->nallocs; i++) if (c->allocs[i]) (void)hipFree(c->allocs[i]);
  c->nallocs = 0;
  if (c->s2) { (void)hipStreamSynchronize((hipStream_t)c->s2); (void)hipStreamDestroy((hipStream_t)c->s2); (void)hipEventDestroy((hipEvent_t)c->ev0); (void)hipEventDestroy((hipEvent_t)c->ev1); c->s2 = nullptr; }
  { void* ps[] = { c->t_wsd, c->t_wsi, c->t_z, c->t_st, c->t_piv, c->t_zsz, c->t_rng, c->t_work, c->solved_at };
    for (void* q : ps) if (q) (void)hipFree(q);
    c->t_wsd = nullptr; c->t_wsi = nullptr; c->t_z = nullptr; c->t_st = nullptr; c->t_piv = nullptr; c->t_zsz = nullptr; c->t_rng = nullptr; c->t_work = nullptr; c->solved_at = nullptr; c->t_cap = 0; }
  if (c->hmax) { (void)hipHostFree(c->hmax); c->hmax = nullptr; }
}

int mh_imp_core_create(mh_imp_core* c, int B, int nb, int ncmax, int nk, int nmax)
{
  namespace im = mh::imp;
  std::memset(c, 0, sizeof(*c));
  if (nb > im::MAXB) return fail(MH_ERR_INVALID_ARG, "nb = %d > %d bodies per world", nb, im::MAXB);
  if (ncmax > im::MAXC) return fail(MH_ERR_INVALID_ARG, "nc = %d > %d contacts per world", ncmax, im::MAXC);
  if (nk < 4 || (nk & 1)) return fail(MH_ERR_INVALID_ARG, "nk must be even and >= 4 (ContactParameters.cpp:128-135), got %d", nk);
  if (nmax < 1 || nmax > MH_LCP_MAX_N_BLOCK) return fail(MH_ERR_UNSUPPORTED_N, "LCP capacity n = %d outside [1, %d]", nmax, MH_LCP_MAX_N_BLOCK);
  c->B = B; c->nb = nb; c->ncmax = ncmax; c->nk = nk; c->kh = nk / 2; c->nmax = nmax;
  c->islmax = nb < ncmax ? nb : ncmax;
  bool okall = true;
  auto A = [&](size_t bytes, bool zero) -> void* {
    void* p = nullptr;
    if (!okall) return nullptr;
    if (c->nallocs >= (int)(sizeof(c->allocs) / sizeof(c->allocs[0])) || hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { okall = false; return nullptr; }
    c->allocs[c->nallocs++] = p;
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) okall = false;
    return p;
  };
  const size_t sB = (size_t)B, sn = (size_t)nmax, snc = (size_t)ncmax;
  c->order = (int*)A(sB * snc * 4, true); c->cbody = (int*)A(sB * snc * 8, true); c->cpar = (double*)A(sB * snc * 32, true);
  c->W = (double*)A(sB * snc * 36 * 8, true); c->XJ = (double*)A(sB * snc * 36 * 8, true);
  c->Cv = (double*)A(sB * 3 * snc * 8, true); c->xinv = (double*)A(sB * nb * 80, true);
  c->thrown = (int*)A(sB * 4, true);
  c->nisl = (int*)A(sB * 4, true); c->isl_start = (int*)A(sB * c->islmax * 4, true); c->isl_len = (int*)A(sB * c->islmax * 4, true);
  c->isl_model = (int*)A(sB * c->islmax * 4, true);
  c->vns = (double*)A(sB * MH_NOSLIP_MAX * 8, true); c->vns_size = (int*)A(sB * 4, true);
  c->maxisl = (int*)A(4, true);
  c->G = (double*)A(sB * 6 * snc * snc * 8, true);
  c->MM = (double*)A(sB * sn * sn * 8, false); c->qq = (double*)A(sB * sn * 8, true);
  c->z = (double*)A(sB * sn * 8, true); c->zsz = (int*)A(sB * 4, true); c->ncur = (int*)A(sB * 4, true);
  c->zlast = (double*)A(sB * sn * 8, true); c->zbuf = (double*)A(sB * sn * 8, true);
  c->zlast_size = (int*)A(sB * 4, true); c->zbuf_size = (int*)A(sB * 4, true); c->zbuf_cap = (int*)A(sB * 4, true);
  c->run = (int*)A(sB * 4, true); c->need2 = (int*)A(sB * 4, true); c->again = (int*)A(sB * 4, true);
  c->lst1 = (int*)A(sB * 4, true); c->lst2 = (int*)A(sB * 4, true);
  c->piv1 = (unsigned*)A(sB * 4, true); c->piv2 = (unsigned*)A(sB * 4, true);
  c->imp = (double*)A(sB * snc * 3 * 8, true);
  c->cnt = (unsigned long long*)A(sB * 5 * 8, true);
  c->work = (double*)A(sB * MH_WORK * 8, true);
  double* dcos = (double*)A(c->kh * 8, false); double* dsin = (double*)A(c->kh * 8, false);
  c->nk4 = (nk > 4) ? (nk + 4) / 4 : 1;                                     // ICH-AP:113-118
  double* acos_ = (double*)A(c->nk4 * 8, false); double* asin_ = (double*)A(c->nk4 * 8, false);
  c->apw = (double*)A(sB * snc * 6 * 8, true);
  if (nmax > MH_LCP_MAX_N_WAVE) {
    c->ws_d = (double*)A(sB * (sn * sn + 5 * sn) * 8, false);
    c->ws_i = (int*)A(sB * 4 * sn * 4, false);
  }
  if (okall && hipHostMalloc((void**)&c->hmax, sizeof(int)) != hipSuccess) okall = false;
  if (!okall) {
    mh_imp_core_destroy(c);
    return fail(MH_ERR_HIP, "device allocation failed (B = %d, n = %d: %.1f GB for _MM + LU workspace)", B, nmax, 2.0 * sB * sn * sn * 8 / 1e9);
  }
  c->fcos = dcos; c->fsin = dsin;
  // friction polygon directions (ICH-QP:452-470) with the host libm, like the world kernel's table
  std::vector<double> hc(c->kh), hs(c->kh);
  for (int j = 0; j < c->kh; j++) { const double theta = (double)j / (c->kh - 1) * M_PI_2; hc[j] = std::cos(theta); hs[j] = std::sin(theta); }
  // Anitescu-Potra polygon rows (ICH-AP:250-295): cos / sin(pi k / (2 nk4)); one row of ones for a 4-edge cone
  std::vector<double> ac(c->nk4), as(c->nk4);
  for (int k = 0; k < c->nk4; k++) { ac[k] = (nk > 4) ? std::cos((M_PI * k) / (2.0 * c->nk4)) : 1.0; as[k] = (nk > 4) ? std::sin((M_PI * k) / (2.0 * c->nk4)) : 1.0; }
  c->apcos = acos_; c->apsin = asin_;
  if (hipMemcpy(acos_, ac.data(), c->nk4 * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(asin_, as.data(), c->nk4 * 8, hipMemcpyHostToDevice) != hipSuccess
      || hipMemcpy(dcos, hc.data(), c->kh * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dsin, hs.data(), c->kh * 8, hipMemcpyHostToDevice) != hipSuccess) {
    mh_imp_core_destroy(c);
    return fail(MH_ERR_HIP, "upload of the friction table failed");
  }
  return MH_OK;
}

int mh_imp_core_enable_joints(mh_imp_core* c, int nj, const int* jtype, const int* jin, const int* jout, const double* janchor_in,
                              const double* janchor_out, const double* jvec_in, const double* jvec_out, const unsigned char* jointed)
{
  c->nj = nj; c->jtype = jtype; c->jin = jin; c->jout = jout; c->janchor_in = janchor_in; c->janchor_out = janchor_out;
  c->jvec_in = jvec_in; c->jvec_out = jvec_out; c->jointed = jointed;
  bool okall = true;
  auto A = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (!okall) return nullptr;
    if (c->nallocs >= (int)(sizeof(c->allocs) / sizeof(c->allocs[0])) || hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { okall = false; return nullptr; }
    c->allocs[c->nallocs++] = p;
    if (hipMemset(p, 0, bytes) != hipSuccess) okall = false;
    return p;
  };
  const size_t sB = (size_t)c->B, KG = 6 * MH_IJOINT_MAX_BODIES, KM = MH_IJOINT_MAX_EQNS;
  c->isl_nbod = (int*)A(sB * c->islmax * 4); c->isl_bod = (int*)A(sB * c->islmax * MH_IJOINT_MAX_BODIES * 4);
  c->bT1 = (double*)A(sB * KG * KG * 8); c->bT2 = (double*)A(sB * KG * KG * 8); c->bT3 = (double*)A(sB * KG * KG * 8); c->bT4 = (double*)A(sB * KG * KG * 8);
  c->bXCn = (double*)A(sB * c->ncmax * KG * 8);
  c->bJiM = (double*)A(sB * KM * KG * 8); c->blam = (double*)A(sB * KM * 8); c->bact = (int*)A(sB * KM * 4); c->bk = (int*)A(sB * 4);
  if (!okall) { c->bT1 = nullptr; return fail(MH_ERR_HIP, "device allocation of the joint scratch failed (B = %d)", c->B); }
  return MH_OK;
}

// ---- the Lemke ladder as (world, attempt) tasks ------------------------------------------------------------------------------
// lcp_lemke_regularized tries lambda = 0, then 10^min_exp ... one after the other until an attempt succeeds and verifies (LCP.cpp:353-487).
// The attempts are independent of one another (mh_lcp_block.h, LadderTask), so they run as B x R workgroups dispatched attempt-major:
// every world's lower attempts first, higher ones as slots free up, none above an attempt already known to have succeeded.  The slowest
// world of a batch then no longer runs its 8-12 attempts one after the other while the rest of the chip idles.  k_ladder_select takes,
// per world, the first successful attempt in ladder order and reproduces what the sequence would have left behind: z, its size, the
// summed pivot count, and the rand() stream (lcp_lemke draws n values whenever it is entered with z.size() != n, LCP.cpp:618-621).
namespace mh { namespace imp {
__global__ __launch_bounds__(T)
void k_ladder_select(Dev d, const int* __restrict__ need, int R, int* __restrict__ lst, unsigned* __restrict__ piv)
{
  const int w = blockIdx.x, t = threadIdx.x, B = d.B;
  if (!need[w]) return;
  const int n = d.ncur[w];
  if (n <= MH_LCP_MAX_N_WAVE) return;                            // (the wave solver ran this one's ladder)
  __shared__ int s_sel;
  if (t == 0) {
    int zs = d.zsz[w], chosen = -1, last = 0;
    unsigned total = 0u; long draws = 0; double wf = 0.0, wb = 0.0;
    for (int r = 0; r < R; r++) {
      const size_t i = (size_t)r * B + w;
      const int st = d.t_st[i];
      if (st < 0) break;                                         // not run: only above a successful attempt, never reached (defensive)
      const int zo = d.t_zsz[i];
      if (zs != n && !(zo & MH_TASK_NODRAW)) draws += n;       // (lcp_lemke's trivial exit, LCP.cpp:578, returns before the draws of :618-620)
      total += d.t_piv[i]; zs = zo & ~MH_TASK_NODRAW; last = r;
      wf += d.t_work[MH_WORK * i]; wb += d.t_work[MH_WORK * i + 1];
      if (st == 1) { chosen = r; break; }
    }
    lst[w] = (chosen >= 0) ? 1 : 0; piv[w] = total; d.zsz[w] = zs;
    d.work[MH_WORK * (size_t)w] += wf; d.work[MH_WORK * (size_t)w + 1] += wb;
    for (int r = 0; r < R; r++) {                                  // issued flops and time: every attempt that ran, selected or not
      const size_t i = (size_t)r * B + w;
      if (d.t_st[i] == -1) continue;
      d.work[MH_WORK * (size_t)w + 2] += d.t_work[MH_WORK * i + 2]; d.work[MH_WORK * (size_t)w + 3] += d.t_work[MH_WORK * i + 3];
    }
    uint32_t* rg = d.rng + (size_t)w * MH_RAND_WORDS;            // glibc TYPE_3, as mh_rand_next
    unsigned idx = rg[31];
    for (long k = 0; k < draws; k++) { rg[idx] = rg[idx] + rg[(idx + 28) % 31]; idx = (idx + 1) % 31; }
    rg[31] = idx;
    s_sel = (chosen >= 0) ? chosen : last;
  }
  __syncthreads();
  const size_t i = (size_t)s_sel * B + w;
  for (int k = t; k < n; k += T) d.z[(size_t)w * d.nmax + k] = d.t_z[i * d.nmax + k];
}
}}

static int core_ladder_alloc(mh_imp_core* c, long ntasks)
{
  if (c->t_cap >= ntasks) return MH_OK;
  void* ps[] = { c->t_wsd, c->t_wsi, c->t_z, c->t_st, c->t_piv, c->t_zsz, c->t_rng, c->t_work };
  for (void* q : ps) if (q) (void)hipFree(q);
  c->t_cap = 0;
  const size_t n = (size_t)c->nmax, nt = (size_t)ntasks;
  const size_t nsl = (size_t)((ntasks < mh_task_slots(c->nmax)) ? ntasks : mh_task_slots(c->nmax));     // one LU workspace per PERSISTENT workgroup, not per task
  c->t_slots = (long)nsl;
  bool ok = hipMalloc((void**)&c->t_wsd, nsl * (n * n + 5 * n) * 8) == hipSuccess && hipMalloc((void**)&c->t_wsi, nsl * 4 * n * 4) == hipSuccess
         && hipMalloc((void**)&c->t_z, nt * n * 8) == hipSuccess && hipMalloc((void**)&c->t_st, nt * 4) == hipSuccess
         && hipMalloc((void**)&c->t_piv, nt * 4) == hipSuccess && hipMalloc((void**)&c->t_zsz, nt * 4) == hipSuccess
         && hipMalloc((void**)&c->t_rng, nt * MH_RAND_WORDS * 4) == hipSuccess && hipMalloc((void**)&c->t_work, nt * MH_WORK * 8) == hipSuccess;
  if (ok && !c->solved_at) ok = hipMalloc((void**)&c->solved_at, (5 * (size_t)c->B + 2) * 4) == hipSuccess;   // (+ the launch order of a full-chip round)   // (+ the gate's counter and lcp_fast's verdicts, core_solve_round)   // + next attempt, attempts over (mh_lcp_block.h pick_task)
  if (ok) ok = hipMemset(c->t_rng, 0, nt * MH_RAND_WORDS * 4) == hipSuccess;
  if (!ok) {                                                      // no room: the caller runs the ladder in sequence
    void* qs[] = { c->t_wsd, c->t_wsi, c->t_z, c->t_st, c->t_piv, c->t_zsz, c->t_rng, c->t_work };
    for (void* q : qs) if (q) (void)hipFree(q);
    c->t_wsd = nullptr; c->t_wsi = nullptr; c->t_z = nullptr; c->t_st = nullptr; c->t_piv = nullptr; c->t_zsz = nullptr; c->t_rng = nullptr; c->t_work = nullptr;
    (void)hipGetLastError();
    return MH_ERR_HIP;
  }
  c->t_cap = ntasks;
  return MH_OK;
}

// lcp_lemke_regularized(_MM, _qq, z, o) as tasks, in two halves so that the first can run on another stream while lcp_fast is still
// at work: ladder_launch starts the B x R attempts for the worlds `mask` selects (it reads _MM, _qq, the sizes -- nothing lcp_fast
// writes); ladder_finish, on the pipeline's stream and after lcp_fast, runs the wave solver's whole ladder for the problems of at most
// 64 rows and the selection for the worlds need[b] selects (a subset of `mask`): lst / piv receive result flags and pivot counts.
struct LadderPlan { int R; long ntasks; bool ok; mh_lcp_opts o; bool has_o; };
static LadderPlan core_ladder_plan(mh_imp_core* c, const mh_lcp_opts* o)
{
  LadderPlan L; L.has_o = o != nullptr; if (o) L.o = *o;
  const int n = c->nmax;
  const int min_exp = o ? o->min_exp : -20, max_exp = o ? o->max_exp : 1; const unsigned step = o ? o->step_exp : 1u;
  L.R = 1; if (step > 0) for (int rf = min_exp; rf < max_exp; rf += (int)step) L.R++;
  L.ntasks = (long)c->B * L.R;
  const double nsl = (double)((L.ntasks < mh_task_slots(n)) ? L.ntasks : mh_task_slots(n));
  const double bytes = nsl * (((double)n * n + 5.0 * n) * 8.0 + 16.0 * n) + (double)L.ntasks * (8.0 * n + 200.0);
  // the tasks' LU workspaces (n^2 + 5 n doubles each) number the persistent workgroups of a task launch since round 5 (mh_task_slots: 2 GB at most for 16-box stacks,
  // where round 4's one per (world, attempt) took 48 GB at 1024 worlds and made larger batches run their ladders in sequence); per task only z, sizes, rand() scratch and
  // counters are kept.  Everything is kept until the batch is destroyed and may take at most 70 % of what the device has free when first allocated
  bool fits = c->t_cap >= L.ntasks;
  if (!fits) { size_t fr = 0, tot = 0; fits = hipMemGetInfo(&fr, &tot) == hipSuccess && bytes < 0.7 * (double)fr; (void)hipGetLastError(); }
  L.ok = mh_g_debug_tasks != 0 && n > MH_LCP_MAX_N_WAVE && step > 0 && fits && core_ladder_alloc(c, L.ntasks) == MH_OK;
  return L;
}
// sched: the tasks are handed out by need (mh_lcp_block.h pick_task: as many workgroups as the chip holds, each taking tasks until none is
// left) instead of by block index.  Not beside lcp_fast: workgroups that stay would keep its kernel off the CUs they occupy.
// by_verdict: beside lcp_fast's kernel, only the worlds it has failed on (mh_lcp_block.h pick_task); resume: the second launch of that scheme -- the
// hand-out continues where the first left it (nothing is reset)
static int core_ladder_launch(mh_imp_core* c, hipStream_t st, const LadderPlan& L, const int* mask, bool sched, bool by_verdict = false, bool resume = false, bool ordered = false)
{
  const int B = c->B, n = c->nmax;
  const mh_lcp_opts* o = L.has_o ? &L.o : nullptr;
  if (!resume) {
    MH_HIP(hipMemsetAsync(c->solved_at, 0x7f, (size_t)B * 4, st));
    MH_HIP(hipMemsetAsync(c->solved_at + B, 0, 2 * (size_t)B * 4, st));
    if (sched) MH_HIP(hipMemsetAsync(c->t_st, 0xff, (size_t)L.ntasks * 4, st));                     // -1: never handed out
    MH_HIP(hipMemsetAsync(c->t_work, 0, (size_t)L.ntasks * MH_WORK * 8, st));
  }
  mh::LcpParams P; P.kind = MH_LCP_LEMKE_REG; P.min_exp = o ? o->min_exp : -20; P.step_exp = o ? o->step_exp : 1u; P.max_exp = o ? o->max_exp : 1;
  P.piv_tol = o ? o->piv_tol : -1.0; P.zero_tol = o ? o->zero_tol : -1.0;
  static const mh::Pow10Table p10 = [] { mh::Pow10Table t; for (int i = 0; i < 64; i++) t.v[i] = std::pow(10.0, (double)(i - 32)); return t; }();
  // the ladder offers B x (8-12) useful workgroups at once: the narrow geometry (three problems per CU) unless its compact path does not take n
  const bool wide = mh_g_debug_blk ? mh_g_debug_blk == 2 && n >= 192 : n > 512;
  const bool one_wave = n <= 512 && (mh_g_debug_blk == 3 || (mh_g_debug_blk == 0 && !(by_verdict || resume) && L.ntasks >= (long)MH_BLK1_MIN_PER_CU * mh_cu_count()));
  // (B worlds: the ladder then runs in sequence per world, paced by throughput.  Beside lcp_fast's kernel -- by_verdict / resume -- the step is paced by the
  //  longest attempt, and that runs faster with two 256-thread problems on a CU than with four of 128: warm steps 7.4 -> 6.7 s, cold the same,
  //  profiles/r04_d_config4_steps.txt)
  const bool two_waves = n <= 512 && (mh_g_debug_blk == 4 || (mh_g_debug_blk == 0 && !(by_verdict || resume) && B >= MH_BLK2_MIN_PER_CU * mh_cu_count()));
  const bool wide2 = n >= MH_BLKX_MIN_N && n <= MH_BLKX_MAX_N && (mh_g_debug_blk == 0 || mh_g_debug_blk == 2);     // (two rows per lane: the structure-exploiting LU up to 2048 rows)
  // 512 < n <= 1024 with more tasks than the chip has CUs: two 256-thread problems per CU, four rows per lane (mh_lcp_blky.hip); mh_debug_set(2, 2) keeps the wide one
  const bool narrow4 = n > 512 && n <= 1024 && (mh_g_debug_blk == 5 || (mh_g_debug_blk == 0 && L.ntasks >= (long)MH_BLKY_MIN_TASKS_PER_CU * mh_cu_count()));
  const hipError_t le = (narrow4 ? mh_launch_lcp_blky : wide2 ? mh_launch_lcp_blkx : two_waves ? mh_launch_lcp_blk2 : one_wave ? mh_launch_lcp_blk1 : (wide ? mh_launch_lcp_blkw : mh_launch_lcp_blk))(st, MH_LCP_LEMKE_REG, (int)L.ntasks, n, c->MM, n, (long)n * n, c->qq, c->t_z, nullptr, c->t_zsz,
      c->t_rng, c->t_st, c->t_piv, nullptr, 0, nullptr, &P, &p10, c->t_wsd, c->t_wsi, mask, c->ncur, mh_g_debug_compact | (mh_g_debug_reuse << 2) | (sched ? 8 : 64) | ((sched && by_verdict) ? 32 : 0) | ((sched && ordered) ? 128 : 0), c->t_work, B, c->solved_at);
  MH_HIP(le);
  return MH_OK;
}
static int core_ladder_finish(mh_imp_core* c, hipStream_t s, const LadderPlan& L, const int* need, int* lst, unsigned* piv)
{
  namespace im = mh::imp;
  const int B = c->B, n = c->nmax;
  int rc = mh_lcp_solve_dev_masked(s, MH_LCP_LEMKE_REG, B, n, c->MM, n, (long)n * n, c->qq, c->z, c->zsz, c->zsz, c->rng, lst, piv,
                                   nullptr, 0, nullptr, L.has_o ? &L.o : nullptr, need, c->ws_d, c->ws_i, c->ncur, c->work, 1);
  if (rc != MH_OK) return rc;
  hipLaunchKernelGGL(im::k_ladder_select, dim3(B), dim3(im::T), 0, s, *c, need, L.R, lst, piv);
  MH_HIP(hipGetLastError());
  static const bool stats = getenv("MH_LADDER_STATS") != nullptr;       // diagnostic: how much of the tasks' work the selection used
  if (stats) {
    MH_HIP(hipStreamSynchronize(s));
    std::vector<int> st((size_t)L.ntasks), nd((size_t)B), sa((size_t)B); std::vector<unsigned> pv((size_t)L.ntasks);
    MH_HIP(hipMemcpy(st.data(), c->t_st, st.size() * 4, hipMemcpyDeviceToHost)); MH_HIP(hipMemcpy(pv.data(), c->t_piv, pv.size() * 4, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(nd.data(), need, nd.size() * 4, hipMemcpyDeviceToHost)); MH_HIP(hipMemcpy(sa.data(), c->solved_at, sa.size() * 4, hipMemcpyDeviceToHost));
    double used = 0, all = 0, longest = 0; long run = 0, worlds = 0;
    for (int w = 0; w < B; w++) {
      if (!nd[w]) continue;
      worlds++;
      double mine = 0;
      for (int r = 0; r < L.R; r++) { const size_t i = (size_t)r * B + w; if (st[i] < 0) continue; run++; all += pv[i]; if (r <= sa[w]) { used += pv[i]; mine += pv[i]; } }
      if (mine > longest) longest = mine;
    }
    fprintf(stderr, "ladder tasks: %ld worlds, %ld tasks run, pivots used %.0f of %.0f run (%.1f %%), longest world %.0f, mean world %.0f\n", worlds, run, used, all, all > 0 ? 100.0 * used / all : 0.0, longest, worlds ? used / worlds : 0.0);
  }
  return MH_OK;
}
static int core_lemke_stage(mh_imp_core* c, hipStream_t s, const mh_lcp_opts* o, const int* need, int* lst, unsigned* piv)
{
  const int B = c->B, n = c->nmax;
  const LadderPlan L = core_ladder_plan(c, o);
  if (!L.ok)
    return mh_lcp_solve_dev_masked(s, MH_LCP_LEMKE_REG, B, n, c->MM, n, (long)n * n, c->qq, c->z, c->zsz, c->zsz, c->rng, lst, piv,
                                   nullptr, 0, nullptr, o, need, c->ws_d, c->ws_i, c->ncur, c->work);
  const int rc = core_ladder_launch(c, s, L, need, mh_g_debug_sched != 0);
  return rc != MH_OK ? rc : core_ladder_finish(c, s, L, need, lst, piv);
}

// The gate between lcp_fast's kernel and the ladder's tasks on the second stream: one thread that waits until every workgroup of lcp_fast has
// STARTED (they count themselves in, mh_lcp_block.h).  Launched earlier, the tasks' workgroups -- which stay until no task is left -- hold CUs that
// 1024-thread workgroups still waiting for their turn need whole; launched behind the gate they only take what finished worlds leave.
// (Bounded: it gives up after 20 s of the constant-rate clock -- the tasks then merely start early.)
// Longest processing time first: the worlds ranked by the solver time they have used since the batch was created (work[.][3], ticks: lcp_fast's kernel and the
// ladder's tasks both add to it), the most expensive first; ties by index.  order[] is a permutation of 0 .. B-1.  On a full chip lcp_fast's kernel lasts as long as
// its slowest world PLUS the time that world waited for a CU (a second, with four dispatch rounds of 256), and the ladder's hand-out breaks its ties this way.
__global__ void k_rank(const double* __restrict__ work, int* __restrict__ order, int B)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double key = work[MH_WORK * (size_t)b + 3];
  int rank = 0;
  for (int j = 0; j < B; j++) { const double kj = work[MH_WORK * (size_t)j + 3]; rank += (kj > key || (kj == key && j < b)) ? 1 : 0; }
  order[rank] = b;
}

__global__ void k_gate(const int* started, int target)
{
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 2000000000ull) break;
  }
}

// the solver chain of one round over the worlds with run_if set
static int core_solve_round(mh_imp_core* c, hipStream_t s, const int* run_if, int mode)
{
  namespace im = mh::imp;
  const int B = c->B, n = c->nmax;
  int rc;
  if (mode == MH_CORE_IMPACT && c->ap) {
    // every island the mask selects takes the Anitescu-Potra model: lcp_lemke_regularized(_MM, _qq, z, -20, 1, -2) alone (ICH-AP:333)
    mh_lcp_opts oa; oa.min_exp = -20; oa.step_exp = 1u; oa.max_exp = -2; oa.piv_tol = -1.0; oa.zero_tol = -1.0;
    return core_lemke_stage(c, s, &oa, run_if, c->lst1, c->piv1);
  }
  // Speculation: on LCPs of 256 rows and more lcp_fast[_regularized] almost never succeeds (box stacks: 19 % at n = 256, 0 % at n = 512), so
  // the Lemke ladder that will be needed anyway starts at once, on a second stream, for every world of the round; its tasks read nothing
  // lcp_fast writes, and the selection afterwards only looks at the worlds whose lcp_fast did fail.  mh_debug_set(4, 1) turns it off.
  LadderPlan L; L.ok = false;
  // ... unless the batch fills the chip with LCPs on which lcp_fast practically never succeeds (16-box stacks, n = 512, x 1024): running ahead
  // buys nothing there, and the tasks are better handed out by need afterwards (31.7 s instead of 35.7 s per full step)
  // (Smaller batches keep the speculation: with the full-chip launch order 16-box stacks x 64 / 256 / 512 worlds took 2.78 / 2.90 / 4.03 s per cold
  //  impact-handler call against 2.48 / 3.37 / 4.70 s, but 2.5 / 13 / 9.1 s per warm call against 3.3 / 4.7 / 5.3 s: when every world's lcp_fast starts at
  //  once there are no verdicts yet for the first launch's workgroups, they leave, and the ladder runs after lcp_fast's slowest world.)
  const bool full_chip = mh_g_debug_sched != 0 && n >= 384 && B >= 3 * mh_cu_count();
  // Full chip, mh_debug_set(4, 3) (default): lcp_fast's kernel is launched FIRST, in its own (1024-thread) geometry, and the ladder's tasks -- handed
  // out by need -- behind it on the second stream, behind a GATE (k_gate) that opens when the last of lcp_fast's workgroups has started.  That
  // kernel's time is its slowest world's (16-box stacks x 1024: the workgroups sum to 0.9 s of the chip, the slowest runs 3-4 s): the CUs the
  // finished worlds leave take the ladder's workgroups instead of idling until the last world is through.  Measured (profiles/r04_d_config4_*):
  // cold step 25.05 -> 22.04 s; warm steps 9.8 -> 9.6 s only, the two kernels slow each other down (lcp_fast's tail 4.2 -> 5.5 s, the ladder
  // 3.4 -> 6.7 s) by about what the overlap saves.  Without the gate (round 4's first attempt) the tasks' workgroups, which stay until no task is
  // left, held CUs that waiting 1024-thread workgroups need whole: 25.8 / 12.0 s.  Above n = 512 it has not been measured: 4 forces it there.
  const bool overlap = full_chip && n >= 256 && (mh_g_debug_tasks >= 4 || (mh_g_debug_tasks == 3 && n <= 512));
  const bool spec_wanted = (mh_g_debug_tasks >= 2 && n >= 256 && !full_chip) || overlap;
  if (spec_wanted) L = core_ladder_plan(c, nullptr);
  bool spec = spec_wanted && L.ok;
  if (spec && !c->s2) {
    hipStream_t s2; hipEvent_t e0, e1;
    if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess
        && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) { c->s2 = s2; c->ev0 = e0; c->ev1 = e1; }
    else { (void)hipGetLastError(); spec = false; }
  }
  // (lcp_fast then runs in the narrow geometry: a 1024-thread workgroup needs a whole CU and would wait for the tasks' workgroups to leave
  //  it -- 16 boxes x 256 worlds: 4.05 -> 3.55 s per cold call)
  // The gate in BOTH speculative forms since round 5: a task launch's workgroups are persistent now (one LU workspace each, mh_lcp_block.h), so launched ahead of
  // lcp_fast's kernel they would hold every CU until the last task is done and lcp_fast would run AFTER the ladder instead of beside it.  lcp_fast's kernel goes first,
  // the gate opens when its last workgroup has started, and the ladder's workgroups take what is left of the chip -- and the rest as lcp_fast's worlds finish.
  int* const gate = spec ? c->solved_at + 3 * (size_t)B : nullptr;
  const bool ordered = spec && overlap && mh_g_debug_lpt != 0;
  if (spec) {
    MH_HIP(hipMemsetAsync(gate, 0, ((size_t)B + 2) * 4, s));         // workgroups started, verdicts published, then one verdict per world (0: lcp_fast has not spoken)
    if (ordered) { hipLaunchKernelGGL(k_rank, dim3((B + 255) / 256), dim3(256), 0, s, c->work, gate + B + 2, B); MH_HIP(hipGetLastError()); }
    MH_HIP(hipEventRecord((hipEvent_t)c->ev0, s));
    MH_HIP(hipStreamWaitEvent((hipStream_t)c->s2, (hipEvent_t)c->ev0, 0));
  }
  const int fast_geom = (spec && !overlap) ? 2 : 0;
  if (mode == MH_CORE_IMPACT) {
    mh_lcp_opts o1; o1.min_exp = -20; o1.step_exp = 4u; o1.max_exp = -8; o1.piv_tol = -1.0; o1.zero_tol = -1.0;   // ICH-QP:219
    rc = mh_lcp_solve_dev_masked(s, MH_LCP_FAST_REG, B, n, c->MM, n, (long)n * n, c->qq, c->z, c->zsz, nullptr, c->rng, c->lst1, c->piv1,
                                 nullptr, 0, nullptr, &o1, run_if, c->ws_d, c->ws_i, c->ncur, c->work, fast_geom, gate, ordered ? 1 : 0);
  } else {
    rc = mh_lcp_solve_dev_masked(s, MH_LCP_FAST, B, n, c->MM, n, (long)n * n, c->qq, c->z, c->zsz, c->zsz, c->rng, c->lst1, c->piv1,
                                 nullptr, 0, nullptr, nullptr, run_if, c->ws_d, c->ws_i, c->ncur, c->work, fast_geom, gate, ordered ? 1 : 0);   // CStab:954
  }
  if (rc == MH_OK && spec) {                                     // the ladder's tasks behind lcp_fast's launch and the gate: by need and verdict on a full chip, every masked world's otherwise
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, (hipStream_t)c->s2, gate, B);
    { const hipError_t ge = hipGetLastError(); if (ge != hipSuccess) { (void)hipStreamSynchronize((hipStream_t)c->s2); return fail(MH_ERR_HIP, "k_gate failed: %s", hipGetErrorString(ge)); } }
    rc = overlap ? core_ladder_launch(c, (hipStream_t)c->s2, L, run_if, true, true, false, ordered) : core_ladder_launch(c, (hipStream_t)c->s2, L, run_if, false);
    if (rc == MH_OK) { const hipError_t e = hipEventRecord((hipEvent_t)c->ev1, (hipStream_t)c->s2); if (e != hipSuccess) rc = fail(MH_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e)); }
    if (rc != MH_OK) { (void)hipStreamSynchronize((hipStream_t)c->s2); return rc; }
  } else
  if (rc != MH_OK) { if (spec) (void)hipStreamSynchronize((hipStream_t)c->s2); return rc; }     // (the tasks' stream joins this one on every way out)
  hipLaunchKernelGGL(im::k_lemke_prep, dim3(B), dim3(im::T), 0, s, *c, run_if, mode);
  { const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (spec) (void)hipStreamWaitEvent(s, (hipEvent_t)c->ev1, 0); return fail(MH_ERR_HIP, "k_lemke_prep failed: %s", hipGetErrorString(e)); } }
  if (spec) {                                                     // the tasks have been running beside lcp_fast: wait for them, then select
    MH_HIP(hipStreamWaitEvent(s, (hipEvent_t)c->ev1, 0));
    if (overlap) {                                                // what lcp_fast decided after the first launch's workgroups had left
      rc = core_ladder_launch(c, s, L, c->need2, true, false, true, ordered);
      if (rc != MH_OK) return rc;
    }
    return core_ladder_finish(c, s, L, c->need2, c->lst2, c->piv2);
  }
  return core_lemke_stage(c, s, nullptr, c->need2, c->lst2, c->piv2);                                                // ICH-QP:224, CStab:955
}

int mh_imp_core_process(mh_imp_core* c, void* stream, int mode)
{
  namespace im = mh::imp;
  hipStream_t s = (hipStream_t)stream;
  const int B = c->B, ncm = c->ncmax;
  MH_HIP(hipMemsetAsync(c->maxisl, 0, sizeof(int), s));
  hipLaunchKernelGGL(im::k_prep, dim3(B), dim3(im::T), 0, s, *c, mode);
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(c->hmax, c->maxisl, sizeof(int), hipMemcpyDeviceToHost, s));
  MH_HIP(hipStreamSynchronize(s));
  const int rounds = *c->hmax;
  // MH_IMP_TRACE=1: synchronise after every stage and name it on stderr (locates a faulting kernel)
  static const bool trace_stages = std::getenv("MH_IMP_TRACE") != nullptr;
  auto stage = [&](const char* what, int r) { if (trace_stages) { hipError_t e = hipStreamSynchronize(s); std::fprintf(stderr, "[mh_imp] round %d %s: %s\n", r, what, hipGetErrorString(e)); std::fflush(stderr); } };
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL(im::k_gram, dim3((ncm * ncm + im::T - 1) / im::T, B), dim3(im::T), 0, s, *c, r, mode);
    if (mode == MH_CORE_STAB && c->bT1) hipLaunchKernelGGL(im::k_bilat_X, dim3(B), dim3(im::T), 0, s, *c, r);
    hipLaunchKernelGGL(im::k_mm, dim3(c->nmax + 1, B), dim3(im::T), 0, s, *c, r, mode, 0);
    MH_HIP(hipGetLastError());
    stage("gram + mm", r);
    int rc = core_solve_round(c, s, c->run, mode);
    if (rc != MH_OK) return rc;
    stage("first solve", r);
    if (mode == MH_CORE_STAB) {
      hipLaunchKernelGGL(im::k_stab_apply, dim3(B), dim3(im::T), 0, s, *c, r);
      MH_HIP(hipGetLastError());
      continue;
    }
    hipLaunchKernelGGL(c->ap ? im::k_post_ap : im::k_post, dim3(B), dim3(im::T), 0, s, *c, r, 0);
    if (c->ap) hipLaunchKernelGGL(im::k_apply_ap, dim3(B), dim3(im::T), 0, s, *c, r);
    stage("post 0", r);
    // second solve for the worlds whose restitution impulses left a contact approaching (ICH:591-600): same _MM, new _qq.
    // The mask `again` is computed on the device (k_post, phase 0) and the round is always enqueued -- masked
    // workgroups exit at once -- so contacts written through device_ptrs() get it too (no host-side epsilon scan).
    hipLaunchKernelGGL(im::k_mm, dim3(1, B), dim3(im::T), 0, s, *c, r, mode, 1);
    MH_HIP(hipGetLastError());
    stage("mm 1", r);
    rc = core_solve_round(c, s, c->again, mode);
    if (rc != MH_OK) return rc;
    stage("second solve", r);
    hipLaunchKernelGGL(c->ap ? im::k_post_ap : im::k_post, dim3(B), dim3(im::T), 0, s, *c, r, 1);
    if (c->ap) hipLaunchKernelGGL(im::k_apply_ap, dim3(B), dim3(im::T), 0, s, *c, r);
    stage("post 1", r);
    static const mh::Pow10Table p10 = [] { mh::Pow10Table t; for (int i = 0; i < 64; i++) t.v[i] = std::pow(10.0, (double)(i - 32)); return t; }();   // LCP.cpp:285
    hipLaunchKernelGGL(im::k_noslip, dim3(B), dim3(64), 0, s, *c, r, p10);      // the islands of this round that take the no-slip model
    hipLaunchKernelGGL(im::k_unwind, dim3((B + 63) / 64), dim3(64), 0, s, *c);
    MH_HIP(hipGetLastError());
  }
  if (mode == MH_CORE_IMPACT) { hipLaunchKernelGGL(im::k_finish, dim3(B), dim3(im::T), 0, s, *c); MH_HIP(hipGetLastError()); }
  return MH_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// seam B2: include/moby_hip_impact.h
struct mh_impact_batch {
  int device;                // the HIP device the batch lives on (current at create); every entry point runs there (MH_ON_DEVICE)
  int B, nb, nc, nk, n;
  mh_imp_core c;
  double* d_mass; double* d_inertia; double* d_state; mh_contact* d_contacts; uint32_t* d_rng; int* d_status;
};

extern "C" {

int mh_impact_batch_device(const mh_impact_batch* ib) { return ib ? ib->device : fail(MH_ERR_INVALID_ARG, "null batch"); }

int mh_impact_batch_destroy(mh_impact_batch* ib)
{
  if (!ib) return MH_OK;
  MH_ON_DEVICE(ib);
  (void)hipDeviceSynchronize();
  mh_imp_core_destroy(&ib->c);
  void* ps[] = { ib->d_mass, ib->d_inertia, ib->d_state, ib->d_contacts, ib->d_rng, ib->d_status };
  for (void* p : ps) if (p) (void)hipFree(p);
  delete ib;
  return MH_OK;
}

int mh_impact_batch_create(int B, int nb, int nc, int nk, const double* mass, const double* inertia, mh_impact_batch** out)
{
  if (!out) return fail(MH_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (B <= 0 || nb <= 0 || nc <= 0) return fail(MH_ERR_INVALID_ARG, "B, nb, nc must be positive");
  if (!mass || !inertia) return fail(MH_ERR_INVALID_ARG, "null mass/inertia");
  if (nk < 4 || (nk & 1)) return fail(MH_ERR_INVALID_ARG, "nk must be even and >= 4 (ContactParameters.cpp:128-135), got %d", nk);
  if (nb > mh::imp::MAXB) return fail(MH_ERR_INVALID_ARG, "nb = %d > %d bodies per world", nb, mh::imp::MAXB);
  if (nc > mh::imp::MAXC) return fail(MH_ERR_INVALID_ARG, "nc = %d > %d contacts per world", nc, mh::imp::MAXC);
  const long n = 6L * nc + (long)nc * (nk / 2);
  if (n > MH_LCP_MAX_N_BLOCK) return fail(MH_ERR_UNSUPPORTED_N, "impact LCP n = %ld > %d", n, MH_LCP_MAX_N_BLOCK);
  for (int i = 0; i < nb; i++) {
    if (!(mass[i] > 0.0) || !(inertia[3*i] > 0.0) || !(inertia[3*i+1] > 0.0) || !(inertia[3*i+2] > 0.0))
      return fail(MH_ERR_INVALID_ARG, "body %d: mass and principal inertias must be positive", i);
  }
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  mh_impact_batch* ib = new mh_impact_batch();
  std::memset(ib, 0, sizeof(*ib));
  if (hipGetDevice(&ib->device) != hipSuccess) { delete ib; return fail(MH_ERR_HIP, "hipGetDevice failed"); }     // (after the memset: the batch belongs to the device current at create)
  ib->B = B; ib->nb = nb; ib->nc = nc; ib->nk = nk; ib->n = (int)n;
  int rc = mh_imp_core_create(&ib->c, B, nb, nc, nk, (int)n);
  if (rc != MH_OK) { delete ib; return rc; }
  const size_t sB = (size_t)B;
  bool ok = hipMalloc((void**)&ib->d_mass, nb * 8) == hipSuccess && hipMalloc((void**)&ib->d_inertia, nb * 24) == hipSuccess
         && hipMalloc((void**)&ib->d_state, sB * nb * 13 * 8) == hipSuccess && hipMalloc((void**)&ib->d_contacts, sB * nc * sizeof(mh_contact)) == hipSuccess
         && hipMalloc((void**)&ib->d_rng, sB * MH_RAND_WORDS * 4) == hipSuccess && hipMalloc((void**)&ib->d_status, sB * 4) == hipSuccess;
  if (ok) {
    std::vector<uint32_t> hr((size_t)B * MH_RAND_WORDS);
    mh_rand_seed(hr.data(), 1u);
    for (int b = 1; b < B; b++) std::memcpy(&hr[(size_t)b * MH_RAND_WORDS], hr.data(), MH_RAND_WORDS * 4);
    ok = hipMemcpy(ib->d_mass, mass, nb * 8, hipMemcpyHostToDevice) == hipSuccess
      && hipMemcpy(ib->d_inertia, inertia, nb * 24, hipMemcpyHostToDevice) == hipSuccess
      && hipMemcpy(ib->d_rng, hr.data(), hr.size() * 4, hipMemcpyHostToDevice) == hipSuccess
      && hipMemset(ib->d_state, 0, sB * nb * 13 * 8) == hipSuccess && hipMemset(ib->d_contacts, 0, sB * nc * sizeof(mh_contact)) == hipSuccess
      && hipMemset(ib->d_status, 0, sB * 4) == hipSuccess;
  }
  if (!ok) { mh_impact_batch_destroy(ib); return fail(MH_ERR_HIP, "device allocation / upload of body tables failed"); }
  ib->c.mass = ib->d_mass; ib->c.inertia = ib->d_inertia; ib->c.state = ib->d_state; ib->c.contacts = ib->d_contacts;
  ib->c.ncount = nullptr; ib->c.cdist = nullptr; ib->c.rng = ib->d_rng; ib->c.status = ib->d_status;
  *out = ib;
  return MH_OK;
}

int mh_impact_batch_lcp_size(const mh_impact_batch* ib) { return ib ? ib->n : 0; }

int mh_impact_batch_set_model(mh_impact_batch* ib, int model)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  if (model != MH_IMPACT_MODEL_DS && model != MH_IMPACT_MODEL_AP) return fail(MH_ERR_INVALID_ARG, "impact model %d (MH_IMPACT_MODEL_DS / _AP)", model);
  ib->c.ap = (model == MH_IMPACT_MODEL_AP) ? 1 : 0;
  return MH_OK;
}

int mh_impact_batch_upload(mh_impact_batch* ib, const double* state, const mh_contact* contacts)
{
  if (!ib || !state || !contacts) return fail(MH_ERR_INVALID_ARG, "null batch/state/contacts");
  MH_ON_DEVICE(ib);
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
  MH_HIP(hipMemcpy(ib->d_state, state, (size_t)ib->B * ib->nb * 13 * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d_contacts, contacts, ncon * sizeof(mh_contact), hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_impact_batch_process(mh_impact_batch* ib, void* stream)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  MH_HIP(hipMemsetAsync(ib->c.cnt, 0, (size_t)ib->B * 5 * 8, (hipStream_t)stream));     // pivots / solves are per call
  return mh_imp_core_process(&ib->c, stream, MH_CORE_IMPACT);
}

int mh_impact_batch_download(mh_impact_batch* ib, double* state, double* impulses, int* status, unsigned* pivots, int* solves)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  if (state) MH_HIP(hipMemcpy(state, ib->d_state, B * ib->nb * 13 * 8, hipMemcpyDeviceToHost));
  if (impulses) MH_HIP(hipMemcpy(impulses, ib->c.imp, B * ib->nc * 3 * 8, hipMemcpyDeviceToHost));
  if (status) MH_HIP(hipMemcpy(status, ib->d_status, B * 4, hipMemcpyDeviceToHost));
  if (pivots || solves) {
    std::vector<unsigned long long> cnt(B * 5);
    MH_HIP(hipMemcpy(cnt.data(), ib->c.cnt, B * 5 * 8, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < B; b++) { if (solves) solves[b] = (int)cnt[5 * b]; if (pivots) pivots[b] = (unsigned)cnt[5 * b + 2]; }
  }
  return MH_OK;
}

int mh_imp_core_lu_work(mh_imp_core* c, double* work, int reset)
{
  MH_HIP(hipDeviceSynchronize());
  if (work) {
    MH_HIP(hipMemcpy(work, c->work, (size_t)c->B * MH_WORK * 8, hipMemcpyDeviceToHost));
    int dev = 0, khz = 0;                                          // [3]: ticks of the constant-rate wall clock -> seconds
    MH_HIP(hipGetDevice(&dev)); MH_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    for (int b = 0; b < c->B; b++) work[MH_WORK * (size_t)b + 3] = (khz > 0) ? work[MH_WORK * (size_t)b + 3] / (1e3 * (double)khz) : 0.0;
  }
  if (reset) MH_HIP(hipMemset(c->work, 0, (size_t)c->B * MH_WORK * 8));
  return MH_OK;
}

int mh_impact_batch_lu_work(mh_impact_batch* ib, double* work, int reset)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  return mh_imp_core_lu_work(&ib->c, work, reset);
}

int mh_impact_batch_debug_lcp(mh_impact_batch* ib, double* MM, double* qq)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B, n = (size_t)ib->n;
  if (MM) MH_HIP(hipMemcpy(MM, ib->c.MM, B * n * n * 8, hipMemcpyDeviceToHost));
  if (qq) MH_HIP(hipMemcpy(qq, ib->c.qq, B * n * 8, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_save_solver_state(mh_impact_batch* ib, double* zlast, int* zlast_size, uint32_t* rng, int* status)
{
  if (!ib || !zlast || !zlast_size || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ib);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  MH_HIP(hipMemcpy(zlast, ib->c.zlast, B * ib->n * 8, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(zlast_size, ib->c.zlast_size, B * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(rng, ib->d_rng, B * MH_RAND_WORDS * 4, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(status, ib->d_status, B * 4, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_load_solver_state(mh_impact_batch* ib, const double* zlast, const int* zlast_size, const uint32_t* rng,
                                      const int* status)
{
  if (!ib || !zlast || !zlast_size || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ib);
  for (int b = 0; b < ib->B; b++)
    if (zlast_size[b] < 0 || zlast_size[b] > ib->n) return fail(MH_ERR_INVALID_ARG, "world %d: _zlast of size %d in a batch of capacity n = %d", b, zlast_size[b], ib->n);
  MH_HIP(hipDeviceSynchronize());
  const size_t B = (size_t)ib->B;
  MH_HIP(hipMemcpy(ib->c.zlast, zlast, B * ib->n * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->c.zlast_size, zlast_size, B * 4, hipMemcpyHostToDevice));
  // _z's storage after a solve holds the same vector (its first 5 nc entries re-packed in place): a resumed handler
  // starts from _zlast wherever sizes match, and from this copy where they do not
  MH_HIP(hipMemcpy(ib->c.zbuf, zlast, B * ib->n * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->c.zbuf_cap, zlast_size, B * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d_rng, rng, B * MH_RAND_WORDS * 4, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->d_status, status, B * 4, hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_impact_batch_save_noslip_state(mh_impact_batch* ib, double* v, int* v_size)
{
  if (!ib || !v || !v_size) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ib);
  MH_HIP(hipDeviceSynchronize());
  MH_HIP(hipMemcpy(v, ib->c.vns, (size_t)ib->B * MH_NOSLIP_MAX * 8, hipMemcpyDeviceToHost));
  MH_HIP(hipMemcpy(v_size, ib->c.vns_size, (size_t)ib->B * 4, hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_impact_batch_load_noslip_state(mh_impact_batch* ib, const double* v, const int* v_size)
{
  if (!ib || !v || !v_size) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ib);
  for (int b = 0; b < ib->B; b++)
    if (v_size[b] < 0 || v_size[b] > MH_NOSLIP_MAX) return fail(MH_ERR_INVALID_ARG, "world %d: _v of size %d (0 .. %d)", b, v_size[b], MH_NOSLIP_MAX);
  MH_HIP(hipDeviceSynchronize());
  MH_HIP(hipMemcpy(ib->c.vns, v, (size_t)ib->B * MH_NOSLIP_MAX * 8, hipMemcpyHostToDevice));
  MH_HIP(hipMemcpy(ib->c.vns_size, v_size, (size_t)ib->B * 4, hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_impact_batch_device_ptrs(mh_impact_batch* ib, double** state_dev, mh_contact** contacts_dev)
{
  if (!ib) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ib);
  if (state_dev) *state_dev = ib->d_state;
  if (contacts_dev) *contacts_dev = ib->d_contacts;
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
