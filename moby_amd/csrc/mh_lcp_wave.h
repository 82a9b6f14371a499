// Dense LCP solvers, one wavefront per problem, n <= 64 (lane i <-> variable i).
//
// Replaces (same algorithms, same floating-point operation order as the CPU
// oracle in oracle/lcp.hpp, which restates the reference line by line):
//   LCP::lcp_fast               /root/reference/src/LCP.cpp:41-196
//   LCP::rand_min               src/LCP.cpp:199-209
//   LCP::lcp_fast_regularized   src/LCP.cpp:212-350
//   LCP::lcp_lemke_regularized  src/LCP.cpp:353-487
//   LCP::lcp_lemke (dense)      src/LCP.cpp:545-1003
//   LinAlgd::solve_fast (dgesv) call sites src/LCP.cpp:120,838
//
// MI355X mapping (not a translation of the reference's data structures):
//   * the sorted index vectors _bas/_nonbas become ONE 64-bit ballot mask in
//     SGPRs; "position in the list" is a popcount prefix, "r-th element" a
//     scalar bit scan -- no list maintenance, no insertion sort;
//   * M is anything that can produce M(lane, c) for a wave-uniform column c (an
//     LDS array for the C-ABI entry, an implicit accessor over G = C X C' in the
//     world kernels); the LU scratch lives in LDS (k <= 4: in registers, held
//     redundantly by every lane); the right-hand side, q, z, w and Lemke's x/d
//     vectors live one element per lane in VGPRs; pivot rows/values are broadcast
//     with v_readlane, reductions (argmin with first-index ties, idamax) are DPP
//     row operations + one ballot;
//   * the per-world libc rand() ring is spread over lanes 0..30.
// Build with -ffp-contract=off: parity with the oracle is bit-exact.
#pragma once
#include "mh_wave.h"
#include "../../include/moby_hip.h"

namespace mh {

// optional in-solver cycle accumulators (diagnostic launches of the world kernel only)
enum { LP_SETUP = 0, LP_GATHER, LP_LU, LP_GEMV, LP_RANDMIN, LP_VERIFY, LP_LEMKE, LP_COUNT };
__shared__ unsigned long long g_lcp_prof[LP_COUNT];
__shared__ int g_lcp_prof_on;
// Compiled in only in the PROFILE translation units (-DMH_PROFILE_BUILD: mh_world_*_prof.hip, what mh_world_batch_profile launches).  Until round 5 the
// switch was the run-time flag alone and every stamp point of the production kernels paid a ds_read + s_waitcnt lgkmcnt(0) + branch for it -- a full LDS
// round trip on the dependent chain of a latency-bound kernel, about a hundred times per world-step.
#ifdef MH_PROFILE_BUILD
MH_DEV unsigned long long lp_tick() { return g_lcp_prof_on ? __builtin_amdgcn_s_memtime() : 0ull; }
MH_DEV void lp_tock(int ph, unsigned long long t0) { if (g_lcp_prof_on) { const unsigned long long d = __builtin_amdgcn_s_memtime() - t0; if (lane_id() == 0) g_lcp_prof[ph] += d; } }
#else
MH_DEV unsigned long long lp_tick() { return 0ull; }
MH_DEV void lp_tock(int, unsigned long long) {}
#endif

struct Trace {
  int32_t* buf; int cap; int len;
  MH_DEV void push(int32_t v) { if (buf && len < cap && lane_id() == 0) buf[len] = v; len++; }
};

#define MH_DBL_EPS 2.220446049250313e-16
#define MH_NEAR_ZERO 1.4901161193847656e-08 /* sqrt(eps), Constants.h:21 */
#define MH_SFMIN 2.2250738585072014e-308

// Matrix accessors.  The solvers only ever read M(lane, c) for a wave-uniform
// column c (plus the lane's own diagonal entry), so a matrix is anything that
// provides
//   at_row(c)      M(lane, c), c uniform
//   diag()         M(lane, lane)
//   offdiag_max()  max |M(r,c)|, r != c  (uniform; regularisation ladder only)
// DenseLds is an explicit column-major n x n array in LDS (the C-ABI LCP entry);
// the world kernels plug in implicit matrices that are never materialised.
struct DenseLds {
  const double* M; int n;
  MH_DEV double at_row(int c) const { const int l = lane_id(); return (l < n) ? M[l + n * c] : 0.0; }
  MH_DEV double diag() const { const int l = lane_id(); return (l < n) ? M[l + n * l] : 0.0; }
  MH_DEV double offdiag_max() const {
    const int lane = lane_id();
    double m = 0.0;
    if (lane < n)
      for (int c = 0; c < n; c++) {
        const double a = fabs(M[lane + n * c]);
        if (c != lane && a > m) m = a;
      }
    return wave_max(m);
  }
};

// M(lane,c) of the caller's (possibly regularised) copy _MM = M + lam*I
template <class MatT>
MH_DEV double mat_at(const MatT& M, int c, double lam) {
  const double m = M.at_row(c);
  return (lane_id() == c) ? m + lam : m;
}

// dgesv for one rhs: A (k x k, col-major, ld=k) in LDS, lane r owns row r and
// rhs element b.  Returns LAPACK info (uniform); on info != 0 b is garbage.
// dgetf2 order: idamax -> row swap (all columns) -> scale by reciprocal ->
// rank-1 update; then dlaswp / unit-lower / upper triangular solves.
// `A` may point to LDS or to the HBM workspace (flat addressing).  Inlined at its two call
// sites (lcp_fast / lcp_lemke gathers): as an out-of-line function its callee-saved VGPR
// spills made the world kernel write ~50 GB of scratch per 200-step launch.
MH_DEV int lu_solve_wave(int k, double* A, double& b)
{
  const int lane = lane_id();
  for (int j = 0; j < k; j++) {
    const bool mine = (lane >= j) && (lane < k);
    double a = mine ? fabs(A[lane + k * j]) : -1.0;
    double amax; int jp;
    argmax_first(a, mine, amax, jp);
    if (!(amax != 0.0)) return j + 1;         // exact zero pivot column: singular
    if (jp != j) {
      if (lane < k) { double t0 = A[j + k * lane], t1 = A[jp + k * lane]; A[j + k * lane] = t1; A[jp + k * lane] = t0; }
      double bj = read_lane(b, j), bjp = read_lane(b, jp);
      if (lane == j) b = bjp; else if (lane == jp) b = bj;
      wave_sync();
    }
    if (j < k - 1) {
      const double piv = A[j + k * j];        // LDS broadcast
      if (lane > j && lane < k) {
        double l = A[lane + k * j];
        if (fabs(piv) >= MH_SFMIN) { const double r = 1.0 / piv; l = l * r; } else l = l / piv;
        A[lane + k * j] = l;
        // rank-1 update, 8 columns per trip: all LDS reads of a trip are
        // issued before its writes so their latencies overlap
        for (int c0 = j + 1; c0 < k; c0 += 8) {
          double a[8], u[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const int c = (c0 + t < k) ? c0 + t : k - 1;
            a[t] = A[lane + k * c]; u[t] = A[j + k * c];
          }
#pragma unroll
          for (int t = 0; t < 8; t++)
            if (c0 + t < k) A[lane + k * (c0 + t)] = a[t] - l * u[t];
        }
      }
      wave_sync();
    }
  }
  // unit lower
  for (int kk = 0; kk < k; kk++) {
    const double bk = read_lane(b, kk);
    if (lane > kk && lane < k) b = b - bk * A[lane + k * kk];
  }
  // upper
  for (int kk = k - 1; kk >= 0; kk--) {
    if (lane == kk) b = b / A[kk + k * kk];
    const double bk = read_lane(b, kk);
    if (lane < kk) b = b - bk * A[lane + k * kk];
  }
  return 0;
}

// The same factorisation for a matrix that lives in the HBM workspace (k above the LDS block, every Lemke
// basis of a 42-row island).  lu_solve_wave costs ~4 dependent memory phases per COLUMN there (~250 us for
// k = 42, ten times a CPU core).  Here a panel of 8 columns is loaded into registers once (lane r = logical
// row r), factorised with v_readlane broadcasts, and the trailing columns take the panel's 8 updates 8
// columns per round trip; rows are never swapped in memory -- lanes exchange their register contents and
// the index of the physical row they stand for.  Every element still receives dgetf2's updates
// a <- a - l*u in ascending column order with separately rounded products: bit-identical factors.
MH_DEV int lu_solve_wave_hbm(int k, double* A, double& b)
{
  constexpr int PB = 8;
  const int lane = lane_id();
  const bool valid = lane < k;
  int prow = lane;                                   // physical row this lane's logical row lives in
  for (int j0 = 0; j0 < k; j0 += PB) {
    const int nb = (k - j0 < PB) ? k - j0 : PB;
    double p[PB];
#pragma unroll
    for (int c = 0; c < PB; c++) p[c] = (valid && c < nb) ? A[prow + k * (j0 + c)] : 0.0;
#pragma unroll
    for (int jj = 0; jj < PB; jj++) {
      if (jj < nb) {
        const int j = j0 + jj;
        const bool mine = (lane >= j) && valid;
        double amax; int jp;
        argmax_first(mine ? fabs(p[jj]) : -1.0, mine, amax, jp);
        if (!(amax != 0.0)) return j + 1;
        if (jp != j) {                               // logical rows j <-> jp: registers and row index, no memory traffic
#pragma unroll
          for (int c = 0; c < PB; c++) { const double vj = read_lane(p[c], j), vp = read_lane(p[c], jp); if (lane == j) p[c] = vp; else if (lane == jp) p[c] = vj; }
          { const double vj = read_lane(b, j), vp = read_lane(b, jp); if (lane == j) b = vp; else if (lane == jp) b = vj; }
          { const int rj = read_lane(prow, j), rp = read_lane(prow, jp); if (lane == j) prow = rp; else if (lane == jp) prow = rj; }
        }
        if (j < k - 1) {
          const double piv = read_lane(p[jj], j);
          const bool below = (lane > j) && valid;
          if (below) { double l = p[jj]; if (fabs(piv) >= MH_SFMIN) { const double r = 1.0 / piv; l = l * r; } else l = l / piv; p[jj] = l; }
#pragma unroll
          for (int c = 0; c < PB; c++) if (c > jj && c < nb) { const double u = read_lane(p[c], j); if (below) p[c] = p[c] - p[jj] * u; }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < PB; c++) if (valid && c < nb) A[prow + k * (j0 + c)] = p[c];
    // trailing columns, 8 per round trip: element (r, c) takes the updates of the panel's steps in order
    for (int cb = j0 + nb; cb < k; cb += PB) {
      double a[PB];
#pragma unroll
      for (int u = 0; u < PB; u++) a[u] = (valid && cb + u < k) ? A[prow + k * (cb + u)] : 0.0;
#pragma unroll
      for (int u = 0; u < PB; u++) {
#pragma unroll
        for (int s2 = 0; s2 < PB; s2++) if (s2 < nb) { const double us = read_lane(a[u], j0 + s2); if (lane > j0 + s2 && valid) a[u] = a[u] - p[s2] * us; }
      }
#pragma unroll
      for (int u = 0; u < PB; u++) if (valid && cb + u < k) A[prow + k * (cb + u)] = a[u];
    }
  }
  wave_sync();
  // dgetrs: unit lower, then upper; 8 columns of factors per round trip
  for (int kb = 0; kb < k; kb += PB) {
    double l[PB];
#pragma unroll
    for (int u = 0; u < PB; u++) l[u] = (valid && kb + u < k) ? A[prow + k * (kb + u)] : 0.0;
#pragma unroll
    for (int u = 0; u < PB; u++) if (kb + u < k) { const double bk = read_lane(b, kb + u); if (lane > kb + u && valid) b = b - bk * l[u]; }
  }
  for (int ke = k; ke > 0; ke -= PB) {
    const int kb = (ke - PB > 0) ? ke - PB : 0;
    double uu[PB];
#pragma unroll
    for (int u = 0; u < PB; u++) uu[u] = (valid && kb + u < ke) ? A[prow + k * (kb + u)] : 1.0;
#pragma unroll
    for (int u = PB - 1; u >= 0; u--) if (kb + u < ke) {
      const int kk = kb + u;
      if (lane == kk) b = b / uu[u];
      const double bk = read_lane(b, kk);
      if (lane < kk) b = b - bk * uu[u];
    }
  }
  return 0;
}

// LCP.cpp:199-209: v is defined on the lanes of `mask` (list order = lane
// order).  Consumes exactly one rand().  Returns the chosen lane.
MH_DEV int rand_min_wave(double v, uint64_t mask, double tol, WaveRand& rng, double& val, bool& tie)
{
  const int lane = lane_id();
  const bool in = (mask >> lane) & 1ull;
  double vmin; int imin;
  if (popc(mask) <= 4) {
    // few candidates (the nonbasic z's): scan them in list order, first minimum wins
    uint64_t m = mask;
    imin = ctz(m); m &= m - 1;
    vmin = read_lane(v, imin);
    while (m) { const int i = ctz(m); m &= m - 1; const double x = read_lane(v, i); if (x < vmin) { vmin = x; imin = i; } }
  } else argmin_first(v, in, vmin, imin);
  const uint64_t qm = ballot(in && lane != imin && v < vmin + tol);
  const int cnt = 1 + popc(qm);
  tie = tie || cnt > 1;                          // the draw decides something
  const int rv = rng.next();                    // always consumed (LCP.cpp:208)
  const int r = (cnt == 1) ? 0 : rv % cnt;
  const int chosen = (r == 0) ? imin : nth_set_bit(qm, r - 1);
  val = read_lane(v, chosen);
  return chosen;
}

// norm_inf of M + lam*I given the off-diagonal max and this lane's diagonal
MH_DEV double norm_reg(double offmax, double dii, bool valid, double lam) {
  double d = valid ? fabs(dii + lam) : 0.0;
  double m = wave_max(d);
  return (m > offmax) ? m : offmax;
}

// LCP.cpp:41-196.  qi/zi: this lane's q[i], z[i].  zsize (uniform): z.size()
// on entry -- == n selects the warm start (LCP.cpp:65); left unchanged when the
// solver fails before writing z (LCP.cpp:125,195), set to n on success.
// LU scratch: `small` holds up to ka x ka doubles (LDS in the world kernel), `big`
// n x n (LDS in the LCP-entry kernel, an HBM workspace in the world kernel).
struct LuScratch { double* small; int ka; double* big; };

// dgesv for K = 2..4 with the whole system held redundantly in every lane's registers: no LDS
// round trips, no cross-lane reductions, pivot decisions are wave-uniform scalar branches.  Same
// operation order as lu_solve_wave (dgetf2 + dgetrs), so the result is bit-identical.
template <int K>
MH_DEV int lu_small(const double* A, double& b)
{
  double a[K][K], x[K];
#pragma unroll
  for (int c = 0; c < K; c++)
#pragma unroll
    for (int r = 0; r < K; r++) a[r][c] = A[r + K * c];       // LDS broadcast reads
#pragma unroll
  for (int r = 0; r < K; r++) x[r] = read_lane(b, r);
#pragma unroll
  for (int j = 0; j < K; j++) {
    int jp = j; double amax = fabs(a[j][j]);
#pragma unroll
    for (int i = j + 1; i < K; i++) { const double v = fabs(a[i][j]); if (v > amax) { amax = v; jp = i; } }
    jp = uni(jp);
    if (ballot(!(amax != 0.0)) != 0ull) return j + 1;
#pragma unroll
    for (int i = j + 1; i < K; i++)
      if (jp == i) {
#pragma unroll
        for (int c = 0; c < K; c++) { const double t = a[j][c]; a[j][c] = a[i][c]; a[i][c] = t; }
        const double t = x[j]; x[j] = x[i]; x[i] = t;
      }
    if (j < K - 1) {
      const double piv = a[j][j];
      const bool big = fabs(piv) >= MH_SFMIN;
      const double r = 1.0 / piv;
#pragma unroll
      for (int i = j + 1; i < K; i++) {
        const double l = big ? a[i][j] * r : a[i][j] / piv;
        a[i][j] = l;
#pragma unroll
        for (int c = j + 1; c < K; c++) a[i][c] = a[i][c] - l * a[j][c];
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < K; kk++)
#pragma unroll
    for (int i = kk + 1; i < K; i++) x[i] = x[i] - x[kk] * a[i][kk];
#pragma unroll
  for (int kk = K - 1; kk >= 0; kk--) {
    x[kk] = x[kk] / a[kk][kk];
#pragma unroll
    for (int i = 0; i < kk; i++) x[i] = x[i] - x[kk] * a[i][kk];
  }
  const int lane = lane_id();
  double out = 0.0;
#pragma unroll
  for (int r = 0; r < K; r++) out = (lane == r) ? x[r] : out;
  b = out;
  return 0;
}

#ifndef MH_COLCACHE
#define MH_COLCACHE 4     /* 8 cost 36 more spilled VGPRs at the 128-register budget: -5 % */
#endif
// REGLU: allow the register-resident factorisations for k = 2..4 (the LDS call site; the HBM one keeps
// the code small and always takes the general routine -- same arithmetic, bit-identical results)
template <bool REGLU, class MatT>
MH_DEV int gather_and_solve(const MatT& M, double lam, uint64_t nbmask, int k, bool is_nb, int pos, double qi, double* A, double& b,
                            double (&colv)[MH_COLCACHE])
{
  const int lane = lane_id();
  unsigned long long t0 = lp_tick();
  if (k == 1) {
    // 1 x 1 system: dgetf2 finds the only pivot, dgetrs divides once (same arithmetic as the general path)
    const int j = ctz(nbmask);
    const double v = mat_at(M, j, lam);
    colv[0] = v;
    const double a00 = read_lane(v, j);
    const double rhs = read_lane(-qi, j);
    lp_tock(LP_GATHER, t0);
    if (!(fabs(a00) != 0.0)) return 1;
    b = (lane == 0) ? rhs / a00 : 0.0;
    return 0;
  }
  wave_sync();
  uint64_t m = nbmask;
#pragma unroll
  for (int c = 0; c < MH_COLCACHE; c++) {
    if (c < k) {
      const int j = ctz(m); m &= m - 1;
      const double v = mat_at(M, j, lam);
      colv[c] = v;
      if (is_nb) A[pos + k * c] = v;
    }
  }
  for (int c = MH_COLCACHE; c < k; c++) {
    const int j = ctz(m); m &= m - 1;
    const double v = mat_at(M, j, lam);
    if (is_nb) A[pos + k * c] = v;
  }
  // route -q[i] from the variable's lane to its row lane (nonbasic
  // variables to rows 0..k-1, everything else to the unused lanes above)
  b = push_to(-qi, is_nb ? pos : k + popc(~nbmask & lanes_below(lane)));
  wave_sync();
  lp_tock(LP_GATHER, t0);
  t0 = lp_tick();
  int info;
  if (REGLU && k == 2) info = lu_small<2>(A, b);
  else if (REGLU && k == 3) info = lu_small<3>(A, b);
#ifndef MH_NO_LU4
  else if (REGLU && k == 4) info = lu_small<4>(A, b);
#endif
  else info = REGLU ? lu_solve_wave(k, A, b) : lu_solve_wave_hbm(k, A, b);     // REGLU <=> the LDS call site
  lp_tock(LP_LU, t0);
  return info;
}

template <class MatT>
MH_DEV bool lcp_fast_wave(int n, const MatT& M, double lam, LuScratch S,
                          double qi, double& zi, int& zsize, double zero_tol,
                          double nrm_lam, WaveRand& rng, unsigned& pivots, Trace& tr)
{
  const int lane = lane_id();
  const bool valid = lane < n;
  const uint64_t vmask = lanes_below(n);
  if (zero_tol < 0.0) zero_tol = (double)n * nrm_lam * MH_DBL_EPS;
  uint64_t nbmask;
  if (zsize == n) {
    nbmask = ballot(valid && !(fabs(zi) < zero_tol));
  } else {
    double qmin; int minw;
    argmin_first(qi, valid, qmin, minw);
    if (qmin > -zero_tol) { zi = 0.0; zsize = n; pivots = 0; return true; }
    nbmask = bit(minw);
  }
  const unsigned MAX_PIV = 2u * (unsigned)n;
  for (pivots = 0; pivots < MAX_PIV; pivots++) {
    const int k = popc(nbmask);
    const bool is_nb = (nbmask >> lane) & 1ull;
    const bool is_b = valid && !is_nb;
    const int pos = popc(nbmask & lanes_below(lane));
    // gather _Msub (rows by position) and the rhs -q[nonbas]; the columns M(:, nonbas[c]) are
    // kept in registers for the product below
    double b = 0.0;
    double colv[MH_COLCACHE];
    if (k > 0) {
      // two call sites on purpose: with the pointers kept apart the LDS block is addressed with ds_ instructions,
      // a merged (generic) pointer turns every access of the factorisation into a FLAT memory operation
      int info;
      if (k <= S.ka) info = gather_and_solve<true>(M, lam, nbmask, k, is_nb, pos, qi, S.small, b, colv);
      else info = gather_and_solve<false>(M, lam, nbmask, k, is_nb, pos, qi, S.big, b, colv);
      if (info != 0) return false;
    }
    // w = Mmix * z + qbas on the basic lanes (dgemv column order)
    unsigned long long tg = lp_tick();
    double w = 0.0;
    {
      uint64_t m = nbmask;
#pragma unroll
      for (int c = 0; c < MH_COLCACHE; c++) {
        if (c < k) {
          m &= m - 1;
          const double t = read_lane(b, c);
          if (is_b) w = w + t * colv[c];
        }
      }
      for (int c = MH_COLCACHE; c < k; c++) {
        const int j = ctz(m); m &= m - 1;
        const double t = read_lane(b, c);
        const double v = mat_at(M, j, lam);
        if (is_b) w = w + t * v;
      }
      w = w + qi;
    }
    // z value of this variable (position -> variable routing)
    const double zv = __shfl(b, pos);
    lp_tock(LP_GEMV, tg);
    tg = lp_tick();
    const uint64_t bmask = vmask & ~nbmask;
    // both draws of the iteration (LCP.cpp:147 then :153/:172): every path takes the w draw when a basic
    // variable exists and then the z draw when a nonbasic one does, in that order
    double wsel = 0.0; int minw = -1;
    bool tie = false;
    if (bmask != 0ull) minw = rand_min_wave(w, bmask, zero_tol, rng, wsel, tie);
    double zsel = 0.0; int minz = -1;
    if (k > 0) minz = rand_min_wave(zv, nbmask, zero_tol, rng, zsel, tie);
    lp_tock(LP_RANDMIN, tg);
    if (minw < 0 || wsel > -zero_tol) {
      if (minz >= 0 && zsel < -zero_tol) {
        nbmask &= ~bit(minz);
        tr.push(-(int32_t)(minz + 1));
      } else {
        zi = is_nb ? zv : 0.0;
        zsize = n;
        return true;
      }
    } else {
      const uint64_t nb_new = nbmask | bit(minw);
      tr.push((int32_t)(minw + 1));
      nbmask = nb_new;
      if (minz >= 0 && zsel < -zero_tol) {
        // LCP.cpp:176-187: the POSITION found in the old _z indexes the NEW,
        // re-sorted _nonbas
        const int posz = popc((nb_new & ~bit(minw)) & lanes_below(minz));
        const int idx2 = nth_set_bit(nb_new, posz);
        nbmask &= ~bit(idx2);
        tr.push(-(int32_t)(idx2 + 1));
#ifndef MH_NO_REPEAT_SKIP
        // Repeating pivot sequences.  The position rule above can send the variable that has just entered straight out again.  The set
        // is then what it was at the top of this iteration, and an iteration is a function of the set alone unless a draw decided
        // something (rand_min with several minima): every remaining iteration repeats this one, and the loop spins on one basis until
        // MAX_PIV.  That is how lcp_fast fails on resting stacks, and what the slow worlds of a long run spend their time on (80 % of
        // their lcp_fast iterations at step 4200 of the sphere-stack batch, tests/tools/slow_world_diag.py).  What the repetitions leave
        // behind is known without running them: two draws and two trace entries each, z untouched.  Period 1 only here -- the ring of
        // sets that periods 2-3 need (a fifth of the repeats in box stacks, none in the sphere stacks) cost the headline kernel 6 %;
        // mh_lcp_block.h, for n > 64, recognises periods up to 8.
        if (idx2 == minw && !tie) {
          const unsigned rest = MAX_PIV - (pivots + 1u);
          rng.skip(2u * rest);
          if (tr.buf && tr.len < tr.cap) for (unsigned r = 0; r < rest; r++) { tr.push((int32_t)(minw + 1)); tr.push(-(int32_t)(idx2 + 1)); }
          else tr.len += (int)(2u * rest);
          pivots = MAX_PIV;
          return false;
        }
#endif
      }
    }
  }
  return false;
}

// solution check shared by the regularised wrappers
// (LCP.cpp:240-249 strict=false; :303-312 strict=true, against M + lam*I)
template <class MatT>
MH_DEV bool verify_wave(int n, const MatT& M, double lam, double qi, double zi, double ZERO_TOL, bool strict)
{
  // "min_element(v) >= -T" holds iff no element violates it: one ballot per test instead
  // of a wave reduction (LCP.cpp:240-249 with >=, :303-312 with >)
  const int lane = lane_id();
  const bool valid = lane < n;
  const double nT = -ZERO_TOL;
  // (a NaN among the values -- a world whose velocities have overflowed -- takes the tests through std::min_element's own semantics, as the reference's
  //  *std::min_element(...) >= -ZERO_TOL does: skipped unless it is the first element)
  if (ballot(valid && zi != zi) != 0ull) { const double mz = min_element_value(zi, valid); if (!(strict ? (mz > nT) : (mz >= nT))) return false; }
  else if (ballot(valid && !(strict ? (zi > nT) : (zi >= nT))) != 0ull) return false;
  double w = 0.0;
  uint64_t nz = ballot(valid && zi != 0.0);
  while (nz) {
    const int c = ctz(nz); nz &= nz - 1;
    const double t = read_lane(zi, c);
    const double m = mat_at(M, c, lam);
    if (valid) w = w + t * m;
  }
  w = w + qi;
  const double zw = zi * w;
  if (ballot(valid && (w != w || zw != zw)) != 0ull) {
    const double mw = min_element_value(w, valid);
    if (!(strict ? (mw > nT) : (mw >= nT))) return false;
    const double mn = min_element_value(zw, valid), mx = max_element_value(zw, valid);
    return (strict ? (mn > nT) : (mn >= nT)) && mx < ZERO_TOL;
  }
  if (ballot(valid && !(strict ? (w > nT) : (w >= nT))) != 0ull) return false;
  if (ballot(valid && !(strict ? (zw > nT) : (zw >= nT))) != 0ull) return false;
  return ballot(valid && !(zw < ZERO_TOL)) == 0ull;
}

template <bool HBM, class MatT>
MH_DEV int lemke_gather_and_solve(int n, const MatT& M, double lam, int bv, int t, const double* art, double* A, double& d)
{
  const int lane = lane_id();
  const bool valid = lane < n;
  wave_sync();
  for (int p = 0; p < n; p++) {
    const int id = read_lane(bv, p);
    double a;
    if (id == t) a = valid ? art[lane] : 0.0;
    else if (id >= n) a = (lane == id - n) ? -1.0 : 0.0;
    else a = mat_at(M, id, lam);
    if (valid) A[lane + n * p] = a;
  }
  wave_sync();
  return HBM ? lu_solve_wave_hbm(n, A, d) : lu_solve_wave(n, A, d);
}

template <class MatT>
MH_DEV bool lcp_lemke_wave(int n, const MatT& M, double lam, LuScratch S, double* art,
                           double qi, double& zi, int& zsize, double piv_tol, double zero_tol,
                           double nrm_lam, WaveRand& rng, unsigned& pivots, Trace& tr)
{
  const int lane = lane_id();
  const bool valid = lane < n;
  const double INF = __longlong_as_double(0x7ff0000000000000ll);
  const unsigned MAXITER = (50u * (unsigned)n < 1000u) ? 50u * (unsigned)n : 1000u;
  pivots = 0;
  zi = 0.0;                                   // z.set_zero() (:564)
  const int z0size = zsize;
  if (zero_tol <= 0.0) zero_tol = MH_DBL_EPS * nrm_lam * (double)n;
  const double qmin = wave_min(valid ? qi : INF);
  if (qmin > -zero_tol) { zsize = n; return true; }
  zsize = 2 * n;                              // z.set_zero(2n) (:596)
  const int t = 2 * n;
  if (z0size != n) for (int i = 0; i < n; i++) (void)rng.next();   // _restart_z0 (:618-620)
  int bv = n + lane;                          // all w variables basic, B = -I
  double x = valid ? qi : 0.0;
  if (ballot(valid && x < 0.0) == 0ull) { zsize = n; return true; }  // (:737)
  const double PIV_TOL = (piv_tol > 0.0) ? piv_tol : MH_DBL_EPS * (double)n * ((nrm_lam > 1.0) ? nrm_lam : 1.0);
  double xmin; int lvindex;
  argmin_first(x, valid, xmin, lvindex);
  const double tval = -xmin;
  int leaving = n + lvindex;
  int entering = t;
  double u = (valid && x < 0.0) ? 1.0 : 0.0;
  wave_sync();
  if (valid) art[lane] = u;                   // Be = -(Bl*u) with Bl = -I
  u = u * tval;
  x = x + u;
  if (lane == lvindex) { x = tval; bv = t; }
  wave_sync();
  for (pivots = 0; pivots < MAXITER; pivots++) {
    if (leaving == t) {
      // z[_bas[p]] = x[p] (:804-806): route x from position lanes to variable lanes
      // (art is dead once t has left the basis: reuse it as the routing buffer)
      wave_sync();
      if (valid) art[lane] = 0.0;
      wave_sync();
      if (valid && bv < n) art[bv] = x;
      wave_sync();
      zi = valid ? art[lane] : 0.0;
      zsize = n;
      return true;
    }
    double be;
    if (leaving < n) { entering = n + leaving; be = (lane == leaving) ? -1.0 : 0.0; }
    else { entering = leaving - n; be = valid ? mat_at(M, entering, lam) : 0.0; }
    // gather Al = Bl from the basis description and solve Al d = Be
    double d = be;
    {
      int info;
      if (n <= S.ka) info = lemke_gather_and_solve<false>(n, M, lam, bv, t, art, S.small, d);
      else info = lemke_gather_and_solve<true>(n, M, lam, bv, t, art, S.big, d);
      if (info != 0) return false;                                  // singular basis (:840-850), size stays 2n
    }
    const uint64_t jm = ballot(valid && d > PIV_TOL);
    if (jm == 0ull) return false;                              // ray termination (:892-903)
    const bool inj = (jm >> lane) & 1ull;
    // theta = *std::min_element(ratios) (LCP.cpp:920): a NaN ratio is skipped by the scan unless it is the FIRST candidate's, which
    // then stays the "minimum" and empties the set below -- a degenerate basis late in a regularised attempt does produce that
    const double rat = inj ? (x + zero_tol) / d : INF;
    { const double r0 = read_lane(rat, ctz(jm)); if (r0 != r0) { zsize = n; return false; } }
    const double theta = wave_min(rat);
    const uint64_t keep = ballot(inj && (x / d <= theta));
    if (keep == 0ull) { zsize = n; return false; }            // (:946-958)
    const uint64_t tm = ballot(valid && bv == t);
    lvindex = (keep & tm) ? ctz(tm) : ctz(keep);
    leaving = read_lane(bv, lvindex);
    const double ratio = read_lane(x, lvindex) / read_lane(d, lvindex);
    d = d * ratio;
    x = x - d;
    if (lane == lvindex) { x = ratio; bv = entering; }
    tr.push((int32_t)entering + 1); tr.push((int32_t)leaving + 1);
  }
  zsize = n;
  return false;
}

// kind selectors: MH_LCP_* of include/moby_hip.h

struct LcpParams { int kind; int min_exp; unsigned step_exp; int max_exp; double piv_tol; double zero_tol; };

// 10^rf exactly as std::pow(10.0, rf) rounds it for the exponents the
// handlers use (-20..1); table produced by the host's libm at load time.
struct Pow10Table { double v[64]; }; // index rf + 32

// Dispatch over the four public solvers.  nrm0 = norm_inf(M) (max |m|), dii =
// this lane's diagonal entry (for norm_inf of M + lam*I on the ladder).
template <class MatT>
MH_DEV bool lcp_solve_wave(const LcpParams& P, const Pow10Table& p10, int n, const MatT& M, LuScratch A, double* art,
                           double nrm0, double dii, double qi, double& zi, int& zsize,
                           WaveRand& rng, unsigned& pivots, Trace& tr)
{
  const bool valid = lane_id() < n;
  const bool reg = (P.kind == MH_LCP_FAST_REG) || (P.kind == MH_LCP_LEMKE_REG);
  const bool fast = (P.kind == MH_LCP_FAST) || (P.kind == MH_LCP_FAST_REG);
  // plain norm_inf(M) -- the unregularised matrix -- sets ZERO_TOL (LCP.cpp:228,369)
  const double ZERO_TOL = (P.zero_tol > 0.0) ? P.zero_tol : (double)n * nrm0 * MH_NEAR_ZERO;
  unsigned total = 0;
  double offmax = 0.0;
  // attempt 0 is the unregularised solve (verified against M with >=, LCP.cpp:236-256),
  // attempts 1.. walk the ladder (verified against M + lambda I with >, :281-340); one
  // loop so that each solver is instantiated once
  int rf = P.min_exp;
  for (int attempt = 0; ; attempt++) {
    double lam = 0.0, nrm = nrm0;
    if (attempt > 0) {
      if (!reg || !(rf < P.max_exp)) break;
      if (attempt == 1) offmax = M.offdiag_max();
      lam = p10.v[rf + 32];
      nrm = norm_reg(offmax, dii, valid, lam);
    }
    if (reg) tr.push(0x40000000 | attempt);
    unsigned long long tl = lp_tick();
    const bool ok = fast ? lcp_fast_wave(n, M, lam, A, qi, zi, zsize, P.zero_tol, nrm, rng, pivots, tr)
                         : lcp_lemke_wave(n, M, lam, A, art, qi, zi, zsize, P.piv_tol, P.zero_tol, nrm, rng, pivots, tr);
    if (!fast) lp_tock(LP_LEMKE, tl);
    if (!reg) return ok;                                      // plain lcp_fast / lcp_lemke
    tl = lp_tick();
    const bool good = ok && verify_wave(n, M, lam, qi, zi, ZERO_TOL, attempt > 0);
    lp_tock(LP_VERIFY, tl);
    if (attempt == 0) { if (good) return true; total += pivots; }
    else { total += pivots; if (good) { pivots = total; return true; } rf += (int)P.step_exp; }
  }
  pivots = total;
  return false;
}

} // namespace mh
