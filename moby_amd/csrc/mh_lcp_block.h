// Dense LCP solvers for n > 64: one 256-thread workgroup per problem, M read in
// place from HBM, vectors / index sets / the LU scratch in a per-problem HBM
// workspace (L2-resident for the sizes the handlers produce), LDS only for the
// block reductions.
//
// Same algorithms and the same floating-point operation order as the wave
// solvers in mh_lcp_wave.h (and so as the CPU oracle oracle/lcp.hpp):
//   LCP::lcp_fast               /root/reference/src/LCP.cpp:41-196
//   LCP::rand_min               src/LCP.cpp:199-209
//   LCP::lcp_fast_regularized   src/LCP.cpp:212-350
//   LCP::lcp_lemke_regularized  src/LCP.cpp:353-487
//   LCP::lcp_lemke (dense)      src/LCP.cpp:545-1003
//   LinAlgd::solve_fast (dgesv = dgetf2 + dgetrs) call sites src/LCP.cpp:120,838
//
// Mapping: thread t owns rows t, t+256, ...; the sorted index vectors
// _bas/_nonbas are a flag array + a list rebuilt by a block scan; every control
// decision is block-uniform (broadcast through LDS), the libc rand() ring lives
// in LDS and is advanced by thread 0.  The LU is right-looking in panels of 8 columns:
// the panel lives in LDS and is factorised by ONE wave (wave reductions + wavefront
// fences, no block barriers), its row swaps reach the rest of the matrix once per
// panel, the trailing matrix is updated once per panel by a (64 rows x 4 column
// groups) thread grid with four loads in flight, the triangular solves take 8
// columns per barrier round with the diagonal blocks on lanes.  Measured (n = 256,
// 512 pivots in the slowest problem): 296 ms -> 177 ms; per-phase cycles with
// -DMH_BLK_PROF: panel 30 %, trailing update 27 %, solves 12 %.
// No include guard: one translation unit per thread geometry (mh_lcp_blk.hip, mh_lcp_blkw.hip) defines MH_BLK_NS / MH_BLK_T / ...
// and MH_BLK_LAUNCHER and includes this file; each compiles two kernels (the lcp_fast kinds, the lcp_lemke kinds).
#include <type_traits>
#include <utility>
#include "mh_lcp_wave.h"
#include "mh_host.h"

#if !defined(MH_BLK_NS) || !defined(MH_BLK_T) || !defined(MH_BLK_UCH) || !defined(MH_BLK_PANEL_CAP) || !defined(MH_BLK_KATTR) || !defined(MH_BLK_CN)
#error "define MH_BLK_NS, MH_BLK_T, MH_BLK_UCH, MH_BLK_PANEL_CAP, MH_BLK_CN and MH_BLK_KATTR before including mh_lcp_block.h"
#endif
namespace mh { namespace MH_BLK_NS {

constexpr int T = MH_BLK_T;              // threads per problem (a multiple of 64)
constexpr int NW = T / 64;               // waves per problem

struct Ws {
  double* A;      // n x n LU scratch (col-major, ld = k)
  double* b;      // rhs / solution by position
  double* w;      // w by variable
  double* x;      // Lemke: basic values by position
  double* d;      // Lemke: direction by position
  double* art;    // Lemke: artificial column
  int* list;      // sorted nonbasic variables
  int* flag;      // 1 = nonbasic
  int* pos;       // position of variable i in list (-1 if basic)
  int* bv;        // Lemke: basis by position
};
MH_DEV size_t ws_doubles(int n) { return (size_t)n * n + 5 * (size_t)n; }
MH_DEV size_t ws_ints(int n) { return 4 * (size_t)n; }

// SURVEY 8(d)'s model of one factorisation (LCP.cpp:120, :837-838): dgesv of a k x k matrix = 2/3 k^3 flops over 8 k^2 bytes,
// whatever this build skips of it; summed per problem for the roofline figures of bench.py (thread 0 only)
// [2]: the flops the factorisation routines really ISSUE (every lane-operation of their update loops, masked rows included; the exact
// zeros they skip are not in it), [3]: ticks of the constant-rate wall clock this workgroup spent on the problem
__shared__ double s_work[MH_WORK];
MH_DEV void account_lu(int k) { if (threadIdx.x == 0) { const double kk = (double)k; s_work[0] += (2.0 / 3.0) * kk * kk * kk; s_work[1] += 8.0 * kk * kk; } }
MH_DEV void account_issued(double flops) { if (threadIdx.x == 0) s_work[2] += flops; }
__shared__ double s_bd[4];
__shared__ int s_bi[4];
__shared__ unsigned s_rng[32];

MH_DEV int tid() { return (int)threadIdx.x; }
// -DMH_BLK_PROF: per-phase cycle totals of block 0, printed by the kernel (diagnostic builds only)
enum { BP_LIST = 0, BP_GATHER, BP_PANEL, BP_SWAP, BP_TRAIL, BP_SOLVE, BP_GEMV, BP_RANDMIN, BP_COMPACT, BP_C_SETUP, BP_C_PANEL, BP_C_U12, BP_C_TRAIL, BP_C_BACK, BP_C_STEPS, BP_C_PANELS, BP_COUNT };
#ifdef MH_BLK_PROF
__shared__ unsigned long long s_prof[BP_COUNT];
MH_DEV unsigned long long bp_tick() { return __builtin_amdgcn_s_memtime(); }
MH_DEV void bp_tock(int ph, unsigned long long t0) { const unsigned long long d = __builtin_amdgcn_s_memtime() - t0; if (threadIdx.x == 0) s_prof[ph] += d; }
#else
MH_DEV unsigned long long bp_tick() { return 0ull; }
MH_DEV void bp_tock(int, unsigned long long) {}
#endif
MH_DEV void sync() { __syncthreads(); }
// barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding global stores (s_waitcnt vmcnt(0)),
// a microsecond each time one is in flight; where the threads exchange nothing through global memory this one is enough
#ifdef MH_BLK_NO_LSYNC
MH_DEV void lsync() { __syncthreads(); }
#else
MH_DEV void lsync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
// ordering point between the lanes of ONE wave of the block (compiler ordering; ds_ ops of a wave execute in order)
MH_DEV void wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
MH_DEV double inf() { return __longlong_as_double(0x7ff0000000000000ll); }

__shared__ double s_wd[NW];
__shared__ int s_wi[NW];
MH_DEV int wave_of() { return tid() >> 6; }
// lexicographic (value ascending, index ascending) minimum over the block: DPP reductions inside each of
// the four waves, four partials through LDS -- two barriers instead of a nine-level tree
MH_DEV void red_min_first(double v, int idx, double& vout, int& iout) {
  const double wm = wave_min(v);
  const double wi = wave_min((v == wm) ? (double)idx : 1.0e300);     // indices are exact in a double
  if (lane_id() == 0) { s_wd[wave_of()] = wm; s_wi[wave_of()] = (wi < 1.0e299) ? (int)wi : 0x7fffffff; }
  sync();
  double bv = s_wd[0]; int bi = s_wi[0];
#pragma unroll
  for (int w = 1; w < T / 64; w++) { const double v2 = s_wd[w]; const int i2 = s_wi[w]; if (v2 < bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; } }
  vout = bv; iout = bi; sync();
}
MH_DEV void red_max_first(double v, int idx, double& vout, int& iout) {
  const double wm = wave_max(v);
  const double wi = wave_min((v == wm) ? (double)idx : 1.0e300);
  if (lane_id() == 0) { s_wd[wave_of()] = wm; s_wi[wave_of()] = (wi < 1.0e299) ? (int)wi : 0x7fffffff; }
  sync();
  double bv = s_wd[0]; int bi = s_wi[0];
#pragma unroll
  for (int w = 1; w < T / 64; w++) { const double v2 = s_wd[w]; const int i2 = s_wi[w]; if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; } }
  vout = bv; iout = bi; sync();
}
MH_DEV double red_max(double v) { double o; int i; red_max_first(v, 0, o, i); return o; }
MH_DEV double red_min(double v) { double o; int i; red_min_first(v, 0, o, i); return o; }
MH_DEV int wave_sum_int(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
MH_DEV int red_sum_int(int v) {
  v = wave_sum_int(v);
  if (lane_id() == 0) s_wi[wave_of()] = v;
  sync();
  int o = 0;
#pragma unroll
  for (int w = 0; w < T / 64; w++) o += s_wi[w];
  sync();
  return o;
}
// exclusive prefix sum of one int per thread (thread order); total returned uniformly
MH_DEV int excl_scan_int(int v, int& total) {
  const int lane = lane_id();
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
  if (lane == 63) s_wi[wave_of()] = inc;
  sync();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < T / 64; w++) { const int x = s_wi[w]; if (w < wave_of()) base += x; tot += x; }
  sync();
  total = tot;
  return base + inc - v;
}
MH_DEV int bcast_i(int v) { if (tid() == 0) s_bi[0] = v; sync(); const int o = s_bi[0]; sync(); return o; }
MH_DEV int rand_next() {           // block-uniform; glibc TYPE_3 ring in LDS (oracle/glibc_rand.h layout)
  if (tid() == 0) {
    const unsigned idx = s_rng[31];
    const unsigned v = s_rng[idx] + s_rng[(idx + 28) % 31];
    s_rng[idx] = v; s_rng[31] = (idx + 1) % 31;
    s_bi[1] = (int)(v >> 1);
  }
  sync(); const int o = s_bi[1]; sync();
  return o;
}

struct Trace2 { int32_t* buf; int cap; int len;
  MH_DEV void push(int32_t v) { if (buf && len < cap && tid() == 0) buf[len] = v; len++; } };

struct Mat { const double* M; int ld; int n;
  MH_DEV double at(int r, int c, double lam) const { const double m = M[r + (size_t)ld * c]; return (r == c) ? m + lam : m; } };

// sorted list of the flagged variables + their positions; returns the count
MH_DEV int build_list(int n, const Ws& W) {
  const int t = tid();
  const int per = (n + T - 1) / T, lo = t * per, hi = (lo + per < n) ? lo + per : n;
  int c = 0;
  for (int i = lo; i < hi; i++) c += W.flag[i] ? 1 : 0;
  int total;
  int o = excl_scan_int(c, total);
  for (int i = lo; i < hi; i++) { if (W.flag[i]) { W.list[o] = i; W.pos[i] = o; o++; } else W.pos[i] = -1; }
  sync();
  return total;
}

// dgesv, one rhs: A k x k col-major (ld = k) and b in the workspace.  Returns LAPACK info (uniform).
//
// Right-looking LU in panels of NB columns.  Every element receives exactly the updates
// a <- a - l*u of dgetf2, in the same order (column steps ascending, each product rounded on its own),
// so the factors are bit-identical to the unblocked routine.  What changes is where the work happens:
//   * the panel (rows below the diagonal block x NB columns) is factorised in LDS when it fits
//     (PANEL_CAP doubles): pivot search, scaling and the in-panel updates run at LDS latency;
//   * the trailing matrix is read and written once per PANEL, with the NB multipliers of a row in
//     registers and the NB pivot rows staged through LDS (s_u), instead of once per column;
//   * row swaps outside the panel are plain global-memory swaps off the critical path.
#ifndef MH_BLK_NB
#define MH_BLK_NB 8
#endif
constexpr int NB = MH_BLK_NB;           // panel width.  8 everywhere today; what a wider panel would still have to change is asserted where it is assumed
constexpr int UCH = MH_BLK_UCH;          // columns of the pivot-row block staged in LDS at a time
#ifndef MH_BLK_TCOLS
#define MH_BLK_TCOLS 8
#endif
constexpr int TCOLS = MH_BLK_TCOLS;      // trailing-update columns a thread has in flight
constexpr int PANEL_CAP = MH_BLK_PANEL_CAP;   // doubles (14 / 28 KB): rows x NB of the panel held in LDS
__shared__ double s_u[NB][UCH + 8];          // (+ 8: the padding group of the compact LU's update lists)
__shared__ int s_li[UCH + 16];
#ifndef MH_BLK_LIST_CAP
#define MH_BLK_LIST_CAP 1024
#endif
constexpr int LIST_CAP = MH_BLK_LIST_CAP;
__shared__ int s_list[LIST_CAP];
__shared__ double s_panel[PANEL_CAP];
#ifndef MH_BLK_RHS_CAP
#define MH_BLK_RHS_CAP 1024
#endif
constexpr int RHS_CAP = MH_BLK_RHS_CAP;  // the right-hand side stays in LDS for the whole factorisation when k fits
__shared__ double s_b[RHS_CAP];
__shared__ int s_ipiv[NB];
// Exact skips of the trailing update.  a <- a - l*u leaves a as it is (up to the sign of a zero) when l == 0 or
// u == 0 and the other factor is finite.  The matrices this solver sees are mostly zeros -- Lemke's basis is identity
// columns plus a few columns of M, and the impact LCP's M is block-sparse (two contacts couple only through a shared
// body) -- so per panel: rows whose NB multipliers are all zero and columns whose NB pivot-row entries are all zero
// are neither read nor written.  Flags: [0] every multiplier of the panel below its diagonal block is zero,
// [1] ... is finite, [2 + parity] every pivot-row entry of the current column chunk is finite.
__shared__ int s_sk[4];
__shared__ unsigned char s_nz[UCH];
MH_DEV bool finite_d(double x) { return fabs(x) <= 1.7976931348623157e308; }
// a - l*u of dgesv.  -DMH_BLK_FMA_EXPERIMENT (timing / outcome experiment only, DESIGN 8 "the oracle's dgesv above n = 64"): fused, one rounding -- what
// oracle_dbg_lu_fma makes of the oracle; the shipped build and the oracle's definition are the unfused form at every size
#ifdef MH_BLK_FMA_EXPERIMENT
MH_DEV double blk_upd(double a, double l, double u) { return __builtin_fma(-l, u, a); }
#else
MH_DEV double blk_upd(double a, double l, double u) { return a - l * u; }
#endif

// A22[r][c] -= L[r][j0+s] * U12[s][c], s ascending; one row per thread, its multipliers in registers.
// Thread (tr, tc): rows c0 + tr + 64 i, columns tc, tc + 4, ... of the chunk -- a wave still reads 64 consecutive
// rows of a column; TCOLS columns are in flight per thread before any is used.  FULL (a whole panel of NB columns,
// every panel but the last) and INLDS are compile-time so that the NB updates of an element are straight-line code:
// with the `s < nbk` tests left in, every update was a ds_read + s_waitcnt + branch chain.  (Tried on top: requesting the
// next TCOLS columns before updating the current ones -- 3 % slower; TCOLS 4 / 8 / 16 -- within 1 %.)
template <bool FULL, bool INLDS>
MH_DEV void trail_update(double* __restrict__ A, int k, int j0, int c0, int cb, int ncb, int nbk, int R, bool Ufin)
{
  const int t = tid();
  const int tr = t & 63, tc = t >> 6;
  for (int r = c0 + tr; r < k; r += 64) {
    double l[NB];
#pragma unroll
    for (int s2 = 0; s2 < NB; s2++) l[s2] = (FULL || s2 < nbk) ? (INLDS ? s_panel[(r - j0) + R * s2] : A[r + (size_t)k * (j0 + s2)]) : 0.0;
    if (Ufin) {
      bool lz = true;
#pragma unroll
      for (int s2 = 0; s2 < NB; s2++) lz = lz && (l[s2] == 0.0);
      if (lz) continue;                              // this row's multipliers are all zero
    }
    for (int cq = tc; cq < ncb; cq += NW * TCOLS) {
      double a[TCOLS]; bool go[TCOLS];
#pragma unroll
      for (int u4 = 0; u4 < TCOLS; u4++) { const int c = cq + NW * u4; go[u4] = (c < ncb) && s_nz[c]; a[u4] = go[u4] ? A[r + (size_t)k * (cb + c)] : 0.0; }
#pragma unroll
      for (int u4 = 0; u4 < TCOLS; u4++) {
        const int c = cq + NW * u4;
        if (go[u4]) {
          double v = a[u4];
#pragma unroll
          for (int s2 = 0; s2 < NB; s2++) if (FULL || s2 < nbk) {
            v = blk_upd(v, l[s2], s_u[s2][c]);      /* (fused under -DMH_BLK_FMA_EXPERIMENT: what an FP64 MFMA trailing update would compute) */
          }
          A[r + (size_t)k * (cb + c)] = v;
        }
      }
    }
  }
}

MH_DEV int lu_solve(int k, double* A, double* b) {
  const int t = tid();
  double* bb = b;
  const bool b_lds = k <= RHS_CAP;
  if (b_lds) { for (int i = t; i < k; i += T) s_b[i] = b[i]; sync(); bb = s_b; }
  for (int j0 = 0; j0 < k; j0 += NB) {
    const int nbk = (k - j0 < NB) ? k - j0 : NB;
    const int R = k - j0;                                   // panel rows (global rows j0 .. k-1)
    const bool in_lds = R * nbk <= PANEL_CAP;
    // panel element (global row j0 + r, global column j0 + c): LDS s_panel[r + R*c] or A
    unsigned long long tp = bp_tick();
    if (in_lds) {
      for (int c = 0; c < nbk; c++) for (int r = t; r < R; r += T) s_panel[r + R * c] = A[(j0 + r) + (size_t)k * (j0 + c)];
      sync();
    }
    if (in_lds) {
      // The panel is factorised by wave 0 alone: its columns cost wave-level reductions and wavefront-scope
      // fences (the LDS unit executes one wave's ds_ operations in order) instead of ~6 block barriers each;
      // the other three waves wait at the barrier below.
      if (t == 0) s_bi[0] = 0;
      sync();
      if (t < 64) {
        const int lane = t;
        for (int jj = 0; jj < nbk; jj++) {
          const int j = j0 + jj;
          double best = -1.0; int bi = 0x7fffffff;
          for (int r = jj + lane; r < R; r += 64) { const double a = fabs(s_panel[r + R * jj]); if (a > best) { best = a; bi = j0 + r; } }
          const double amax = wave_max(best);
          const double wi = wave_min((best == amax) ? (double)bi : 1.0e300);
          const int jp = (wi < 1.0e299) ? (int)wi : 0x7fffffff;
          if (!(amax != 0.0)) { if (lane == 0) s_bi[0] = j + 1; break; }
          if (lane == 0) s_ipiv[jj] = jp;
          if (jp != j) {
            if (lane < nbk) { const double t0 = s_panel[jj + R * lane], t1 = s_panel[(jp - j0) + R * lane]; s_panel[jj + R * lane] = t1; s_panel[(jp - j0) + R * lane] = t0; }
            if (lane == 63) { const double t0 = bb[j]; bb[j] = bb[jp]; bb[jp] = t0; }
            wsync();
          }
          if (j < k - 1) {
            const double piv = s_panel[jj + R * jj];
            const bool big = fabs(piv) >= MH_SFMIN;
            const double rcp = 1.0 / piv;
            bool lfin = true;
            for (int r = jj + 1 + lane; r < R; r += 64) { double l = s_panel[r + R * jj]; l = big ? l * rcp : l / piv; s_panel[r + R * jj] = l; lfin = lfin && finite_d(l); }
            wsync();
            const bool all_fin = mh::ballot(!lfin) == 0ull;
            for (int c = jj + 1; c < nbk; c++) { const double u = s_panel[jj + R * c];
              if (u == 0.0 && all_fin) continue;              // a - l*0 = a: exact no-op (see s_sk above)
              for (int r = jj + 1 + lane; r < R; r += 64) s_panel[r + R * c] = blk_upd(s_panel[r + R * c], s_panel[r + R * jj], u); }
            wsync();
          }
        }
      }
      sync();
      const int info = s_bi[0];
      sync();
      if (info != 0) return info;
    } else
    for (int jj = 0; jj < nbk; jj++) {
      const int j = j0 + jj;
      double best = -1.0; int bi = 0x7fffffff;
      for (int r = j + t; r < k; r += T) { const double a = fabs(A[r + (size_t)k * j]); if (a > best) { best = a; bi = r; } }
      double amax; int jp; red_max_first(best, bi, amax, jp);
      if (!(amax != 0.0)) return j + 1;
      if (jp != j) {                                       // full-row swap, as dgetf2 (dlaswp on both sides)
        for (int c = t; c < k; c += T) { const double t0 = A[j + (size_t)k * c], t1 = A[jp + (size_t)k * c]; A[j + (size_t)k * c] = t1; A[jp + (size_t)k * c] = t0; }
        if (t == 0) { const double t0 = bb[j]; bb[j] = bb[jp]; bb[jp] = t0; }
        sync();
      }
      if (j < k - 1) {
        const double piv = A[j + (size_t)k * j];
        const bool big = fabs(piv) >= MH_SFMIN;
        const double rcp = 1.0 / piv;
        for (int r = j + 1 + t; r < k; r += T) { double l = A[r + (size_t)k * j]; l = big ? l * rcp : l / piv; A[r + (size_t)k * j] = l; }
        sync();
        if (nbk - jj - 1 > 0) {
          for (int c = j + 1; c < j0 + nbk; c++) { const double u = A[j + (size_t)k * c];
            for (int r = j + 1 + t; r < k; r += T) A[r + (size_t)k * c] = blk_upd(A[r + (size_t)k * c], A[r + (size_t)k * j], u); }
          sync();
        }
      }
    }
    bp_tock(BP_PANEL, tp); tp = bp_tick();
    if (in_lds) {                                           // L and U of the panel back to the workspace (the solves read them)
      for (int c = 0; c < nbk; c++) for (int r = t; r < R; r += T) A[(j0 + r) + (size_t)k * (j0 + c)] = s_panel[r + R * c];
      // dlaswp of the columns outside the panel: one column per thread, the panel's swaps in sequence
      sync();
      for (int c = t; c < k; c += T) {
        if (c >= j0 && c < j0 + nbk) continue;
        for (int jj = 0; jj < nbk; jj++) { const int jp = s_ipiv[jj], j = j0 + jj;
          if (jp != j) { const double t0 = A[j + (size_t)k * c], t1 = A[jp + (size_t)k * c]; A[j + (size_t)k * c] = t1; A[jp + (size_t)k * c] = t0; } }
      }
      sync();
    }
    bp_tock(BP_SWAP, tp); tp = bp_tick();
    const int c0 = j0 + nbk;                               // first trailing column
    if (c0 >= k) break;
    account_issued(2.0 * (double)nbk * (double)(k - c0) * (double)(k - c0) + (double)nbk * (double)nbk * (double)(k - j0));
    if (t == 0) { s_sk[0] = 1; s_sk[1] = 1; s_sk[2] = 1; s_sk[3] = 1; }
    sync();
    { bool z = true, f = true;
      for (int s2 = 0; s2 < nbk; s2++)
        for (int r = c0 + t; r < k; r += T) {
          const double l = in_lds ? s_panel[(r - j0) + R * s2] : A[r + (size_t)k * (j0 + s2)];
          if (l != 0.0) z = false;
          if (!finite_d(l)) f = false;
        }
      if (!z) s_sk[0] = 0;
      if (!f) s_sk[1] = 0; }
    sync();
    const bool Lz = s_sk[0] != 0, Lfin = s_sk[1] != 0;
    int par = 0;
    // ---- trailing columns, UCH at a time ----
    for (int cb = c0; cb < k; cb += UCH, par ^= 1) {
      const int ncb = (k - cb < UCH) ? k - cb : UCH;
      // pivot-row block U12[s][c] = A[j0+s][c] - sum_{s' < s} L[j0+s][j0+s'] * U12[s'][c]  (updates of steps j0.. in order)
      for (int c = t; c < ncb; c += T) {
        double u[NB];
        bool nz = false, fin = true;
#pragma unroll
        for (int s2 = 0; s2 < NB; s2++) {
          {                                                // nbk == NB here (see trail_update)
            double a = A[(j0 + s2) + (size_t)k * (cb + c)];
#pragma unroll
            for (int s1 = 0; s1 < NB; s1++) if (s1 < s2) a = blk_upd(a, (in_lds ? s_panel[s2 + R * s1] : A[(j0 + s2) + (size_t)k * (j0 + s1)]), u[s1]);
            u[s2] = a;
            A[(j0 + s2) + (size_t)k * (cb + c)] = a;
            s_u[s2][c] = a;
            nz = nz || (a != 0.0);
            fin = fin && finite_d(a);
          }
        }
        s_nz[c] = (nz || !Lfin) ? 1 : 0;                   // 0: this column's update is an exact no-op
        if (!fin) s_sk[2 + par] = 0;
      }
      sync();
      const bool Ufin = s_sk[2 + par] != 0;
      if (t == 0) s_sk[2 + (par ^ 1)] = 1;                 // re-arm the other chunk's flag (idle during this phase)
      if (!(Lz && Ufin)) {
        // a trailing matrix exists only behind a full panel (nbk < NB is the last panel, c0 == k)
        if (in_lds) trail_update<true, true>(A, k, j0, c0, cb, ncb, nbk, R, Ufin); else trail_update<true, false>(A, k, j0, c0, cb, ncb, nbk, R, Ufin);
      }
      sync();
    }
    bp_tock(BP_TRAIL, tp);
  }
  const unsigned long long ts = bp_tick();
  // triangular solves (the rhs is in LDS when k <= RHS_CAP)
  // SNB columns per round of two barriers: the SNB x SNB diagonal block is staged in LDS and finished by one wave (lane r = row r, values broadcast
  // with v_readlane), the rows outside it then take their SNB updates in the same (ascending / descending) column order as dgetrs -- and while they
  // do, the NEXT diagonal block is already on its way from the workspace (it holds factors, which the solves do not change).  Sixteen columns and
  // the early request since round 4's second half (eight, three barriers and two round trips per block before): the solves were 16 % of the time
  // of lcp_fast's slowest worlds, which pace a warm config-4 step.
  constexpr int SNB = (T >= 256) ? 16 : 8;
  static_assert(2 * SNB * SNB <= NB * (UCH + 8), "two staging areas in s_u");
  double* const su = &s_u[0][0];
  auto stage = [&](int kb, int nbk, int buf) {
    if (t < SNB * SNB) { const int r = t % SNB, c = t / SNB; if (r < nbk && c < nbk) su[buf * SNB * SNB + t] = A[(kb + r) + (size_t)k * (kb + c)]; }
  };
  { int buf = 0;
    stage(0, (k < SNB) ? k : SNB, 0);
    sync();
    for (int kb = 0; kb < k; kb += SNB, buf ^= 1) {            // unit lower
      const int nbk = (k - kb < SNB) ? k - kb : SNB;
      if (t < 64) {
        const int lane = t;
        double v = (lane < nbk) ? bb[kb + lane] : 0.0, Lr[SNB];
#pragma unroll
        for (int c = 0; c < SNB; c++) Lr[c] = (lane < nbk && c < nbk) ? su[buf * SNB * SNB + lane + SNB * c] : 0.0;
#pragma unroll
        for (int c = 0; c < SNB; c++) if (c < nbk) { const double bk = read_lane(v, c); if (lane > c) v = blk_upd(v, bk, Lr[c]); }
        if (lane < nbk) bb[kb + lane] = v;
      }
      sync();
      if (kb + SNB < k) stage(kb + SNB, (k - kb - SNB < SNB) ? k - kb - SNB : SNB, buf ^ 1);
      for (int i = kb + nbk + t; i < k; i += T) {
        double v = bb[i];
#pragma unroll
        for (int c = 0; c < SNB; c++) if (c < nbk) v = blk_upd(v, bb[kb + c], A[i + (size_t)k * (kb + c)]);
        bb[i] = v;
      }
      sync();
    } }
  { int buf = 0;
    { const int kb0 = (k - SNB > 0) ? k - SNB : 0; stage(kb0, k - kb0, 0); }
    sync();
    for (int ke = k; ke > 0; ke -= SNB, buf ^= 1) {            // upper, blocks [kb, ke) from the bottom
      const int kb = (ke - SNB > 0) ? ke - SNB : 0, nbk = ke - kb;
      if (t < 64) {
        const int lane = t;
        double v = (lane < nbk) ? bb[kb + lane] : 0.0, Ur[SNB];
#pragma unroll
        for (int c = 0; c < SNB; c++) Ur[c] = (lane < nbk && c < nbk) ? su[buf * SNB * SNB + lane + SNB * c] : 1.0;
#pragma unroll
        for (int c = SNB - 1; c >= 0; c--) if (c < nbk) { if (lane == c) v = v / Ur[c]; const double bk = read_lane(v, c); if (lane < c) v = blk_upd(v, bk, Ur[c]); }
        if (lane < nbk) bb[kb + lane] = v;
      }
      sync();
      if (kb > 0) { const int kbn = (kb - SNB > 0) ? kb - SNB : 0; stage(kbn, kb - kbn, buf ^ 1); }
      for (int i = t; i < kb; i += T) {
        double v = bb[i];
#pragma unroll
        for (int c = SNB - 1; c >= 0; c--) if (c < nbk) v = blk_upd(v, bb[kb + c], A[i + (size_t)k * (kb + c)]);
        bb[i] = v;
      }
      sync();
    } }
  if (b_lds) { for (int i = t; i < k; i += T) b[i] = s_b[i]; sync(); }
  bp_tock(BP_SOLVE, ts);
  return 0;
}

#include "mh_lu_compact.inc"
#include "mh_lu_left.inc"
#include "mh_lu_reg.inc"

// LCP.cpp:199-209 over the variables i with member(i) (list order = index order); val(i) reads
// the candidate.  Consumes exactly one rand().  Returns the chosen variable (uniform).
template <class Val, class Mem>
MH_DEV int rand_min(int n, Val val, Mem member, double tol, double& vsel, bool& tie) {
  const int t = tid();
  double best = inf(); int bi = 0x7fffffff;
  for (int i = t; i < n; i += T) if (member(i)) { const double v = val(i); if (v < best) { best = v; bi = i; } }
  double vmin; int imin; red_min_first(best, bi, vmin, imin);
  int c = 0;
  for (int i = t; i < n; i += T) if (member(i) && i != imin && val(i) < vmin + tol) c++;
  const int cnt = 1 + red_sum_int(c);
  tie = tie || cnt > 1;
  const int r = rand_next() % cnt;
  int chosen = imin;
  if (r != 0) {
    // the (r - 1)-th of the other candidates in index order: contiguous index ranges per thread, a block scan of their counts, and the one
    // thread whose range holds it walks its own few elements (one thread walking all n: 40 % of lcp_fast's time on box stacks, whose
    // symmetric corner contacts tie all the time)
    const int per = (n + T - 1) / T, lo = t * per, hi = (lo + per < n) ? lo + per : n;
    int c2 = 0;
    for (int i = lo; i < hi; i++) if (member(i) && i != imin && val(i) < vmin + tol) c2++;
    int tot;
    const int base = excl_scan_int(c2, tot);
    if (t == 0) s_bi[2] = imin;
    sync();
    if (r - 1 >= base && r - 1 < base + c2) { int seen = base; for (int i = lo; i < hi; i++) if (member(i) && i != imin && val(i) < vmin + tol) { if (seen == r - 1) { s_bi[2] = i; break; } seen++; } }
    sync(); chosen = s_bi[2]; sync();
  }
  vsel = val(chosen);
  return chosen;
}

// LCP.cpp:41-196.  z (n, in/out) and zsize as in lcp_fast_wave.
// Repeating pivot sequences.  One iteration of the loop below is a function of the index set alone (the matrices are rebuilt from it;
// the rand() VALUE matters only when rand_min finds several minima), so once the set at the top of an iteration equals the set h
// iterations earlier and none of those h iterations saw a tie, the remaining iterations repeat them for ever -- and this is how
// lcp_fast fails on the contact LCPs of resting stacks: LCP.cpp:176-187 takes the position found in the old _z as an index into the
// NEW _nonbas, moves the variable that has just entered straight out again, and the loop spins on one basis until MAX_PIV (period 1
// in all but a few percent of the failing calls, 2-4 in the rest; 85-95 % of the iterations of such a call).  What the reference
// leaves behind after r repetitions is known without running them: z untouched, pivots advanced, 1-2 rand() calls and 0-2 trace entries
// per iteration.  So the repetitions are skipped (whole periods only; the < h iterations left over are run).  FH_P: the longest period
// recognised; the sets of the last FH_P iterations are kept as bit words in LDS.
constexpr int FH_P = 8;
constexpr int FH_N = 2048, FH_W = FH_N / 32 + 2;    // larger problems run every iteration
__shared__ unsigned s_fh[FH_P][FH_W], s_fcur[FH_W];
__shared__ int s_fm[FH_P], s_frc[FH_P], s_ftc[FH_P], s_ftr[FH_P][2];
MH_DEV void rand_skip(unsigned m) {      // m rand() calls whose values nobody looks at
  if (tid() == 0) {
    unsigned idx = s_rng[31];
    for (unsigned i = 0; i < m; i++) { unsigned j = idx + 28; if (j >= 31) j -= 31; s_rng[idx] = s_rng[idx] + s_rng[j]; idx = (idx + 1 == 31) ? 0 : idx + 1; }
    s_rng[31] = idx;
  }
  sync();
}

MH_DEV bool lcp_fast(const Mat& M, double lam, const Ws& W, const double* q, double* z, int& zsize, double zero_tol,
                     double nrm_lam, unsigned& pivots, Trace2& tr, bool skip_repeats, bool reg_lu)
{
  const int n = M.n, t = tid();
  if (zero_tol < 0.0) zero_tol = (double)n * nrm_lam * MH_DBL_EPS;
  if (zsize == n) {
    for (int i = t; i < n; i += T) W.flag[i] = !(fabs(z[i]) < zero_tol) ? 1 : 0;
  } else {
    double best = inf(); int bi = 0x7fffffff;
    for (int i = t; i < n; i += T) { const double v = q[i]; if (v < best) { best = v; bi = i; } }
    double qmin; int minw; red_min_first(best, bi, qmin, minw);
    if (qmin > -zero_tol) { for (int i = t; i < n; i += T) z[i] = 0.0; sync(); zsize = n; pivots = 0; return true; }
    for (int i = t; i < n; i += T) W.flag[i] = (i == minw) ? 1 : 0;
  }
  sync();
  const unsigned MAX_PIV = 2u * (unsigned)n;
  const int nwords = (n + 31) >> 5;
  int it = 0;                      // iterations RUN (the ring of index sets is addressed by it, pivots counts the skipped ones too)
  int last_tie = -1, ncalls = 0, npush = 0;
  bool skipped = !skip_repeats || n > FH_N;
  for (pivots = 0; pivots < MAX_PIV; pivots++, it++) {
    if (!skipped) {
      if (it > 0 && t == 0) { s_frc[(it - 1) % FH_P] = ncalls; s_ftc[(it - 1) % FH_P] = npush; }
      ncalls = 0; npush = 0;
      for (int i0 = 0; i0 < n; i0 += T) {
        const int i = i0 + t;
        const unsigned long long m = __ballot(i < n && W.flag[i] != 0);
        if ((t & 63) == 0 && i < n) { s_fcur[i >> 5] = (unsigned)m; s_fcur[(i >> 5) + 1] = (unsigned)(m >> 32); }
      }
      if (t < FH_P) s_fm[t] = 1;
      sync();
      const int depth = (it < FH_P) ? it : FH_P;
      for (int e = t; e < FH_P * nwords; e += T) { const int h = e / nwords, wd = e - h * nwords; if (s_fh[h][wd] != s_fcur[wd]) s_fm[h] = 0; }
      sync();
      int lag = 0;
      for (int h = 1; h <= depth; h++) if (lag == 0 && s_fm[(it - h) % FH_P] != 0 && last_tie < it - h) lag = h;
      const unsigned reps = lag ? (MAX_PIV - pivots) / (unsigned)lag : 0u;
      if (reps > 0) {
        unsigned calls = 0, pushes = 0;
        for (int h = lag; h >= 1; h--) { calls += (unsigned)s_frc[(it - h) % FH_P]; pushes += (unsigned)s_ftc[(it - h) % FH_P]; }
        rand_skip(calls * reps);
        for (unsigned r = 0; r < reps; r++) {
          if (!tr.buf || tr.len >= tr.cap) { tr.len += (int)(pushes * (reps - r)); break; }
          for (int h = lag; h >= 1; h--) { const int sl = (it - h) % FH_P; for (int e = 0; e < s_ftc[sl]; e++) tr.push(s_ftr[sl][e]); }
        }
        pivots += reps * (unsigned)lag;
        skipped = true;
        if (pivots >= MAX_PIV) { pivots = MAX_PIV; break; }
      } else {
        for (int wd = t; wd < nwords; wd += T) s_fh[it % FH_P][wd] = s_fcur[wd];
      }
      sync();
    }
    auto push = [&](int32_t v) { tr.push(v); if (t == 0 && npush < 2) s_ftr[it % FH_P][npush] = v; npush++; };
    bool tie = false;
    unsigned long long tq = bp_tick();
    const int k = build_list(n, W);
    bp_tock(BP_LIST, tq); tq = bp_tick();
#ifdef MH_BLK_HAS_REGLU
    if (k > 0 && k <= RL_KMAX && reg_lu) {                  // the whole system in the registers of the sixteen waves (mh_lu_reg.inc)
      for (int i = t; i < k; i += T) s_list[i] = W.list[i];
      sync();
      bp_tock(BP_GATHER, tq);
      account_lu(k);
      tq = bp_tick();
      const int info = lu_gather_solve_reg(M, lam, q, k, W.A, W.b);
      bp_tock(BP_C_PANEL, tq);                              // (diagnostic builds: its cycles, calls and rows under the compact LU's names, unused by this kernel)
#ifdef MH_BLK_PROF
      if (t == 0) { s_prof[BP_C_STEPS] += 1ull; s_prof[BP_C_PANELS] += (unsigned long long)k; }
#endif
      if (info != 0) return false;
    } else
#endif
    if (k > 0) {
      { int r = t % k, c = t / k;                          // element e = t + m T, walked without a division per element
        const int dr = T % k, dc = T / k;
        const bool l_lds = k <= LIST_CAP;                   // the index list through LDS: one global load per element, not two dependent ones
        if (l_lds) { for (int i = t; i < k; i += T) s_list[i] = W.list[i]; sync(); }
        const int* L = l_lds ? s_list : W.list;
        for (long e = t; e < (long)k * k; e += T) {
          W.A[e] = M.at(L[r], L[c], lam);
          r += dr; c += dc; if (r >= k) { r -= k; c++; }
        } }
      for (int r = t; r < k; r += T) W.b[r] = -q[W.list[r]];
      sync();
      bp_tock(BP_GATHER, tq);
      account_lu(k);
      if (lu_solve(k, W.A, W.b) != 0) return false;
    }
    tq = bp_tick();
    // w = Mmix z + qbas on the basic variables (dgemv column order)
    // (z values and their variables are staged through LDS UCH at a time: the inner loop then has one
    //  independent global load per term; the running sums stay in the workspace between chunks)
    for (int cb = 0; cb < k || cb == 0; cb += UCH) {
      const int ncb = (k - cb < UCH) ? ((k - cb > 0) ? k - cb : 0) : UCH;
      for (int c = t; c < ncb; c += T) { s_u[0][c] = W.b[cb + c]; s_li[c] = W.list[cb + c]; }
      sync();
      for (int i = t; i < n; i += T) if (!W.flag[i]) {
        double w = (cb == 0) ? 0.0 : W.w[i];
        for (int c = 0; c < ncb; c++) w = w + s_u[0][c] * M.at(i, s_li[c], lam);
        W.w[i] = (cb + UCH >= k) ? w + q[i] : w;
      }
      sync();
    }
    bp_tock(BP_GEMV, tq); tq = bp_tick();
    auto wval = [&](int i) { return W.w[i]; };
    auto isb = [&](int i) { return W.flag[i] == 0; };
    auto zval = [&](int i) { return W.b[W.pos[i]]; };
    auto isnb = [&](int i) { return W.flag[i] != 0; };
    double wsel = 0.0; int minw = -1;
    if (k < n) minw = (ncalls++, rand_min(n, wval, isb, zero_tol, wsel, tie));
    if (minw < 0 || wsel > -zero_tol) {
      double zsel = 0.0; int minz = -1;
      if (k > 0) minz = (ncalls++, rand_min(n, zval, isnb, zero_tol, zsel, tie));
      bp_tock(BP_RANDMIN, tq);
      if (minz >= 0 && zsel < -zero_tol) {
        if (t == 0) W.flag[minz] = 0;
        push(-(int32_t)(minz + 1));
        sync();
      } else {
        for (int i = t; i < n; i += T) z[i] = W.flag[i] ? W.b[W.pos[i]] : 0.0;
        sync();
        zsize = n;
        return true;
      }
    } else {
      push((int32_t)(minw + 1));
      double zsel = 0.0; int minzv = -1;
      if (k > 0) minzv = (ncalls++, rand_min(n, zval, isnb, zero_tol, zsel, tie));
      bp_tock(BP_RANDMIN, tq);
      int idx2 = -1;
      if (minzv >= 0 && zsel < -zero_tol) {
        // LCP.cpp:176-187: the POSITION found in the old _z indexes the NEW, re-sorted _nonbas
        if (t == 0) {
          const int posz = W.pos[minzv];
          int lo = 0, hi = k;                               // insertion point of minw in the old list
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (W.list[mid] < minw) lo = mid + 1; else hi = mid; }
          s_bi[3] = (posz < lo) ? W.list[posz] : ((posz == lo) ? minw : W.list[posz - 1]);
        }
        sync(); idx2 = s_bi[3]; sync();
      }
      if (t == 0) { W.flag[minw] = 1; if (idx2 >= 0) W.flag[idx2] = 0; }
      if (idx2 >= 0) push(-(int32_t)(idx2 + 1));
      sync();
    }
    if (tie) last_tie = it;
  }
  return false;
}

// LCP.cpp:240-249 (strict = false) / :303-312 (strict = true, against M + lam I)
MH_DEV bool verify(const Mat& M, double lam, const Ws& W, const double* q, const double* z, double ZERO_TOL, bool strict)
{
  const int n = M.n, t = tid();
  const double nT = -ZERO_TOL;
  int bad = 0;
  for (int i = t; i < n; i += T) {
    const double zi = z[i];
    if (!(strict ? (zi > nT) : (zi >= nT))) bad = 1;
  }
  if (red_sum_int(bad) != 0) return false;
  int bad_w = 0, bad_zw = 0, bad_hi = 0;
  for (int i = t; i < n; i += T) {
    double w = 0.0;
    for (int c = 0; c < n; c++) { const double zc = z[c]; if (zc != 0.0) w = w + zc * M.at(i, c, lam); }
    w = w + q[i];
    if (!(strict ? (w > nT) : (w >= nT))) bad_w = 1;
    const double zw = z[i] * w;
    if (!(strict ? (zw > nT) : (zw >= nT))) bad_zw = 1;
    if (!(zw < ZERO_TOL)) bad_hi = 1;
  }
  if (red_sum_int(bad_w) != 0) return false;
  if (red_sum_int(bad_zw) != 0) return false;
  return red_sum_int(bad_hi) == 0;
}

// LCP.cpp:545-1003 (dense)
// Task mode of the lcp_lemke kinds: the attempts of lcp_lemke_regularized's ladder do not depend on one another (each starts from z = 0,
// LCP.cpp:564, and lcp_lemke never USES the rand() values it draws), so one workgroup can run ONE attempt of ONE problem and a
// selection pass afterwards takes the first attempt, in ladder order, that succeeded.  solved_at[w] = the lowest attempt known to have
// succeeded: attempts above it are not started, and a running one gives up when it learns of a lower success (its result could never
// be the one selected).  rung < 0: not in task mode.
__shared__ int s_nodraw;   // lcp_lemke left through its trivial exit (LCP.cpp:578), before the n rand() draws of :618-620
struct LadderTask { int* solved_at; int rung;
  MH_DEV bool pointless() const { return rung >= 0 && *(volatile int*)solved_at < rung; } };

// the basis of lcp_lemke by columns, for lu_compact: position p holds the slack -e_{id-n}, the artificial column, or column id of M
struct LemkeCol { const Mat* M; const int* bv; const double* art; double lam; int n, tt;
  MH_DEV int unit_row(int p) const { const int id = bv[p]; return (id >= n && id != tt) ? id - n : -1; }
  MH_DEV int id(int p) const { return bv[p]; }
  MH_DEV double load_id(int id, int i) const { return (id == tt) ? art[i] : M->at(i, id, lam); } };

MH_DEV bool lcp_lemke(const Mat& M, double lam, const Ws& W, const double* q, double* z, int& zsize, double piv_tol, double zero_tol,
                      double nrm_lam, unsigned& pivots, Trace2& tr, bool compact, bool reuse, const LadderTask& task)
{
  const int n = M.n, t = tid();
  const unsigned MAXITER = (50u * (unsigned)n < 1000u) ? 50u * (unsigned)n : 1000u;
  pivots = 0;
  for (int i = t; i < n; i += T) z[i] = 0.0;                        // z.set_zero() (:564)
  sync();
  const int z0size = zsize;
  if (zero_tol <= 0.0) zero_tol = MH_DBL_EPS * nrm_lam * (double)n;
  double best = inf(); int bi = 0x7fffffff;
  for (int i = t; i < n; i += T) { const double v = q[i]; if (v < best) { best = v; bi = i; } }
  double xmin; int lvindex; red_min_first(best, bi, xmin, lvindex);
  if (xmin > -zero_tol) { if (t == 0) s_nodraw = 1; zsize = n; return true; }   // (:578: returns BEFORE _restart_z0's draws -- a task reports it, see k_ladder_select)
  zsize = 2 * n;                                                    // z.set_zero(2n) (:596)
  const int tt = 2 * n;
  if (z0size != n) for (int i = 0; i < n; i++) (void)rand_next();   // _restart_z0 (:618-620)
  for (int p = t; p < n; p += T) { W.bv[p] = n + p; W.x[p] = q[p]; }
  sync();
  if (!(xmin < 0.0)) { zsize = n; return true; }                    // no negative entry (:737)
  const double PIV_TOL = (piv_tol > 0.0) ? piv_tol : MH_DBL_EPS * (double)n * ((nrm_lam > 1.0) ? nrm_lam : 1.0);
  const double tval = -xmin;
  int leaving = n + lvindex;
  int entering = tt;
  for (int p = t; p < n; p += T) {
    const double xp = W.x[p];
    double u = (xp < 0.0) ? 1.0 : 0.0;
    W.art[p] = u;
    u = u * tval;
    W.x[p] = xp + u;
  }
  sync();
  if (t == 0) { W.x[lvindex] = tval; W.bv[lvindex] = tt; }
  sync();
  int prev_nd = -1;                                                 // lu_compact's records of the last basis (it differs from the next one at lvindex only)
  for (pivots = 0; pivots < MAXITER; pivots++) {
    if (task.rung >= 0 && (pivots & 15u) == 15u && bcast_i(task.pointless() ? 1 : 0)) { zsize = n; return false; }   // (a lower attempt has succeeded)
    if (leaving == tt) {
      for (int p = t; p < n; p += T) { const int id = W.bv[p]; if (id < n) z[id] = W.x[p]; }   // (:804-806)
      sync();
      zsize = n;
      return true;
    }
    if (leaving < n) { entering = n + leaving; for (int p = t; p < n; p += T) W.d[p] = (p == leaving) ? -1.0 : 0.0; }
    else { entering = leaving - n; for (int p = t; p < n; p += T) W.d[p] = M.at(p, entering, lam); }
    // solve Bl d = Be (LCP.cpp:837-838).  The structure-exploiting routine first; the dense one on the assembled basis when the
    // problem is too large for it or it met a non-finite value
    sync();
    account_lu(n);
    int info = LUC_FALLBACK;
    if (compact && n <= CN) {
      const unsigned long long tc = bp_tick();
      LemkeCol colv; colv.M = &M; colv.bv = W.bv; colv.art = W.art; colv.lam = lam; colv.n = n; colv.tt = tt;
      if (!reuse) prev_nd = -1;
#ifdef MH_BLK_RIGHT_LOOKING      /* the round-3 schedule (mh_lu_compact.inc), kept for A/B builds */
      info = lu_compact(n, colv, W.A, W.d, W.w, W.list, prev_nd, lvindex);   // W.w: unused by lcp_lemke, the trash column of the update lists; W.list .. W.pos: 3 n ints, idle here
#else
      info = lu_left(n, colv, W.A, W.d, W.list, prev_nd, lvindex);           // W.list .. W.pos: 3 n ints, idle here (the step records between calls)
#endif
      bp_tock(BP_COMPACT, tc);
    }
    if (info == LUC_FALLBACK) {
    const unsigned long long tq = bp_tick();
    const long nn = (long)n * n;
    { int r = t % n, p = t / n;                             // element e = t + m T, walked without a division per element
      const int dr = T % n, dp = T / n;
      for (long e = t; e < nn; e += T) {
        const int id = W.bv[p];
        double a;
        if (id == tt) a = W.art[r];
        else if (id >= n) a = (r == id - n) ? -1.0 : 0.0;
        else a = M.at(r, id, lam);
        W.A[e] = a;
        r += dr; p += dp; if (r >= n) { r -= n; p++; }
      } }
    sync();
    bp_tock(BP_GATHER, tq);
    prev_nd = -1;                                                   // (the dense route overwrites the workspace)
    info = lu_solve(n, W.A, W.d);
    }
    if (info != 0) return false;                                    // singular basis (:840-850), size stays 2n
    double th = inf(); int any = 0, cand0 = 0x7fffffff;
    for (int p = t; p < n; p += T) { const double dp = W.d[p]; if (dp > PIV_TOL) { any = 1; if (p < cand0) cand0 = p; const double r = (W.x[p] + zero_tol) / dp; th = (r < th) ? r : th; } }
    if (red_sum_int(any) == 0) return false;                        // ray termination (:892-903)
    // theta = *std::min_element(ratios) (LCP.cpp:920): NaN ratios are skipped by the scan unless the FIRST candidate's is one --
    // it then stays the "minimum" and the set below comes out empty (:946-958)
    { double dm; int c0; red_min_first(0.0, cand0, dm, c0); const double r0 = (W.x[c0] + zero_tol) / W.d[c0]; if (r0 != r0) { zsize = n; return false; } }
    const double theta = red_min(th);
    int first_keep = 0x7fffffff, tkeep = 0x7fffffff;
    for (int p = t; p < n; p += T) {
      const double dp = W.d[p];
      if (dp > PIV_TOL && (W.x[p] / dp <= theta)) { if (p < first_keep) first_keep = p; if (W.bv[p] == tt && p < tkeep) tkeep = p; }
    }
    double dummy; int fk, tk;
    red_min_first(0.0, first_keep, dummy, fk);
    red_min_first(0.0, tkeep, dummy, tk);
    if (fk == 0x7fffffff) { zsize = n; return false; }              // (:946-958)
    lvindex = (tk != 0x7fffffff) ? tk : fk;
    leaving = W.bv[lvindex];
    const double ratio = W.x[lvindex] / W.d[lvindex];
    sync();
    for (int p = t; p < n; p += T) { const double dp = W.d[p] * ratio; W.x[p] = W.x[p] - dp; }
    sync();
    if (t == 0) { W.x[lvindex] = ratio; W.bv[lvindex] = entering; }
    sync();
    tr.push((int32_t)entering + 1); tr.push((int32_t)leaving + 1);
  }
  zsize = n;
  return false;
}

// the four public solvers (lcp_solve_wave's attempt loop)
template <int FAM>      // 0: the lcp_fast kinds, 1: the lcp_lemke kinds -- one kernel each, so that neither carries the other's registers
MH_DEV bool lcp_solve(const LcpParams& P, const Pow10Table& p10, const Mat& M, const Ws& W, const double* q, double* z, int& zsize,
                      unsigned& pivots, Trace2& tr, bool compact, bool skip_repeats, bool reuse, bool reg_lu, int att_first, int att_count, const LadderTask& task)
{
  const int n = M.n, t = tid();
  const bool reg = (P.kind == MH_LCP_FAST_REG) || (P.kind == MH_LCP_LEMKE_REG);
  double m0 = 0.0;
  const long nn = (long)n * n;
  for (long e = t; e < nn; e += T) { const double a = fabs(M.M[(e % n) + (size_t)M.ld * (e / n)]); m0 = (a > m0) ? a : m0; }
  const double nrm0 = red_max(m0);
  const double ZERO_TOL = (P.zero_tol > 0.0) ? P.zero_tol : (double)n * nrm0 * MH_NEAR_ZERO;
  unsigned total = 0;
  double offmax = 0.0;
  bool have_off = false;
  // attempt a >= 1 of the ladder regularises with 10^rf, rf = min_exp + (a - 1) step_exp (LCP.cpp:252-262, 404-414); a launch may run a
  // window [att_first, att_first + att_count) of it -- the whole ladder by default, ONE attempt per workgroup in task mode
  for (int attempt = att_first; attempt - att_first < att_count; attempt++) {
    double lam = 0.0, nrm = nrm0;
    const int rf = P.min_exp + (attempt - 1) * (int)P.step_exp;
    if (attempt > 0) {
      if (!reg || !(rf < P.max_exp)) break;
      if (!have_off) {
        have_off = true;
        double mo = 0.0;
        for (long e = t; e < nn; e += T) { const int r = (int)(e % n), c = (int)(e / n); if (r != c) { const double a = fabs(M.M[r + (size_t)M.ld * c]); mo = (a > mo) ? a : mo; } }
        offmax = red_max(mo);
      }
      lam = p10.v[rf + 32];
      double md = 0.0;
      for (int i = t; i < n; i += T) { const double a = fabs(M.M[i + (size_t)M.ld * i] + lam); md = (a > md) ? a : md; }
      md = red_max(md);
      nrm = (md > offmax) ? md : offmax;
    }
    if (reg) tr.push(0x40000000 | attempt);
    bool ok;
    if constexpr (FAM == 0) ok = lcp_fast(M, lam, W, q, z, zsize, P.zero_tol, nrm, pivots, tr, skip_repeats, reg_lu);
    else ok = lcp_lemke(M, lam, W, q, z, zsize, P.piv_tol, P.zero_tol, nrm, pivots, tr, compact, reuse, task);
    if (!reg) return ok;
    const bool good = ok && verify(M, lam, W, q, z, ZERO_TOL, attempt > 0);
    if (attempt == 0) { if (good) return true; total += pivots; }
    else { total += pivots; if (good) { pivots = total; return true; } }
  }
  pivots = total;
  return false;
}

// The ladder's tasks handed out by need.  Per problem w (3 ints at solved_at[w], [Bw + w], [2 Bw + w]): the lowest attempt known to
// have succeeded, the next attempt to hand out, the number of attempts over.  While at least as many problems are unsolved as there
// are workgroups, a free workgroup takes the next attempt of the problem with the FEWEST attempts running (ties: the lowest index):
// every ladder then runs in sequence and nothing is started above an attempt that will succeed (by block index 11-18 % of the pivots
// of a full batch were spent there).  With fewer problems than workgroups it takes the LOWEST attempt not yet handed out (ties: the
// lowest index) -- the attempt-major order of the block-index launch, which spends the spare workgroups on the attempts most likely
// to be the one selected (fewest-running first ran 19 % more pivots there).
// Returns the task index attempt * Bw + w, or -1 when nothing is left to hand out (block-uniform).
// verdict (or NULL): per problem 0 = lcp_fast is still at work on it, 1 = it failed (the ladder is needed), 2 = it solved the problem -- written by
// lcp_fast's kernel, which runs beside this one (core_solve_round, mh_impact.hip).  Only problems with verdict 1 are handed out.  A workgroup that
// finds nothing while verdicts are outstanding looks again a few times, a millisecond apart, and then LEAVES: the problems lcp_fast decides later are
// taken by the second launch of this kernel, after lcp_fast's (the state of the hand-out -- next attempt, attempts over, solved_at -- carries over).
// Waiting for the last verdict instead was tried and is not safe: with several hundred workgroups resident and waiting, lcp_fast's remaining workgroups
// -- on other CUs, at the full shader clock -- took 40-70 s for what takes them 3 (profiles/r04_d_waiting_workgroups_stall.txt; not the polling: one
// thread per workgroup looked at one word once a millisecond).
// order (or NULL): a permutation of the problems, the ones expected to take longest first (core_solve_round ranks the worlds by the solver time they have
// used so far): ties between candidates go to the earlier place in it instead of the lower index -- longest processing time first, the classic cure for a tail.
MH_DEV int pick_task(int Bw, int R, int* st, const int* __restrict__ run_if, const int* __restrict__ n_arr, const int* verdict, const int* __restrict__ order)
{
  const int t = tid();
  volatile int* solved = st; volatile int* next = st + Bw; volatile int* done = st + 2 * Bw;
  int looks = 0;
  for (;;) {                                       // (a lost compare-and-swap means another workgroup took a task: the whole makes progress)
    // verdict[-1]: how many verdicts lcp_fast has published -- read BEFORE the scan, so that one published during it is not slept through.
    // (RELAXED loads: an agent-scope ACQUIRE invalidates the XCD's L2; nothing read after the wait needs ordering -- the verdicts are read with
    //  atomic loads too, and the tasks only read M and q, written before either kernel started)
    const int published = (verdict && t == 0) ? __hip_atomic_load(verdict - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    double best = inf(), lowest = inf(); int bw = 0x7fffffff, lw = 0x7fffffff, open_ = 0, pend = 0;
    for (int i = t; i < Bw; i += T) {
      const int w = order ? order[i] : i;
      if (run_if && run_if[w] == 0) continue;
      if (n_arr && n_arr[w] <= MH_LCP_MAX_N_WAVE) continue;
      if (verdict) {
        const int v = __hip_atomic_load(verdict + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0) { pend++; continue; }
        if (v == 2) continue;
      }
      const int nr = next[w];
      if (nr >= R || solved[w] < nr) continue;
      open_++;
      const double d = (double)(nr - done[w]);
      if (d < best) { best = d; bw = i; }
      if ((double)nr < lowest) { lowest = (double)nr; lw = i; }
    }
    double dmin; int w, w2; red_min_first(best, bw, dmin, w); red_min_first(lowest, lw, dmin, w2);
    if (order) { if (w != 0x7fffffff) w = order[w]; if (w2 != 0x7fffffff) w2 = order[w2]; }       // (places in the order back to problems)
    if (w == 0x7fffffff) {
      if (verdict == nullptr || red_sum_int(pend) == 0 || looks >= 8) return -1;
      looks++;
      if (t == 0 && __hip_atomic_load(verdict - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == published)
        for (int i = 0; i < 300; i++) __builtin_amdgcn_s_sleep(127);          // about a millisecond
      sync();
      continue;
    }
    if (red_sum_int(open_) < (int)gridDim.x) w = w2;
    int got = -2;
    if (t == 0) {
      const int nr = next[w];
      if (nr < R && !(solved[w] < nr) && atomicCAS(st + Bw + w, nr, nr + 1) == nr) got = nr * Bw + w;
    }
    got = bcast_i(got);
    if (got >= 0) return got;
  }
}

// n > 64: one T-thread workgroup per LCP, M read in place from HBM, everything else in a per-problem HBM workspace.
template <int FAM>
__global__ __launch_bounds__(T) MH_BLK_KATTR
void k_lcp_block(int B, int n, const double* __restrict__ Mg, int ld, long strideM,
                    const double* __restrict__ qg, double* __restrict__ zg,
                    const int* __restrict__ zsz_in, int* __restrict__ zsz_out,
                    uint32_t* __restrict__ rngg, int* __restrict__ status, unsigned* __restrict__ pivots_out,
                    int32_t* __restrict__ trace, int trace_cap, int* __restrict__ trace_len,
                    LcpParams P, Pow10Table p10, double* __restrict__ wsd, int* __restrict__ wsi,
                    const int* __restrict__ run_if, const int* __restrict__ n_arr, int flags, double* __restrict__ work,
                    int task_worlds, int* __restrict__ solved_at)
{
  // b: the index of everything this workgroup OWNS (z, sizes, rand() state, status, pivots, workspace, work counters); bw: the problem it
  // reads (M, q, its size, the mask).  They differ in task mode only (task_worlds > 0: task b = attempt b / task_worlds of problem
  // b % task_worlds).  Two ways of handing tasks out: by block index, attempt-major, so that the lower attempts of every problem are
  // dispatched first (one task per workgroup); or, flags & 8, by pick_task below (a workgroup takes tasks until none is left).
  const int t = tid();
  // the lcp_fast kinds have no tasks: solved_at, when given, counts the workgroups that have STARTED (the gate in front of the ladder's
  // tasks on the second stream waits for the last one: core_solve_round, mh_impact.hip)
  if (FAM == 0 && task_worlds == 0 && solved_at != nullptr && t == 0) atomicAdd(solved_at, 1);
  const int n_launch = n, ld_launch = ld;
  const bool queue = (FAM == 1) && task_worlds > 0 && (flags & 8) != 0;     // (the lcp_fast kinds have no tasks: their kernel keeps the single pass)
  // flags & 64: tasks in BLOCK-INDEX order (attempt-major) from a counter -- the order the hardware would dispatch one-task workgroups in -- by as many workgroups as
  // the chip holds at once.  Either way a workgroup of a task launch is PERSISTENT and owns ONE LU workspace (by blockIdx.x) for all the tasks it runs: the workspaces
  // number the resident workgroups, not the tasks (round 4 allocated one per (world, attempt): 48 GB for 16-box stacks x 1024 worlds, and batches whose tasks did not
  // fit the device ran their ladders in sequence).  The counter is the first word of the hand-out's `next` array, which this order does not use.
  const bool cqueue = (FAM == 1) && task_worlds > 0 && !queue && (flags & 64) != 0;
  for (int round = 0; FAM == 1 || round < 1; round++) {
  int b;
  if (queue) { b = pick_task(task_worlds, B / task_worlds, solved_at, run_if, n_arr, (flags & 32) ? solved_at + 3 * (size_t)task_worlds + 2 : nullptr,
                             (flags & 128) ? solved_at + 4 * (size_t)task_worlds + 2 : nullptr); if (b < 0) return; }
  else if (cqueue) { b = bcast_i((t == 0) ? atomicAdd(solved_at + task_worlds, 1) : 0); if (b >= B) return; }
  else { if (round > 0) return; b = blockIdx.x; if (b >= B) return;
         // the lcp_fast kinds on a full chip (flags & 128): workgroup i takes the problem at place i of the order (behind the gate's counter and the B verdicts), so
         // the worlds expected to run longest START first -- the kernel lasts as long as its slowest world plus the time that world waited for a CU
         if (FAM == 0 && task_worlds == 0 && solved_at != nullptr && (flags & 128)) b = solved_at[B + 2 + b]; }
  n = n_launch; ld = ld_launch;
  const int bw = (task_worlds > 0) ? b % task_worlds : b;
  LadderTask task; task.solved_at = (task_worlds > 0) ? solved_at + bw : nullptr; task.rung = (task_worlds > 0) ? b / task_worlds : -1;
  if (run_if && run_if[bw] == 0) { if (task_worlds > 0 && t == 0) status[b] = -1; continue; }
  // (block-uniform: thread 0 reads solved_at once and broadcasts -- another workgroup's atomicMin may land between the loads of two waves)
  if (task_worlds > 0 && bcast_i(task.pointless() ? 1 : 0)) { if (t == 0) status[b] = -1; if (queue && t == 0) atomicAdd(solved_at + 2 * task_worlds + bw, 1); continue; }     // -1: not run
  Ws W;
  const size_t wslot = (queue || cqueue) ? (size_t)blockIdx.x : (size_t)b;      // (persistent workgroups: one workspace each)
  double* wd = wsd + wslot * ws_doubles(n);
  int* wi = wsi + wslot * ws_ints(n);
  // per-problem sizes: strides of q / z / M / the workspace stay those of the largest problem (the launch's n), M is
  // compact (ld = its own n); problems of at most 64 rows belong to the wave solver of the same call
  const int nstride = n;
  if (n_arr) { n = n_arr[bw]; ld = n; if (n <= MH_LCP_MAX_N_WAVE) { if (task_worlds > 0 && t == 0) status[b] = -1; continue; } }
  W.A = wd; W.b = wd + (size_t)n * n; W.w = W.b + n; W.x = W.w + n; W.d = W.x + n; W.art = W.d + n;
  W.list = wi; W.flag = wi + n; W.pos = wi + 2 * (size_t)n; W.bv = wi + 3 * (size_t)n;
  if (t < 32) s_rng[t] = rngg[(size_t)b * MH_RAND_WORDS + t];
  if (t == 0) { s_luc_bug = 0; s_work[0] = 0.0; s_work[1] = 0.0; s_work[2] = 0.0; s_work[3] = 0.0; s_nodraw = 0; }
  const unsigned long long t_task = wall_clock64();
#ifdef MH_GATE_DIAG
  const unsigned long long c_task = __builtin_amdgcn_s_memtime();
#endif
  Mat M; M.M = Mg + (size_t)bw * strideM; M.ld = ld; M.n = n;
  const double* q = qg + (size_t)bw * nstride;
  double* z = zg + (size_t)b * nstride;
  int zsize = zsz_in ? zsz_in[b] : n;
  if (zsize != n) for (int i = t; i < n; i += T) z[i] = 0.0;
  sync();
  Trace2 tr; tr.buf = trace ? trace + (size_t)b * trace_cap : nullptr; tr.cap = trace_cap; tr.len = 0;
  unsigned piv = 0;
#ifdef MH_BLK_PROF
  if (t < BP_COUNT) s_prof[t] = 0ull;
  sync();
  const unsigned long long t_kernel = bp_tick();
#endif
  const bool ok = lcp_solve<FAM>(P, p10, M, W, q, z, zsize, piv, tr, (flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0, (flags & 16) != 0, (task.rung >= 0) ? task.rung : 0, (task.rung >= 0) ? 1 : 0x3fffffff, task);
  if (task.rung >= 0 && ok && t == 0) atomicMin(task.solved_at, task.rung);
  sync();
#ifdef MH_BLK_PROF
  if (t == 0 && FAM == 0) printf("fastblk %d %u %.3f %llu %llu\n", b, piv, (double)(wall_clock64() - t_task) * 1e-5, s_prof[BP_C_STEPS], s_prof[BP_C_PANELS]);
  if (t == 0 && (b == 0 || b == 5)) printf("blk prof (cycles, block %d, %u pivots, %.3f ms wall, %llu ticks in all): list %llu gather %llu panel %llu swap %llu trail %llu solve %llu gemv %llu randmin %llu compact %llu [setup %llu panel %llu u12 %llu trail %llu back %llu; dense steps %llu panels %llu]\n", b, piv, (double)(wall_clock64() - t_task) * 1e-5, bp_tick() - t_kernel,
                               s_prof[0], s_prof[1], s_prof[2], s_prof[3], s_prof[4], s_prof[5], s_prof[6], s_prof[7], s_prof[8], s_prof[9], s_prof[10], s_prof[11], s_prof[12], s_prof[13], s_prof[14], s_prof[15]);
#endif
  if (t < 32) rngg[(size_t)b * MH_RAND_WORDS + t] = s_rng[t];
  if (t == 0 && work) { double* wk = work + MH_WORK * (size_t)b; wk[0] += s_work[0]; wk[1] += s_work[1]; wk[2] += s_work[2]; wk[3] += (double)(wall_clock64() - t_task); }
  if (t == 0) {
    if (s_luc_bug) printf("mh_lcp_block: index invariant %d of the compact LU violated (problem %d, n %d)\n", s_luc_bug & 15, b, s_luc_bug >> 4);
    status[b] = s_luc_bug ? -7 : (ok ? 1 : 0);
    if (pivots_out) pivots_out[b] = piv;
    if (zsz_out) zsz_out[b] = zsize | ((task.rung >= 0 && s_nodraw) ? MH_TASK_NODRAW : 0);   // (task mode only: bit 30 = this attempt would not have drawn)
    if (trace_len) trace_len[b] = tr.len;
    if (queue) atomicAdd(solved_at + 2 * task_worlds + bw, 1);          // one more attempt of this problem is over
    if (FAM == 0 && task_worlds == 0 && solved_at != nullptr && (flags & 32)) {    // the verdict the ladder's tasks on the second stream wait for (pick_task)
      __threadfence();
      __hip_atomic_store(solved_at + 2 + b, ok ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(solved_at + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // (solved_at[0]: workgroups started, [1]: verdicts published, [2 + b]: the verdicts)
#ifdef MH_GATE_DIAG
      if (wall_clock64() - t_task > 600000000ull) printf("lcp_fast: world %d started at clock %llu, took %.1f s, %u pivots, shader clock %.0f MHz on average\n", b, t_task, (double)(wall_clock64() - t_task) * 1e-8, piv, (double)(__builtin_amdgcn_s_memtime() - c_task) / ((double)(wall_clock64() - t_task) * 1e-2));
#endif
    }
  }
  sync();
  }   // (the next task of this workgroup)
}

} } // namespace mh::MH_BLK_NS

// host side: the launcher of this geometry (declared in mh_host.h)
extern "C" MH_HIDDEN hipError_t MH_BLK_LAUNCHER(void* stream, int kind, int B, int n, const double* M, int ld, long strideM, const double* q, double* z,
                                     const int* zsz_in, int* zsz_out, uint32_t* rng, int* status, unsigned* pivots,
                                     int32_t* trace, int trace_cap, int* trace_len, const mh::LcpParams* P, const mh::Pow10Table* p10,
                                     double* wsd, int* wsi, const int* run_if, const int* n_arr, int flags, double* work, int task_worlds, int* solved_at)
{
  namespace ns = mh::MH_BLK_NS;
#ifdef MH_BLK_LEMKE_ONLY      /* a geometry that exists for the lcp_lemke kinds only (mh_lcp_blkx.hip) */
  if (kind == MH_LCP_FAST || kind == MH_LCP_FAST_REG) return hipErrorInvalidValue;
#else
  if (kind == MH_LCP_FAST || kind == MH_LCP_FAST_REG)
    hipLaunchKernelGGL(ns::k_lcp_block<0>, dim3(B), dim3(ns::T), 0, (hipStream_t)stream, B, n, M, ld, strideM, q, z, zsz_in, zsz_out, rng, status, pivots,
                       trace, trace_cap, trace_len, *P, *p10, wsd, wsi, run_if, n_arr, flags, work, task_worlds, solved_at);
  else
#endif
  {
    int grid = B;
    if (task_worlds > 0 && (flags & (8 | 64))) {                        // as many workgroups as the chip holds at once; each takes tasks until none is left
      static int per_cu = 0;
      if (per_cu == 0) { int v = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, ns::k_lcp_block<1>, ns::T, 0) != hipSuccess || v < 1) v = 1; per_cu = v; }
      long cap = (long)per_cu * mh_cu_count(); if (cap > mh_task_slots(n)) cap = mh_task_slots(n);        // (the caller sized the workspaces by mh_task_slots)
      grid = (B < cap) ? B : (int)cap;
    }
    hipLaunchKernelGGL(ns::k_lcp_block<1>, dim3(grid), dim3(ns::T), 0, (hipStream_t)stream, B, n, M, ld, strideM, q, z, zsz_in, zsz_out, rng, status, pivots,
                       trace, trace_cap, trace_len, *P, *p10, wsd, wsi, run_if, n_arr, flags, work, task_worlds, solved_at);
  }
  return hipGetLastError();
}
