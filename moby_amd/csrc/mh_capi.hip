// libmoby_hip.so: kernels + the C ABI declared in include/moby_hip.h.
// gfx950 only.  Build: see __graft_entry__.build() / moby_amd/csrc/Makefile.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>
#include "../../include/moby_hip.h"
#include "mh_lcp_wave.h"
// the workgroup-per-problem solver in two thread geometries: 256 threads (two problems per CU: throughput when the
// batch is larger than the chip) and 1024 threads (one problem per CU with 16 waves to hide its round trips:
// 1.2x / 1.4x faster per problem at n = 256 / 512, slower at n = 128)
// The 256-thread geometry is the throughput one (batches larger than the chip): half the LDS staging (panel 14 KB,
// pivot-row chunk 128 columns) and a 128-VGPR budget let FOUR problems share a CU instead of two -- 4-box stacks x8192:
// 8.4 s -> 5.6 s per cold call; a single problem is 3 % slower.
#define MH_BLK_NS blk
#define MH_BLK_T 256
#define MH_BLK_UCH 128
#define MH_BLK_PANEL_CAP 1792
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#include "mh_lcp_block.h"
#undef MH_BLK_NS
#undef MH_BLK_T
#undef MH_BLK_UCH
#undef MH_BLK_PANEL_CAP
#undef MH_BLK_KATTR
#define MH_BLK_NS blkw
#define MH_BLK_T 1024
#define MH_BLK_UCH 256
#define MH_BLK_PANEL_CAP 3584
#define MH_BLK_KATTR
#include "mh_lcp_block.h"
#undef MH_BLK_NS
#undef MH_BLK_T
#undef MH_BLK_UCH
#undef MH_BLK_PANEL_CAP
#undef MH_BLK_KATTR

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define MH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return fail(MH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

mh::Pow10Table make_pow10()
{
  mh::Pow10Table t;
  for (int i = 0; i < 64; i++) t.v[i] = std::pow(10.0, (double)(i - 32)); // LCP.cpp:285
  return t;
}

} // namespace

// ---------------------------------------------------------------------------
// One wavefront (= one 64-thread workgroup) per LCP.  M is streamed once from
// HBM into LDS (the only HBM read of size n^2), the LU scratch sits beside it.
// LDS per workgroup: (2 n^2 + n) * 8 bytes (n = 42: 28.6 KB -> 5 worlds per CU).
__global__ __launch_bounds__(64)
void mh_k_lcp_wave(int B, int n, const double* __restrict__ Mg, int ld, long strideM,
                   const double* __restrict__ qg, double* __restrict__ zg,
                   const int* __restrict__ zsz_in, int* __restrict__ zsz_out,
                   uint32_t* __restrict__ rngg, int* __restrict__ status, unsigned* __restrict__ pivots_out,
                   int32_t* __restrict__ trace, int trace_cap, int* __restrict__ trace_len,
                   mh::LcpParams P, mh::Pow10Table p10, const int* __restrict__ run_if)
{
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  if (b >= B) return;
  if (run_if && run_if[b] == 0) return;   // masked problem: every output of it is left untouched
  const int lane = mh::lane_id();
  double* Ms = lds;
  double* A = Ms + n * n;
  double* art = A + n * n;
  const double* Mb = Mg + (size_t)b * strideM;
  // stream M into LDS (the one HBM read of n^2 doubles), 8 x 16 B loads in
  // flight per lane, tracking norm_inf(M) = max |m| on the way
  double nrm0 = 0.0;
  const int nn = n * n;
  if (ld == n && ((((size_t)Mb) & 15) == 0) && (nn & 1) == 0) {
    const double2* src = reinterpret_cast<const double2*>(Mb);
    double2* dst = reinterpret_cast<double2*>(Ms);
    const int n2 = nn >> 1;
    for (int e0 = 0; e0 < n2; e0 += 8 * 64) {
      double2 v[8];
#pragma unroll
      for (int t = 0; t < 8; t++) { const int e = e0 + t * 64 + lane; v[t] = (e < n2) ? src[e] : make_double2(0.0, 0.0); }
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int e = e0 + t * 64 + lane;
        if (e < n2) dst[e] = v[t];
        const double a0 = fabs(v[t].x), a1 = fabs(v[t].y);
        nrm0 = (a0 > nrm0) ? a0 : nrm0; nrm0 = (a1 > nrm0) ? a1 : nrm0;
      }
    }
  } else {
    for (int c = 0; c < n; c++) {
      if (lane < n) {
        const double v = Mb[lane + (size_t)ld * c];
        Ms[lane + n * c] = v;
        const double a = fabs(v);
        nrm0 = (a > nrm0) ? a : nrm0;
      }
    }
  }
  nrm0 = mh::wave_max(nrm0);
  mh::wave_sync();
  const bool valid = lane < n;
  const double dii = valid ? Ms[lane + n * lane] : 0.0;
  const double qi = valid ? qg[(size_t)b * n + lane] : 0.0;
  int zsize = zsz_in ? mh::uni(zsz_in[b]) : n;
  double zi = (valid && zsize == n) ? zg[(size_t)b * n + lane] : 0.0;
  if (lane == 0) mh::g_lcp_prof_on = 0;
  mh::wave_sync();
  mh::WaveRand rng; rng.load(rngg + (size_t)b * MH_RAND_WORDS);
  mh::Trace tr; tr.buf = trace ? trace + (size_t)b * trace_cap : nullptr; tr.cap = trace_cap; tr.len = 0;
  unsigned piv = 0;
  mh::DenseLds Md; Md.M = Ms; Md.n = n;
  mh::LuScratch S; S.small = A; S.ka = n; S.big = A;
  const bool ok = mh::lcp_solve_wave(P, p10, n, Md, S, art, nrm0, dii, qi, zi, zsize, rng, piv, tr);
  if (valid) zg[(size_t)b * n + lane] = zi;
  rng.store(rngg + (size_t)b * MH_RAND_WORDS);
  if (lane == 0) {
    status[b] = ok ? 1 : 0;
    if (pivots_out) pivots_out[b] = piv;
    if (zsz_out) zsz_out[b] = zsize;
    if (trace_len) trace_len[b] = tr.len;
  }
}

// ---------------------------------------------------------------------------
extern "C" {

int mh_version(void) { return 100; }
const char* mh_last_error(void) { return g_err; }

int mh_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void mh_rand_seed(uint32_t* st, uint32_t seed)
{
  // glibc srandom_r, TYPE_3: r[i] = 16807*r[i-1] mod (2^31-1), then 310 draws
  // are discarded.  Ring layout: word i of the sequence lives in slot i % 31.
  int32_t r[31];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) {
    int64_t hi = r[i-1] / 127773, lo = r[i-1] % 127773;
    int64_t w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    r[i] = (int32_t)w;
  }
  for (int i = 0; i < 31; i++) st[i] = (uint32_t)r[i];
  uint32_t idx = 3; // words 31..33 repeat 0..2; word 34 lands in slot 3
  for (int i = 34; i < 344; i++) {
    st[idx] = st[idx] + st[(idx + 28) % 31];
    idx = (idx + 1) % 31;
  }
  st[31] = idx;
}

int mh_rand_next(uint32_t* st)
{
  uint32_t idx = st[31];
  uint32_t v = st[idx] + st[(idx + 28) % 31];
  st[idx] = v;
  st[31] = (idx + 1) % 31;
  return (int)(v >> 1);
}

static int g_debug_blk = 0;      // mh_debug_set(2, v): 0 = choose, 1 = 256-thread block solver, 2 = 1024-thread block solver
static int g_cu_count()
{
  static int cus = 0;
  if (cus == 0) { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount; if (cus <= 0) cus = 256; }
  return cus;
}

static int lcp_params(int kind, const mh_lcp_opts* o, mh::LcpParams& P)
{
  if (kind < MH_LCP_FAST || kind > MH_LCP_LEMKE_REG) return fail(MH_ERR_INVALID_ARG, "unknown LCP kind %d", kind);
  P.kind = kind;
  // defaults of include/Moby/LCP.h:21,26
  P.min_exp = -20; P.step_exp = (kind == MH_LCP_FAST_REG) ? 4u : 1u; P.max_exp = (kind == MH_LCP_FAST_REG) ? 20 : 1;
  P.piv_tol = -1.0; P.zero_tol = -1.0;
  if (o) { P.min_exp = o->min_exp; P.step_exp = o->step_exp; P.max_exp = o->max_exp; P.piv_tol = o->piv_tol; P.zero_tol = o->zero_tol; }
  if ((kind == MH_LCP_FAST_REG || kind == MH_LCP_LEMKE_REG)) {
    if (P.step_exp == 0) return fail(MH_ERR_INVALID_ARG, "step_exp must be > 0");
    if (P.min_exp < -32 || P.max_exp > 32) return fail(MH_ERR_INVALID_ARG, "regularisation exponents must lie in [-32, 32]");
  }
  return MH_OK;
}

// the LCP entry with a per-problem mask (run_if[b] == 0: problem b is skipped, outputs untouched) and an optional
// caller-owned block-solver workspace; the exported entry is the unmasked case
static int lcp_solve_dev_masked(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts, const int* run_if, double* ws_d, int* ws_i);

int mh_lcp_solve_batch_dev(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts)
{
  return lcp_solve_dev_masked(stream, kind, B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                              trace, trace_cap, trace_len, opts, nullptr, nullptr, nullptr);
}

static int lcp_solve_dev_masked(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts, const int* run_if, double* ws_d, int* ws_i)
{
  mh::LcpParams P;
  int rc = lcp_params(kind, opts, P);
  if (rc != MH_OK) return rc;
  if (B < 0 || n < 0) return fail(MH_ERR_INVALID_ARG, "negative batch (%d) or size (%d)", B, n);
  if (B == 0) return MH_OK;
  if (n == 0) return fail(MH_ERR_INVALID_ARG, "n == 0: the reference returns an empty z without work; handle on the host");
  if (!M || !q || !z || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null M/q/z/rng/status");
  if (ld < n) return fail(MH_ERR_INVALID_ARG, "ld (%d) < n (%d)", ld, n);
  if (strideM < (long)ld * (n - 1) + n) return fail(MH_ERR_INVALID_ARG, "strideM (%ld) smaller than one matrix", strideM);
  if (n > MH_LCP_MAX_N_BLOCK)
    return fail(MH_ERR_UNSUPPORTED_N, "n = %d > %d", n, MH_LCP_MAX_N_BLOCK);
  if (trace && trace_cap <= 0) return fail(MH_ERR_INVALID_ARG, "trace given with trace_cap <= 0");
  static const mh::Pow10Table p10 = make_pow10();
  if (n > MH_LCP_MAX_N_WAVE) {
    // workgroup-per-problem solver; its workspace is allocated and freed in stream order
    double* wsd = ws_d; int* wsi = ws_i;
    // thread geometry: wide when a problem is large enough to feed 16 waves -- always from n = 384 up (n = 512: 1.6x at 256
    // problems, still 1.09x at 1024), below that only while the batch does not fill the chip twice over with the narrow one
    // (n = 256: 1.34x at 256 problems, 1.07x at 512, 0.80x at 1024; n = 128 x 1024: 0.85x)
    bool wide = n >= 384 || (n >= 192 && B <= 2 * g_cu_count());
    if (g_debug_blk == 1) wide = false; else if (g_debug_blk == 2) wide = true;
    if (wsd && wsi) {
      if (wide) hipLaunchKernelGGL(mh::blkw::k_lcp_block, dim3(B), dim3(mh::blkw::T), 0, (hipStream_t)stream,
                         B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                         trace, trace_cap, trace_len, P, p10, wsd, wsi, run_if);
      else hipLaunchKernelGGL(mh::blk::k_lcp_block, dim3(B), dim3(mh::blk::T), 0, (hipStream_t)stream,
                         B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                         trace, trace_cap, trace_len, P, p10, wsd, wsi, run_if);
      MH_HIP(hipGetLastError());
      return MH_OK;
    }
    const size_t nd = (size_t)B * ((size_t)n * n + 5 * (size_t)n), ni = (size_t)B * 4 * (size_t)n;
    MH_HIP(hipMallocAsync((void**)&wsd, nd * sizeof(double), (hipStream_t)stream));
    hipError_t e = hipMallocAsync((void**)&wsi, ni * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipFreeAsync(wsd, (hipStream_t)stream); return fail(MH_ERR_HIP, "workspace allocation failed: %s", hipGetErrorString(e)); }
    if (wide) hipLaunchKernelGGL(mh::blkw::k_lcp_block, dim3(B), dim3(mh::blkw::T), 0, (hipStream_t)stream,
                       B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                       trace, trace_cap, trace_len, P, p10, wsd, wsi, run_if);
    else hipLaunchKernelGGL(mh::blk::k_lcp_block, dim3(B), dim3(mh::blk::T), 0, (hipStream_t)stream,
                       B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                       trace, trace_cap, trace_len, P, p10, wsd, wsi, run_if);
    e = hipGetLastError();
    (void)hipFreeAsync(wsd, (hipStream_t)stream); (void)hipFreeAsync(wsi, (hipStream_t)stream);
    if (e != hipSuccess) return fail(MH_ERR_HIP, "block LCP launch failed: %s", hipGetErrorString(e));
    return MH_OK;
  }
  const size_t lds = (size_t)(2 * n * n + n) * sizeof(double);
  hipLaunchKernelGGL(mh_k_lcp_wave, dim3(B), dim3(64), lds, (hipStream_t)stream,
                     B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                     trace, trace_cap, trace_len, P, p10, run_if);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

int mh_lcp_solve_batch(int kind, int B, int n,
                       const double* M, int ld, long strideM,
                       const double* q, double* z,
                       const int* z_size_in, int* z_size_out,
                       uint32_t* rng, int* status, unsigned* pivots,
                       int32_t* trace, int trace_cap, int* trace_len,
                       const mh_lcp_opts* opts)
{
  if (B <= 0) return (B == 0) ? MH_OK : fail(MH_ERR_INVALID_ARG, "negative batch");
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  if (!M || !q || !z || !rng || !status) return fail(MH_ERR_INVALID_ARG, "null M/q/z/rng/status");
  if (n <= 0 || ld < n) return fail(MH_ERR_INVALID_ARG, "bad n/ld");
  double *dM = nullptr, *dq = nullptr, *dz = nullptr; int *dzi = nullptr, *dzo = nullptr, *dst = nullptr, *dtl = nullptr;
  uint32_t* drng = nullptr; unsigned* dpiv = nullptr; int32_t* dtr = nullptr;
  const size_t szM = ((size_t)(B - 1) * strideM + (size_t)ld * (n - 1) + n) * sizeof(double);
  const size_t szv = (size_t)B * n * sizeof(double);
  int rc = MH_OK;
  auto cleanup = [&]() {
    void* ps[] = {dM, dq, dz, dzi, dzo, dst, dtl, drng, dpiv, dtr};
    for (void* p : ps) if (p) (void)hipFree(p);
  };
#define MH_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); \
  return fail(MH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
  MH_TRY(hipMalloc(&dM, szM)); MH_TRY(hipMalloc(&dq, szv)); MH_TRY(hipMalloc(&dz, szv));
  MH_TRY(hipMalloc(&drng, (size_t)B * MH_RAND_WORDS * 4)); MH_TRY(hipMalloc(&dst, (size_t)B * 4));
  MH_TRY(hipMemcpy(dM, M, szM, hipMemcpyHostToDevice));
  MH_TRY(hipMemcpy(dq, q, szv, hipMemcpyHostToDevice));
  MH_TRY(hipMemcpy(dz, z, szv, hipMemcpyHostToDevice));
  MH_TRY(hipMemcpy(drng, rng, (size_t)B * MH_RAND_WORDS * 4, hipMemcpyHostToDevice));
  if (z_size_in) { MH_TRY(hipMalloc(&dzi, (size_t)B * 4)); MH_TRY(hipMemcpy(dzi, z_size_in, (size_t)B * 4, hipMemcpyHostToDevice)); }
  if (z_size_out) MH_TRY(hipMalloc(&dzo, (size_t)B * 4));
  if (pivots) MH_TRY(hipMalloc(&dpiv, (size_t)B * 4));
  if (trace) { MH_TRY(hipMalloc(&dtr, (size_t)B * trace_cap * 4)); MH_TRY(hipMemset(dtr, 0, (size_t)B * trace_cap * 4)); }
  if (trace_len) MH_TRY(hipMalloc(&dtl, (size_t)B * 4));
  rc = mh_lcp_solve_batch_dev(nullptr, kind, B, n, dM, ld, strideM, dq, dz, dzi, dzo, drng, dst, dpiv,
                              dtr, trace_cap, dtl, opts);
  if (rc != MH_OK) { cleanup(); return rc; }
  MH_TRY(hipDeviceSynchronize());
  MH_TRY(hipMemcpy(z, dz, szv, hipMemcpyDeviceToHost));
  MH_TRY(hipMemcpy(rng, drng, (size_t)B * MH_RAND_WORDS * 4, hipMemcpyDeviceToHost));
  MH_TRY(hipMemcpy(status, dst, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (z_size_out) MH_TRY(hipMemcpy(z_size_out, dzo, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (pivots) MH_TRY(hipMemcpy(pivots, dpiv, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (trace) MH_TRY(hipMemcpy(trace, dtr, (size_t)B * trace_cap * 4, hipMemcpyDeviceToHost));
  if (trace_len) MH_TRY(hipMemcpy(trace_len, dtl, (size_t)B * 4, hipMemcpyDeviceToHost));
  cleanup();
#undef MH_TRY
  return MH_OK;
}

} // extern "C"

// ===========================================================================
// many-worlds stepping
#include <mutex>
// three variants of the world kernel (LDS image, occupancy and feature set differ):
//   small  <= 4 bodies, <= 6 pairs, <= 6 contacts, <= 12 Jacobian rows, spheres + Drumwright-Shell model only
//   wheel  <= 2 bodies, <= 3 pairs, <= 4 contacts, + spokes geometry and the no-slip model (rimless wheel)
//   large  <= 8 bodies, <= 36 pairs, <= 40 contacts, <= 24 rows per island, every feature
#define MHW_NS small
#define MHW_NOSLIP 0
#define MHW_BOX 0
#define MHW_NB 4
#define MHW_MAX_PAIRS 6
#define MHW_MAX_CONTACTS 6
#define MHW_MAX_ROWS 12
#define MHW_MAX_GROWS 12
#define MHW_WAVES_PER_SIMD 4
#include "mh_world_wave.inc"
#undef MHW_NS
#undef MHW_NB
#undef MHW_MAX_PAIRS
#undef MHW_MAX_CONTACTS
#undef MHW_MAX_ROWS
#undef MHW_MAX_GROWS
#undef MHW_WAVES_PER_SIMD
#undef MHW_NOSLIP
#define MHW_NS wheel
#define MHW_NOSLIP 1
#define MHW_NB 2
#define MHW_MAX_PAIRS 3
#define MHW_MAX_CONTACTS 4
#define MHW_MAX_ROWS 12
#define MHW_MAX_GROWS 12
#define MHW_WAVES_PER_SIMD 2
#include "mh_world_wave.inc"
#undef MHW_NS
#undef MHW_NB
#undef MHW_MAX_PAIRS
#undef MHW_MAX_CONTACTS
#undef MHW_MAX_ROWS
#undef MHW_MAX_GROWS
#undef MHW_WAVES_PER_SIMD
#undef MHW_NOSLIP
#undef MHW_BOX
#define MHW_NS large
#define MHW_NOSLIP 1
#define MHW_BOX 1
#define MHW_NB MH_MAX_BODIES
#define MHW_MAX_PAIRS MH_MAX_PAIRS
#define MHW_MAX_CONTACTS 40   /* the stabiliser lists one contact per candidate pair (up to 36), a box adds up to 8 */
#define MHW_MAX_ROWS 24
#define MHW_MAX_GROWS 24
#define MHW_WAVES_PER_SIMD 2
#include "mh_world_wave.inc"

namespace {
int g_debug_ka = 64;          // LDS LU block edge (clamped to the variant's MHW_KA_V); mh_debug_set(1, 0) forces the HBM workspace path
std::once_flag g_tables_once;
hipError_t g_tables_err = hipSuccess;
void init_tables()
{
  static mh::FricTable ft;
  for (int kh = 0; kh < 33; kh++)
    for (int j = 0; j < 32; j++) {
      double c = 0.0, s = 0.0;
      if (kh >= 2 && j < kh) { const double theta = (double)j / (kh - 1) * M_PI_2; c = std::cos(theta); s = std::sin(theta); }  // ICH-QP:466-468
      ft.c[kh][j] = c; ft.s[kh][j] = s;
    }
  g_tables_err = hipMemcpyToSymbol(HIP_SYMBOL(mh::c_fric), &ft, sizeof(ft));
  if (g_tables_err != hipSuccess) return;
  const mh::Pow10Table p10 = make_pow10();
  g_tables_err = hipMemcpyToSymbol(HIP_SYMBOL(mh::c_pow10), &p10, sizeof(p10));
}
int check_scene(const mh_scene* sc)
{
  if (!sc) return fail(MH_ERR_INVALID_ARG, "null scene");
  if (sc->nb < 1 || sc->nb > MH_MAX_BODIES) return fail(MH_ERR_INVALID_ARG, "nb = %d outside [1, %d]", sc->nb, MH_MAX_BODIES);
  const int ntot = sc->nb + (sc->has_ground ? 1 : 0);
  int spokes_body = -1;
  for (int b = 0; b < sc->nb; b++) {
    if (sc->geom_type[b] != MH_GEOM_SPHERE && sc->geom_type[b] != MH_GEOM_SPOKES && sc->geom_type[b] != MH_GEOM_BOX)
      return fail(MH_ERR_INVALID_ARG, "body %d: geometry type %d is not built (sphere, spokes, box)", b, sc->geom_type[b]);
    if (sc->geom_type[b] == MH_GEOM_BOX) {
      if (!(sc->geom_dim[b][1] > 0.0) || !(sc->geom_dim[b][2] > 0.0)) return fail(MH_ERR_INVALID_ARG, "body %d: box edge lengths must be > 0", b);
      for (int o = 0; o < sc->nb; o++) if (o != b) {
        const int i = o < b ? o : b, j = o < b ? b : o;
        if (sc->pair_enabled[i * ntot - (i * (i + 1)) / 2 + (j - i - 1)])
          return fail(MH_ERR_INVALID_ARG, "bodies %d,%d: box-box / box-sphere contact is not built; disable the pair (only box-plane is)", i, j);
      }
    }
    if (sc->geom_type[b] == MH_GEOM_SPOKES) {
      const double N = sc->geom_dim[b][1];
      if (!sc->has_ground) return fail(MH_ERR_INVALID_ARG, "body %d: spokes geometry needs the ground plane", b);
      if (!(N >= 1.0 && N <= (double)MH_MAX_SPOKES) || N != (double)(int)N) return fail(MH_ERR_INVALID_ARG, "body %d: number of spokes outside [1, %d]", b, MH_MAX_SPOKES);
      if (spokes_body >= 0 && (sc->geom_dim[b][0] != sc->geom_dim[spokes_body][0] || N != sc->geom_dim[spokes_body][1]))
        return fail(MH_ERR_INVALID_ARG, "body %d: all spokes geometries of a scene must share R and N", b);
      spokes_body = b;
    }
    if (!(sc->geom_dim[b][0] > 0.0) || !(sc->mass[b] > 0.0)) return fail(MH_ERR_INVALID_ARG, "body %d: radius and mass must be > 0", b);
    for (int k = 0; k < 3; k++) if (!(sc->inertia[b][k] > 0.0)) return fail(MH_ERR_INVALID_ARG, "body %d: inertia must be > 0", b);
  }
  for (int p = 0; p < ntot * (ntot - 1) / 2; p++)
    if (sc->cp_nk[p] < 4 || sc->cp_nk[p] > 64) return fail(MH_ERR_INVALID_ARG, "pair %d: friction-cone-edges %d outside [4, 64]", p, sc->cp_nk[p]);
  if (sc->lcp_n_max < 0 || sc->lcp_n_max > MH_LCP_MAX_N_WAVE) return fail(MH_ERR_INVALID_ARG, "lcp_n_max outside [0, %d]", MH_LCP_MAX_N_WAVE);
  return MH_OK;
}
} // namespace

extern "C" {

int mh_debug_set(int key, int value)
{
  if (key == 1) { if (value < 0 || value > 64) return fail(MH_ERR_INVALID_ARG, "LU block edge outside [0, 64]"); g_debug_ka = value; return MH_OK; }
  if (key == 2) { if (value < 0 || value > 2) return fail(MH_ERR_INVALID_ARG, "block solver geometry outside {0, 1, 2}"); g_debug_blk = value; return MH_OK; }
  return fail(MH_ERR_INVALID_ARG, "unknown debug key %d", key);
}

void mh_scene_defaults(mh_scene* s)
{
  std::memset(s, 0, sizeof(*s));
  s->min_step_size = std::sqrt(2.220446049250313e-16);       // TimeSteppingSimulator.cpp:48
  s->contact_dist_thresh = 1e-6;                             // ConstraintSimulator.cpp:56
  s->cstab_eps = std::sqrt(2.220446049250313e-16);           // ConstraintStabilization.cpp:59
  s->cstab_max_iterations = 0xFFFFFFFFu;                     // ConstraintStabilization.cpp:56
  s->plane_R[0] = s->plane_R[4] = s->plane_R[8] = 1.0;
  for (int p = 0; p < MH_MAX_PAIRS; p++) { s->pair_enabled[p] = 1; s->cp_nk[p] = 4; }   // ContactParameters.cpp:26
}

void mh_world_aux_init(mh_world_aux* a, uint32_t seed)
{
  std::memset(a, 0, sizeof(*a));
  mh_rand_seed(a->rng, seed);
}

typedef void (*mh_world_kernel)(const mh_scene*, int, double, int, double*, mh_world_aux*, double*, int, double*, int, unsigned long long*);

struct mh_world_batch {
  mh_scene scene;
  int B;
  int nmax;
  int variant;               // 0 small, 1 large, 2 wheel
  mh_world_kernel kernel;
  mh_scene* d_scene;
  double* d_lu_ws;
  double* d_state;
  mh_world_aux* d_aux;
};

// diagnostic: blocks per CU the runtime's occupancy query reports for the kernel this batch uses
int mh_world_batch_occupancy(mh_world_batch* wb);
int mh_world_batch_occupancy(mh_world_batch* wb)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  int n = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wb->kernel, 64, 0);
  if (e != hipSuccess) return fail(MH_ERR_HIP, "occupancy query failed: %s", hipGetErrorString(e));
  return n;
}

int mh_world_batch_create(const mh_scene* scene, int B, mh_world_batch** out)
{
  if (!out) return fail(MH_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  int rc = check_scene(scene);
  if (rc != MH_OK) return rc;
  if (B <= 0) return fail(MH_ERR_INVALID_ARG, "batch must be > 0");
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  std::call_once(g_tables_once, init_tables);
  if (g_tables_err != hipSuccess) return fail(MH_ERR_HIP, "constant table upload failed: %s", hipGetErrorString(g_tables_err));
  mh_world_batch* wb = new mh_world_batch();
  wb->scene = *scene; wb->B = B;
  wb->nmax = scene->lcp_n_max ? scene->lcp_n_max : MH_LCP_MAX_N_WAVE;
  {
    // small variant: <= 4 bodies, <= 6 pairs, islands of <= 4 contacts (12 Jacobian rows); the
    // caller opts in by bounding the LCP size (lcp_n_max <= 56 = 4 contacts x (6 + 16/2) rows).
    // A world that outgrows the variant's limits at run time gets MH_WORLD_UNSUPPORTED.
    // Spokes geometry or a pair with mu-coulomb >= 100 (=> the no-slip model, ICH:127-135) needs a
    // variant built with those features: "wheel" for one or two bodies, otherwise "large".
    const int ntot = scene->nb + (scene->has_ground ? 1 : 0), npairs = ntot * (ntot - 1) / 2;
    bool noslip = false, box = false;
    for (int b = 0; b < scene->nb; b++) if (scene->geom_type[b] == MH_GEOM_SPOKES) noslip = true;
    for (int b = 0; b < scene->nb; b++) if (scene->geom_type[b] == MH_GEOM_BOX) box = true;
    for (int p = 0; p < npairs; p++) if (scene->pair_enabled[p] && scene->cp_mu_coulomb[p] >= 1e2) noslip = true;
    if (!noslip && !box && scene->nb <= 4 && npairs <= 6 && scene->lcp_n_max > 0 && scene->lcp_n_max <= 56) { wb->variant = 0; wb->kernel = mh::small::mh_k_world_step; }
    else if (noslip && !box && scene->nb <= 2 && npairs <= 3) { wb->variant = 2; wb->kernel = mh::wheel::mh_k_world_step; }
    else { wb->variant = 1; wb->kernel = mh::large::mh_k_world_step; }
  }
  wb->d_scene = nullptr; wb->d_state = nullptr; wb->d_aux = nullptr; wb->d_lu_ws = nullptr;
  // device scene record, followed by the spoke-tip table p1 = (cos(theta) R, sin(theta) R), theta = pi i 2 / N
  // (coldet-plugin.cpp:104-110), evaluated with the host's libm like the oracle does
  double tips[2 * MH_MAX_SPOKES] = {0.0};
  for (int b = 0; b < scene->nb; b++) if (scene->geom_type[b] == MH_GEOM_SPOKES) {
    const double Rr = scene->geom_dim[b][0]; const int N = (int)scene->geom_dim[b][1];
    for (int i = 0; i < N; i++) { const double theta = M_PI * i * 2.0 / N; tips[2*i] = std::cos(theta) * Rr; tips[2*i+1] = std::sin(theta) * Rr; }
  }
  hipError_t e = hipMalloc(&wb->d_scene, sizeof(mh_scene) + sizeof(tips));
  if (e == hipSuccess) e = hipMalloc(&wb->d_lu_ws, (size_t)B * wb->nmax * wb->nmax * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&wb->d_state, (size_t)B * scene->nb * MH_BODY_STATE * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&wb->d_aux, (size_t)B * sizeof(mh_world_aux));
  if (e == hipSuccess) e = hipMemcpy(wb->d_scene, scene, sizeof(mh_scene), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<char*>(wb->d_scene) + sizeof(mh_scene), tips, sizeof(tips), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(wb->d_state, 0, (size_t)B * scene->nb * MH_BODY_STATE * sizeof(double));
  if (e == hipSuccess) {
    std::vector<mh_world_aux> a((size_t)B);
    mh_world_aux_init(&a[0], 1);
    for (int b = 1; b < B; b++) a[b] = a[0];
    e = hipMemcpy(wb->d_aux, a.data(), (size_t)B * sizeof(mh_world_aux), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) { mh_world_batch_destroy(wb); return fail(MH_ERR_HIP, "device allocation/upload failed: %s", hipGetErrorString(e)); }
  *out = wb;
  return MH_OK;
}

int mh_world_batch_destroy(mh_world_batch* wb)
{
  if (!wb) return MH_OK;
  if (wb->d_scene) (void)hipFree(wb->d_scene);
  if (wb->d_state) (void)hipFree(wb->d_state);
  if (wb->d_aux) (void)hipFree(wb->d_aux);
  if (wb->d_lu_ws) (void)hipFree(wb->d_lu_ws);
  delete wb;
  return MH_OK;
}

int mh_world_batch_upload(mh_world_batch* wb, const double* state, const mh_world_aux* aux)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  if (state) MH_HIP(hipMemcpy(wb->d_state, state, (size_t)wb->B * wb->scene.nb * MH_BODY_STATE * sizeof(double), hipMemcpyHostToDevice));
  if (aux) MH_HIP(hipMemcpy(wb->d_aux, aux, (size_t)wb->B * sizeof(mh_world_aux), hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_world_batch_step(mh_world_batch* wb, void* stream, double dt, int nsteps, double* traj_dev)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  if (nsteps < 0) return fail(MH_ERR_INVALID_ARG, "negative step count");
  if (nsteps == 0) return MH_OK;
  if (!(dt > 0.0)) return fail(MH_ERR_INVALID_ARG, "dt must be > 0");
  hipLaunchKernelGGL(wb->kernel, dim3(wb->B), dim3(64), 0, (hipStream_t)stream,
                     (const mh_scene*)wb->d_scene, wb->B, dt, nsteps, wb->d_state, wb->d_aux, traj_dev, wb->nmax, wb->d_lu_ws, g_debug_ka,
                     (unsigned long long*)nullptr);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// diagnostic: one launch with per-phase cycle accumulators (mh::PH_*), averaged over worlds on the host
int mh_world_batch_profile(mh_world_batch* wb, double dt, int nsteps, double* phase_cycles, int nphase)
{
  if (!wb || !phase_cycles) return fail(MH_ERR_INVALID_ARG, "null argument");
  unsigned long long* dprof = nullptr;
  const size_t sz = (size_t)wb->B * mh::large::PH_COUNT * sizeof(unsigned long long);
  MH_HIP(hipMalloc(&dprof, sz));
  MH_HIP(hipMemset(dprof, 0, sz));
  hipLaunchKernelGGL(wb->kernel, dim3(wb->B), dim3(64), 0, (hipStream_t)nullptr,
                     (const mh_scene*)wb->d_scene, wb->B, dt, nsteps, wb->d_state, wb->d_aux, (double*)nullptr, wb->nmax, wb->d_lu_ws, g_debug_ka, dprof);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)wb->B * mh::large::PH_COUNT);
  if (e == hipSuccess) e = hipMemcpy(h.data(), dprof, sz, hipMemcpyDeviceToHost);
  (void)hipFree(dprof);
  if (e != hipSuccess) return fail(MH_ERR_HIP, "profile launch failed: %s", hipGetErrorString(e));
  for (int p = 0; p < nphase; p++) {
    double acc = 0.0;
    if (p < mh::large::PH_COUNT) for (int b = 0; b < wb->B; b++) acc += (double)h[(size_t)b * mh::large::PH_COUNT + p];
    phase_cycles[p] = acc / wb->B;
  }
  // entries PH_COUNT, PH_COUNT+1 (if asked for): the slowest and the fastest world's stamped total --
  // the launch lasts as long as its slowest world
  if (nphase >= mh::large::PH_COUNT + 2) {
    double mx = 0.0, mn = 1e300;
    for (int b = 0; b < wb->B; b++) {
      double t = 0.0;
      for (int p = 0; p < 10; p++) t += (double)h[(size_t)b * mh::large::PH_COUNT + p];
      mx = t > mx ? t : mx; mn = t < mn ? t : mn;
    }
    phase_cycles[mh::large::PH_COUNT] = mx; phase_cycles[mh::large::PH_COUNT + 1] = mn;
  }
  return MH_OK;
}

int mh_world_batch_download(mh_world_batch* wb, double* state, mh_world_aux* aux)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_HIP(hipDeviceSynchronize());
  if (state) MH_HIP(hipMemcpy(state, wb->d_state, (size_t)wb->B * wb->scene.nb * MH_BODY_STATE * sizeof(double), hipMemcpyDeviceToHost));
  if (aux) MH_HIP(hipMemcpy(aux, wb->d_aux, (size_t)wb->B * sizeof(mh_world_aux), hipMemcpyDeviceToHost));
  return MH_OK;
}

int mh_world_batch_device_ptrs(mh_world_batch* wb, double** state_dev, mh_world_aux** aux_dev)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  if (state_dev) *state_dev = wb->d_state;
  if (aux_dev) *aux_dev = wb->d_aux;
  return MH_OK;
}

int mh_world_step_batch(const mh_scene* scene, int B, double dt, int nsteps,
                        double* state, mh_world_aux* aux, double* traj)
{
  if (B == 0 || nsteps == 0) return MH_OK;
  if (B < 0 || nsteps < 0) return fail(MH_ERR_INVALID_ARG, "negative batch or step count");
  if (!state || !aux) return fail(MH_ERR_INVALID_ARG, "null state/aux");
  mh_world_batch* wb = nullptr;
  int rc = mh_world_batch_create(scene, B, &wb);
  if (rc != MH_OK) return rc;
  double* dtraj = nullptr;
  const size_t sz_tr = (size_t)B * nsteps * scene->nb * 7 * sizeof(double);
  rc = mh_world_batch_upload(wb, state, aux);
  if (rc == MH_OK && traj && hipMalloc(&dtraj, sz_tr) != hipSuccess) rc = fail(MH_ERR_HIP, "trajectory allocation failed");
  if (rc == MH_OK) rc = mh_world_batch_step(wb, nullptr, dt, nsteps, dtraj);
  if (rc == MH_OK) rc = mh_world_batch_download(wb, state, aux);
  if (rc == MH_OK && traj && hipMemcpy(traj, dtraj, sz_tr, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MH_ERR_HIP, "trajectory download failed");
  if (dtraj) (void)hipFree(dtraj);
  mh_world_batch_destroy(wb);
  return rc;
}

} // extern "C"

// ===========================================================================
// batched impact handler (include/moby_hip_impact.h)
#include "mh_impact.inc"
