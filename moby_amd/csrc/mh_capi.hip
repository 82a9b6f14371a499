// libmoby_hip.so: kernels + the C ABI declared in include/moby_hip.h.
// gfx950 only.  Build: see __graft_entry__.build() / moby_amd/csrc/Makefile.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>
#include "../../include/moby_hip.h"
#include "mh_host.h"
#include "mh_lcp_wave.h"
// the workgroup-per-problem solver (n > 64) lives in mh_lcp_blk.hip / mh_lcp_blkw.hip: two thread geometries, 256 threads (four
// problems per CU: throughput when the batch is larger than the chip) and 1024 threads (one problem per CU with 16 waves to hide
// its round trips: 1.2x / 1.4x faster per problem at n = 256 / 512, slower at n = 128)

static thread_local char g_err[512] = "";

int mh_fail(int code, const char* fmt, ...)
{
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

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
                   mh::LcpParams P, mh::Pow10Table p10, const int* __restrict__ run_if, const int* __restrict__ n_arr)
{
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  if (b >= B) return;
  if (run_if && run_if[b] == 0) return;   // masked problem: every output of it is left untouched
  // per-problem sizes (islands of different worlds): q / z / M keep the strides of the largest problem, M is
  // compact (ld = its own n); problems this kernel does not cover belong to the block solver of the same call
  const int nstride = n;
  if (n_arr) { n = mh::uni(n_arr[b]); ld = n; if (n < 1 || n > MH_LCP_MAX_N_WAVE) return; }
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
  const double qi = valid ? qg[(size_t)b * nstride + lane] : 0.0;
  int zsize = zsz_in ? mh::uni(zsz_in[b]) : n;
  double zi = (valid && zsize == n) ? zg[(size_t)b * nstride + lane] : 0.0;
  if (lane == 0) mh::g_lcp_prof_on = 0;
  mh::wave_sync();
  mh::WaveRand rng; rng.load(rngg + (size_t)b * MH_RAND_WORDS);
  mh::Trace tr; tr.buf = trace ? trace + (size_t)b * trace_cap : nullptr; tr.cap = trace_cap; tr.len = 0;
  unsigned piv = 0;
  mh::DenseLds Md; Md.M = Ms; Md.n = n;
  mh::LuScratch S; S.small = A; S.ka = n; S.big = A;
  const bool ok = mh::lcp_solve_wave(P, p10, n, Md, S, art, nrm0, dii, qi, zi, zsize, rng, piv, tr);
  if (valid) zg[(size_t)b * nstride + lane] = zi;
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

int mh_version(void) { return MH_VERSION; }
const char* mh_last_error(void) { return g_err; }

int mh_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mh_device_get(void)
{
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return fail(MH_ERR_NO_DEVICE, "no HIP device visible"); }
  return d;
}

int mh_device_set(int device)
{
  const int n = mh_device_count();
  if (n <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(MH_ERR_INVALID_ARG, "device %d of %d", device, n);
  MH_HIP(hipSetDevice(device));
  return MH_OK;
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

int mh_g_debug_tasks = 3;      // mh_debug_set(4, v): 0 = the Lemke ladder in sequence, 1 = as (world, attempt) tasks, 2 = tasks + speculation beside lcp_fast, 3 = 2 + at full chip (n <= 512) the tasks launched behind lcp_fast and a gate on a second stream, 4 = 3 at any n
int mh_g_debug_repeats = 1;  // mh_debug_set(5, v): 1 = lcp_fast (n > 64) skips the repetitions of a repeating pivot sequence (mh_lcp_block.h), 0 = runs them
int mh_g_debug_lpt = 0;      // mh_debug_set(11, v): (measured, no gain: profiles/r05_g_config4_launch_order_lpt.txt; off) 1 = on a full chip lcp_fast's workgroups and the ladder's hand-out take the worlds in the order of the solver time they have used so far (longest first), 0 = by index
int mh_g_debug_sched = 1;    // mh_debug_set(7, v): 1 = the ladder's tasks are handed out by need (pick_task), 0 = by block index, attempt-major
int mh_g_debug_reuse = 1;    // mh_debug_set(6, v): 1 = the structure-exploiting LU keeps the factors of the columns before the one a Lemke pivot changed, 0 = factorises from scratch
int mh_g_debug_compact = 1;  // mh_debug_set(3, v): 1 = Lemke's bases through the structure-exploiting LU (mh_lu_compact.inc), 0 = dense LU only
int mh_g_debug_artic_pack = 0;
int mh_g_debug_reglu = 1;    // mh_debug_set(10, v): 1 = lcp_fast (n > 64, 1024-thread geometry) solves _Msub of up to 191 rows in registers (mh_lu_reg.inc), 0 = through the HBM workspace
int mh_g_debug_fastgeom = 0; // mh_debug_set(8, v): the lcp_fast kinds' thread geometry for n <= 512 -- 0 choose, 1 = 256, 2 = 1024, 3 = 64, 4 = 128 threads per problem
int mh_g_debug_blk = 0;      // mh_debug_set(2, v): 0 = choose, 1 = 256-thread block solver, 2 = 1024-thread block solver, 3 = one wavefront per problem (lcp_lemke kinds, n <= 512)
int mh_g_debug_ka = 64;          // LDS LU block edge of the world kernel (clamped to the variant MHW_KA_V); mh_debug_set(1, 0) forces the HBM workspace path
int mh_cu_count()
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

int mh_lcp_solve_batch_dev(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts)
{
  return mh_lcp_solve_dev_masked(stream, kind, B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                              trace, trace_cap, trace_len, opts, nullptr, nullptr, nullptr, nullptr);
}

int mh_lcp_solve_dev_masked(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts, const int* run_if, double* ws_d, int* ws_i, const int* n_arr, double* work, int wave_only, int* started, int ordered)
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
    bool wide = n >= 384 || (n >= 192 && B <= 2 * mh_cu_count());
    // the lcp_lemke kinds run the structure-exploiting LU (mh_lu_compact.inc), which is bound by its chain of round trips, not
    // by arithmetic: several narrow problems per CU (three now, four when this was measured) overlap theirs (8-box stacks x 1024: 5.7 s narrow, 10.8 s wide); the wide geometry
    // keeps the sizes the narrow one's compact path does not take (n > 512) and batches of at most one problem per CU
    if (kind == MH_LCP_LEMKE || kind == MH_LCP_LEMKE_REG) wide = n > 512 || (n >= 192 && B <= mh_cu_count());
    if (wave_only == 2) wide = false;
    if (mh_g_debug_blk == 1) wide = false; else if (mh_g_debug_blk == 2) wide = true;
    // the lcp_lemke kinds, n <= 512, far more problems than CUs: one wavefront per problem (mh_lcp_blk1.hip)
    const bool lemke_kind = kind == MH_LCP_LEMKE || kind == MH_LCP_LEMKE_REG;
    const bool one_wave = lemke_kind && n <= 512 && (mh_g_debug_blk == 3 || (mh_g_debug_blk == 0 && B >= MH_BLK1_MIN_PER_CU * mh_cu_count()));
    const bool two_waves = lemke_kind && n <= 512 && (mh_g_debug_blk == 4 || (mh_g_debug_blk == 0 && B >= MH_BLK2_MIN_PER_CU * mh_cu_count()));
    auto launcher = two_waves ? mh_launch_lcp_blk2 : one_wave ? mh_launch_lcp_blk1 : (wide ? mh_launch_lcp_blkw : mh_launch_lcp_blk);
    if (lemke_kind && n > 512 && n <= 1024 && (mh_g_debug_blk == 5 || (mh_g_debug_blk == 0 && B >= MH_BLKY_MIN_TASKS_PER_CU * mh_cu_count()))) launcher = mh_launch_lcp_blky;   // four rows per lane, two problems per CU (mh_lcp_blky.hip)
    if (lemke_kind && n >= MH_BLKX_MIN_N && n <= MH_BLKX_MAX_N && (mh_g_debug_blk == 0 || mh_g_debug_blk == 2)) launcher = mh_launch_lcp_blkx;   // two rows per lane (mh_lcp_blkx.hip)
    if (!lemke_kind && n <= 512 && mh_g_debug_fastgeom) launcher = (mh_g_debug_fastgeom == 1) ? mh_launch_lcp_blk : (mh_g_debug_fastgeom == 2) ? mh_launch_lcp_blkw
                                                                 : (mh_g_debug_fastgeom == 3) ? mh_launch_lcp_blk1 : mh_launch_lcp_blk2;
    if (wsd && wsi) {
      if (n_arr) {          // the problems of this call that fit one wavefront (n_arr[b] <= 64) take the wave solver
        // FIRST: the two kernels share no problem, and the wave solver's workgroups ask for 66 KB of LDS each -- behind the block solver's launch they could not be
        // placed beside the ladder's resident workgroups of the other stream and sat in the queue for seconds (profiles/r04_e_config4_3steps_kernel_trace.txt:
        // 5.3 s of "duration" for a launch whose every workgroup returns at once when all problems have more than 64 rows)
        const size_t ldsw = (size_t)(2 * MH_LCP_MAX_N_WAVE * MH_LCP_MAX_N_WAVE + MH_LCP_MAX_N_WAVE) * sizeof(double);
        hipLaunchKernelGGL(mh_k_lcp_wave, dim3(B), dim3(64), ldsw, (hipStream_t)stream,
                           B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                           trace, trace_cap, trace_len, P, p10, run_if, n_arr);
        MH_HIP(hipGetLastError());
      }
      const hipError_t le = (wave_only == 1) ? hipSuccess : launcher(stream, kind, B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                                   trace, trace_cap, trace_len, &P, &p10, wsd, wsi, run_if, n_arr, mh_g_debug_compact | (mh_g_debug_repeats << 1) | (mh_g_debug_reuse << 2) | (mh_g_debug_reglu << 4) | ((started && !lemke_kind) ? 32 : 0) | ((started && !lemke_kind && ordered) ? 128 : 0), work, 0, lemke_kind ? nullptr : started);
      MH_HIP(le);
      return MH_OK;
    }
    if (n_arr) return fail(MH_ERR_INVALID_ARG, "per-problem sizes need a caller-owned workspace");
    const size_t nd = (size_t)B * ((size_t)n * n + 5 * (size_t)n), ni = (size_t)B * 4 * (size_t)n;
    MH_HIP(hipMallocAsync((void**)&wsd, nd * sizeof(double), (hipStream_t)stream));
    hipError_t e = hipMallocAsync((void**)&wsi, ni * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipFreeAsync(wsd, (hipStream_t)stream); return fail(MH_ERR_HIP, "workspace allocation failed: %s", hipGetErrorString(e)); }
    const hipError_t le = launcher(stream, kind, B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                                   trace, trace_cap, trace_len, &P, &p10, wsd, wsi, run_if, n_arr, mh_g_debug_compact | (mh_g_debug_repeats << 1) | (mh_g_debug_reuse << 2) | (mh_g_debug_reglu << 4) | ((started && !lemke_kind) ? 32 : 0) | ((started && !lemke_kind && ordered) ? 128 : 0), work, 0, lemke_kind ? nullptr : started);
    e = le;
    (void)hipFreeAsync(wsd, (hipStream_t)stream); (void)hipFreeAsync(wsi, (hipStream_t)stream);
    if (e != hipSuccess) return fail(MH_ERR_HIP, "block LCP launch failed: %s", hipGetErrorString(e));
    return MH_OK;
  }
  const size_t lds = (size_t)(2 * n * n + n) * sizeof(double);
  hipLaunchKernelGGL(mh_k_lcp_wave, dim3(B), dim3(64), lds, (hipStream_t)stream,
                     B, n, M, ld, strideM, q, z, z_size_in, z_size_out, rng, status, pivots,
                     trace, trace_cap, trace_len, P, p10, run_if, n_arr);
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
extern "C" int mh_debug_set(int key, int value)
{
  if (key == 1) { if (value < 0 || value > 64) return fail(MH_ERR_INVALID_ARG, "LU block edge outside [0, 64]"); mh_g_debug_ka = value; return MH_OK; }
  if (key == 10) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "register-LU switch outside {0, 1}"); mh_g_debug_reglu = value; return MH_OK; }
  if (key == 9) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "articulated packing outside {0, 1}"); mh_g_debug_artic_pack = value; return MH_OK; }
  if (key == 8) { if (value < 0 || value > 4) return fail(MH_ERR_INVALID_ARG, "lcp_fast geometry outside {0 .. 4}"); mh_g_debug_fastgeom = value; return MH_OK; }
  if (key == 2) { if (value < 0 || value > 5) return fail(MH_ERR_INVALID_ARG, "block solver geometry outside {0 .. 5}"); mh_g_debug_blk = value; return MH_OK; }
  if (key == 4) { if (value < 0 || value > 4) return fail(MH_ERR_INVALID_ARG, "ladder-task switch outside {0 .. 4}"); mh_g_debug_tasks = value; return MH_OK; }
  if (key == 11) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "launch-order switch outside {0, 1}"); mh_g_debug_lpt = value; return MH_OK; }
  if (key == 7) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "task-scheduling switch outside {0, 1}"); mh_g_debug_sched = value; return MH_OK; }
  if (key == 6) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "factor-reuse switch outside {0, 1}"); mh_g_debug_reuse = value; return MH_OK; }
  if (key == 5) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "repeat-skipping switch outside {0, 1}"); mh_g_debug_repeats = value; return MH_OK; }
  if (key == 3) { if (value < 0 || value > 1) return fail(MH_ERR_INVALID_ARG, "compact-LU switch outside {0, 1}"); mh_g_debug_compact = value; return MH_OK; }
  return fail(MH_ERR_INVALID_ARG, "unknown debug key %d", key);
}
