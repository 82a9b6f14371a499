// Host side of the many-worlds stepping ABI (include/moby_hip.h: mh_world_batch_*).  The kernels live in
// mh_world_{small,wheel,large}.hip; this file chooses the variant for a scene and owns the device buffers.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/moby_hip.h"
#include "mh_host.h"

namespace mh {
// layouts of the constant tables of mh_world_common.h / mh_lcp_wave.h (checked by size in upload_tables)
struct FricTableH { double c[33][32]; double s[33][32]; };
struct Pow10TableH { double v[64]; };
}
#include <mutex>
namespace {
// The __constant__ tables are one copy per DEVICE (hipMemcpyToSymbol writes the current device's): uploaded once per device, keyed on the
// device current at create, under a lock -- as mh_artic_batch_create does for its own table.
std::mutex g_tables_mu;
std::vector<char> g_tables_done;
hipError_t init_tables()
{
  hipError_t g_tables_err = hipSuccess;
  static mh::FricTableH ft;
  for (int kh = 0; kh < 33; kh++)
    for (int j = 0; j < 32; j++) {
      double c = 0.0, s = 0.0;
      if (kh >= 2 && j < kh) { const double theta = (double)j / (kh - 1) * M_PI_2; c = std::cos(theta); s = std::sin(theta); }  // ICH-QP:466-468
      ft.c[kh][j] = c; ft.s[kh][j] = s;
    }
  mh::Pow10TableH p10;
  for (int i = 0; i < 64; i++) p10.v[i] = std::pow(10.0, (double)(i - 32)); // LCP.cpp:285
  const mh_world_variant* vs[6] = { mh_world_variant_small(), mh_world_variant_wheel(), mh_world_variant_large(),
                                    mh_world_variant_small_prof(), mh_world_variant_wheel_prof(), mh_world_variant_large_prof() };     // (one copy of the tables per code object)
  for (int i = 0; i < 6 && g_tables_err == hipSuccess; i++) g_tables_err = vs[i]->upload_tables(&ft, sizeof(ft), &p10, sizeof(p10));
  return g_tables_err;
}
hipError_t tables_for_current_device()
{
  std::lock_guard<std::mutex> lk(g_tables_mu);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if ((int)g_tables_done.size() <= dev) g_tables_done.resize(dev + 1, 0);
  if (!g_tables_done[dev]) { e = init_tables(); if (e == hipSuccess) g_tables_done[dev] = 1; }
  return e;
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

void mh_scene_defaults(mh_scene* s)
{
  std::memset(s, 0, sizeof(*s));
  s->min_step_size = std::sqrt(2.220446049250313e-16);       // TimeSteppingSimulator.cpp:48
  s->contact_dist_thresh = 1e-6;                             // ConstraintSimulator.cpp:56
  s->cstab_eps = std::sqrt(2.220446049250313e-16);           // ConstraintStabilization.cpp:59
  s->cstab_max_iterations = MH_CSTAB_DEFAULT_MAX_ITERATIONS; // ConstraintStabilization.cpp:56 is UINT_MAX: see moby_hip.h
  s->plane_R[0] = s->plane_R[4] = s->plane_R[8] = 1.0;
  for (int p = 0; p < MH_MAX_PAIRS; p++) { s->pair_enabled[p] = 1; s->cp_nk[p] = 4; }   // ContactParameters.cpp:26
}

void mh_world_aux_init(mh_world_aux* a, uint32_t seed)
{
  std::memset(a, 0, sizeof(*a));
  mh_rand_seed(a->rng, seed);
}


struct mh_world_batch {
  int device;                // the HIP device the batch lives on (current at create); every entry point runs there (MH_ON_DEVICE)
  mh_scene scene;
  int B;
  int nmax;
  int variant;               // 0 small, 1 large, 2 wheel
  mh_world_kernel kernel;
  int ph_count;
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
  MH_ON_DEVICE(wb);
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
  { const hipError_t te = tables_for_current_device();
    if (te != hipSuccess) return fail(MH_ERR_HIP, "constant table upload failed: %s", hipGetErrorString(te)); }
  mh_world_batch* wb = new mh_world_batch();
  if (hipGetDevice(&wb->device) != hipSuccess) { delete wb; return fail(MH_ERR_HIP, "hipGetDevice failed"); }
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
    if (!noslip && !box && scene->nb <= 4 && npairs <= 6 && scene->lcp_n_max > 0 && scene->lcp_n_max <= 56) { wb->variant = 0; wb->kernel = mh_world_variant_small()->kernel; }
    else if (noslip && !box && scene->nb <= 2 && npairs <= 3) { wb->variant = 2; wb->kernel = mh_world_variant_wheel()->kernel; }
    else { wb->variant = 1; wb->kernel = mh_world_variant_large()->kernel; }
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

int mh_world_batch_device(const mh_world_batch* wb) { return wb ? wb->device : fail(MH_ERR_INVALID_ARG, "null batch"); }

int mh_world_batch_destroy(mh_world_batch* wb)
{
  if (!wb) return MH_OK;
  MH_ON_DEVICE(wb);
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
  MH_ON_DEVICE(wb);
  if (state) MH_HIP(hipMemcpy(wb->d_state, state, (size_t)wb->B * wb->scene.nb * MH_BODY_STATE * sizeof(double), hipMemcpyHostToDevice));
  if (aux) MH_HIP(hipMemcpy(wb->d_aux, aux, (size_t)wb->B * sizeof(mh_world_aux), hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_world_batch_step(mh_world_batch* wb, void* stream, double dt, int nsteps, double* traj_dev)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(wb);
  if (nsteps < 0) return fail(MH_ERR_INVALID_ARG, "negative step count");
  if (nsteps == 0) return MH_OK;
  if (!(dt > 0.0)) return fail(MH_ERR_INVALID_ARG, "dt must be > 0");
  hipLaunchKernelGGL(wb->kernel, dim3(wb->B), dim3(64), 0, (hipStream_t)stream,
                     (const mh_scene*)wb->d_scene, wb->B, dt, nsteps, wb->d_state, wb->d_aux, traj_dev, wb->nmax, wb->d_lu_ws, mh_g_debug_ka,
                     (unsigned long long*)nullptr, (const int*)nullptr);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// nsteps x step(dt) of the worlds ids[0 .. count) only (device pointer), on the given stream: worlds are independent, so a batch can be
// split over streams -- e.g. the few worlds whose solver chain runs to its pivot caps in their own launch, so that the next interval of
// the others does not wait for them (bench.py's long_horizon leg).  Launches on different streams must not share a world.
int mh_world_batch_step_ids(mh_world_batch* wb, void* stream, double dt, int nsteps, const int* ids_dev, int count)
{
  if (!wb || !ids_dev) return fail(MH_ERR_INVALID_ARG, "null batch / id list");
  MH_ON_DEVICE(wb);
  if (nsteps < 0 || count < 0 || count > wb->B) return fail(MH_ERR_INVALID_ARG, "bad step count (%d) or id count (%d of %d)", nsteps, count, wb->B);
  if (nsteps == 0 || count == 0) return MH_OK;
  if (!(dt > 0.0)) return fail(MH_ERR_INVALID_ARG, "dt must be > 0");
  hipLaunchKernelGGL(wb->kernel, dim3(count), dim3(64), 0, (hipStream_t)stream,
                     (const mh_scene*)wb->d_scene, wb->B, dt, nsteps, wb->d_state, wb->d_aux, (double*)nullptr, wb->nmax, wb->d_lu_ws, mh_g_debug_ka,
                     (unsigned long long*)nullptr, ids_dev);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

int mh_world_profile_phase_count(void) { return mh_world_variant_large()->ph_count; }

// diagnostic: one launch with per-phase cycle accumulators (mh::PH_*), averaged over worlds on the host
int mh_world_batch_profile(mh_world_batch* wb, double dt, int nsteps, double* phase_cycles, int nphase)
{
  if (!wb || !phase_cycles) return fail(MH_ERR_INVALID_ARG, "null argument");
  MH_ON_DEVICE(wb);
  unsigned long long* dprof = nullptr;
  const int PHC = mh_world_variant_large()->ph_count;   // the same enum in every variant
  const size_t sz = (size_t)wb->B * PHC * sizeof(unsigned long long);
  MH_HIP(hipMalloc(&dprof, sz));
  MH_HIP(hipMemset(dprof, 0, sz));
  // the PROFILE build of the batch's variant: the production kernel has no stamp code (mh_lcp_wave.h lp_tick)
  const mh_world_kernel kprof = (wb->variant == 0 ? mh_world_variant_small_prof() : wb->variant == 2 ? mh_world_variant_wheel_prof() : mh_world_variant_large_prof())->kernel;
  hipLaunchKernelGGL(kprof, dim3(wb->B), dim3(64), 0, (hipStream_t)nullptr,
                     (const mh_scene*)wb->d_scene, wb->B, dt, nsteps, wb->d_state, wb->d_aux, (double*)nullptr, wb->nmax, wb->d_lu_ws, mh_g_debug_ka, dprof, (const int*)nullptr);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)wb->B * PHC);
  if (e == hipSuccess) e = hipMemcpy(h.data(), dprof, sz, hipMemcpyDeviceToHost);
  (void)hipFree(dprof);
  if (e != hipSuccess) return fail(MH_ERR_HIP, "profile launch failed: %s", hipGetErrorString(e));
  for (int p = 0; p < nphase; p++) {
    double acc = 0.0;
    if (p < PHC) for (int b = 0; b < wb->B; b++) acc += (double)h[(size_t)b * PHC + p];
    phase_cycles[p] = acc / wb->B;
  }
  // entries PH_COUNT, PH_COUNT+1 (if asked for): the slowest and the fastest world's stamped total --
  // the launch lasts as long as its slowest world
  if (nphase >= PHC + 2) {
    double mx = 0.0, mn = 1e300;
    for (int b = 0; b < wb->B; b++) {
      double t = 0.0;
      for (int p = 0; p < 10; p++) t += (double)h[(size_t)b * PHC + p];
      mx = t > mx ? t : mx; mn = t < mn ? t : mn;
    }
    phase_cycles[PHC] = mx; phase_cycles[PHC + 1] = mn;
    // PH_COUNT + 2, + 3 (if asked for): the mean world's total, and the total of the world at the 99th percentile (1 % of the
    // worlds take longer): with the launch lasting `mx`, (mx - p99) / mx of it runs with under 1 % of the waves alive
    if (nphase >= PHC + 4) {
      std::vector<double> tt((size_t)wb->B);
      double sum = 0.0;
      for (int b = 0; b < wb->B; b++) { double t = 0.0; for (int p = 0; p < 10; p++) t += (double)h[(size_t)b * PHC + p]; tt[b] = t; sum += t; }
      std::sort(tt.begin(), tt.end());
      phase_cycles[PHC + 2] = sum / wb->B;
      phase_cycles[PHC + 3] = tt[(size_t)((wb->B - 1) * 0.99)];
    }
  }
  return MH_OK;
}

int mh_world_batch_download(mh_world_batch* wb, double* state, mh_world_aux* aux)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(wb);
  MH_HIP(hipDeviceSynchronize());
  if (state) MH_HIP(hipMemcpy(state, wb->d_state, (size_t)wb->B * wb->scene.nb * MH_BODY_STATE * sizeof(double), hipMemcpyDeviceToHost));
  if (aux) MH_HIP(hipMemcpy(aux, wb->d_aux, (size_t)wb->B * sizeof(mh_world_aux), hipMemcpyDeviceToHost));
  return MH_OK;
}

// The per-interval reduction of SURVEY 8(e), device side: what one device contributes to the node's SUM and MAX vectors, left in
// device memory for a collective (RCCL's ncclAllReduce) -- no host round trip, nothing per step.
namespace mh { namespace world_red {
__global__ __launch_bounds__(256)
void k_counters(const mh_world_aux* __restrict__ aux, int B, unsigned long long* __restrict__ sums, unsigned long long* __restrict__ maxs)
{
  __shared__ unsigned long long s_s[MH_COUNTERS][4], s_m[MH_COUNTERS][4];
  unsigned long long v[MH_COUNTERS] = { 0, 0, 0, 0, 0, 0, 0, 0 }, m[MH_COUNTERS] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int w = blockIdx.x * 256 + threadIdx.x; w < B; w += gridDim.x * 256) {
    const mh_world_aux& a = aux[w];
    const unsigned long long c[MH_COUNTERS] = { a.steps, a.lcp_rows, a.lcp_pivots, (unsigned long long)((a.status & ~MH_WORLD_IMPACT_TOL) != 0),
                                                a.lcp_solves, a.mini_steps, a.stab_iters, a.lcp_alg_bytes };
#pragma unroll
    for (int k = 0; k < MH_COUNTERS; k++) { v[k] += c[k]; m[k] = (c[k] > m[k]) ? c[k] : m[k]; }
  }
#pragma unroll
  for (int k = 0; k < MH_COUNTERS; k++) {
    for (int off = 32; off > 0; off >>= 1) { v[k] += __shfl_xor(v[k], off); const unsigned long long o = __shfl_xor(m[k], off); m[k] = (o > m[k]) ? o : m[k]; }
    if ((threadIdx.x & 63) == 0) { s_s[k][threadIdx.x >> 6] = v[k]; s_m[k][threadIdx.x >> 6] = m[k]; }
  }
  __syncthreads();
  if (threadIdx.x < MH_COUNTERS) {
    const int k = threadIdx.x;
    unsigned long long a = 0, b = 0;
    for (int q = 0; q < 4; q++) { a += s_s[k][q]; b = (s_m[k][q] > b) ? s_m[k][q] : b; }
    atomicAdd(sums + k, a); atomicMax(maxs + k, b);
  }
}
}}

int mh_world_batch_counters_dev(mh_world_batch* wb, void* stream, unsigned long long* sums_dev, unsigned long long* maxs_dev)
{
  if (!wb || !sums_dev || !maxs_dev) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(wb);
  MH_HIP(hipMemsetAsync(sums_dev, 0, MH_COUNTERS * sizeof(unsigned long long), (hipStream_t)stream));
  MH_HIP(hipMemsetAsync(maxs_dev, 0, MH_COUNTERS * sizeof(unsigned long long), (hipStream_t)stream));
  const int grid = std::min((wb->B + 255) / 256, 1024);
  hipLaunchKernelGGL(mh::world_red::k_counters, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const mh_world_aux*)wb->d_aux, wb->B, sums_dev, maxs_dev);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

int mh_world_batch_device_ptrs(mh_world_batch* wb, double** state_dev, mh_world_aux** aux_dev)
{
  if (!wb) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(wb);
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