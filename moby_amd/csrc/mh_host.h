// Host-side internals shared by the translation units of libmoby_hip.so (hidden visibility: none of this is ABI).
// Each .hip file is its own code object (no relocatable device code): kernels never call across files, only the
// host functions below do.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include "../../include/moby_hip.h"

#define MH_HIDDEN __attribute__((visibility("hidden")))

extern "C" {   // C linkage only so that definitions may sit inside the extern "C" blocks of the ABI files

// sets the thread-local message mh_last_error() returns, and hands `code` back
MH_HIDDEN int mh_fail(int code, const char* fmt, ...);
#define fail mh_fail

#define MH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return mh_fail(MH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

MH_HIDDEN int mh_cu_count();

}  // extern "C"
// Device affinity (include/moby_hip.h "Devices"): a batch lives on the device that was current when it was created, and every entry
// point that takes it runs THERE whatever device the calling thread has current -- the guard switches and restores on the way out.
struct mh_dev_guard {
  int prev = -1, want = -1; hipError_t err = hipSuccess;
  explicit mh_dev_guard(int dev) : want(dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != want) err = hipSetDevice(want); else prev = -1;
  }
  ~mh_dev_guard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
  mh_dev_guard(const mh_dev_guard&) = delete; mh_dev_guard& operator=(const mh_dev_guard&) = delete;
};
#define MH_ON_DEVICE(h_) mh_dev_guard dev_guard_((h_)->device); \
  if (dev_guard_.err != hipSuccess) return mh_fail(MH_ERR_HIP, "cannot switch to device %d, the batch's: %s", (h_)->device, hipGetErrorString(dev_guard_.err))
extern "C" {

// the LCP entry with a per-problem mask (run_if[b] == 0: problem b is skipped, outputs untouched) and an optional
// caller-owned block-solver workspace (ws_d: B (n^2 + 5n) doubles, ws_i: B 4n ints); the exported entry
// mh_lcp_solve_batch_dev is the unmasked case
MH_HIDDEN int mh_lcp_solve_dev_masked(void* stream, int kind, int B, int n,
                                      const double* M, int ld, long strideM,
                                      const double* q, double* z,
                                      const int* z_size_in, int* z_size_out,
                                      uint32_t* rng, int* status, unsigned* pivots,
                                      int32_t* trace, int trace_cap, int* trace_len,
                                      const mh_lcp_opts* opts, const int* run_if, double* ws_d, int* ws_i,
                                      const int* n_arr,    // n_arr: per-problem sizes (<= n, M compact with ld = its n) or NULL
                                      double* work = nullptr,    // B x MH_WORK doubles or NULL: += the 2/3 k^3 flops / 8 k^2 bytes of every factorisation (n > 64), issued flops, ticks
                                      int wave_only = 0,         // 1: only the problems of at most 64 rows (n_arr) -- the caller runs the others itself; 2: the block solver in its
                                                                 // narrow geometry (the caller has other workgroups on the chip for it to share the CUs with)
                                      int* started = nullptr,    // the lcp_fast kinds, n > 64 (core_solve_round, mh_impact.hip): B + 2 ints, zeroed by the caller --
                                                                 // started[0] += 1 when a workgroup begins (the gate waits for B), started[2 + b] = 1 (failed) / 2 (solved)
                                                                 // when problem b is through, then started[1] += 1 (the ladder's tasks are handed out by these verdicts)
                                      int ordered = 0);          // 1: started[B + 2 ..] holds a permutation of the problems: workgroup i takes problem started[B + 2 + i]
extern MH_HIDDEN int mh_g_debug_repeats;             // mh_debug_set(5, v)
extern MH_HIDDEN int mh_g_debug_sched;               // mh_debug_set(7, v)
extern MH_HIDDEN int mh_g_debug_lpt;                 // mh_debug_set(11, v): longest-processing-time-first launch order on a full chip (core_solve_round)
extern MH_HIDDEN int mh_g_debug_reuse;               // mh_debug_set(6, v)
extern MH_HIDDEN int mh_g_debug_compact;             // mh_debug_set(3, v)
extern MH_HIDDEN int mh_g_debug_tasks;               // mh_debug_set(4, v): the Lemke ladder of the island pipeline as (world, attempt) tasks (1, default) or in sequence (0)

// the workgroup-per-problem LCP solver, one translation unit per thread geometry (mh_lcp_blk.hip: 256 threads, mh_lcp_blkw.hip: 1024)
namespace mh { struct LcpParams; struct Pow10Table; }
// per-problem work counters of the block solver (mh_*_batch_lu_work): [0] model flops, [1] model bytes (one dgesv per pivot),
// [2] flops the factorisation routines issue, [3] wall-clock ticks (device) / seconds (as returned to the caller) spent on the problem
#define MH_WORK 4
// task mode of the block solver's lcp_lemke kinds: bit 30 of a task's z_size_out = the attempt left through lcp_lemke's trivial exit
// (LCP.cpp:578) and would not have drawn from rand() whatever z.size() it was entered with (k_ladder_select)
#define MH_TASK_NODRAW 0x40000000
#define MH_LCP_BLOCK_LAUNCH_ARGS void* stream, int kind, int B, int n, const double* M, int ld, long strideM, const double* q, double* z, \
  const int* zsz_in, int* zsz_out, uint32_t* rng, int* status, unsigned* pivots, int32_t* trace, int trace_cap, int* trace_len, \
  const mh::LcpParams* P, const mh::Pow10Table* p10, double* wsd, int* wsi, const int* run_if, const int* n_arr, int flags, double* work, int task_worlds, int* solved_at
// LU workspaces of a TASK launch (flags & 8 or & 64: persistent workgroups, one workspace each): an upper bound on the workgroups any geometry keeps resident --
// up to eight one-wave problems per CU for n <= 512 (mh_lcp_blk1.hip), two above (mh_lcp_blk.hip / blky; blkw / blkx hold one)
static inline long mh_task_slots(int n) { return (long)((n <= 512) ? 8 : 2) * mh_cu_count(); }
MH_HIDDEN hipError_t mh_launch_lcp_blk(MH_LCP_BLOCK_LAUNCH_ARGS);
MH_HIDDEN hipError_t mh_launch_lcp_blkw(MH_LCP_BLOCK_LAUNCH_ARGS);
MH_HIDDEN hipError_t mh_launch_lcp_blkx(MH_LCP_BLOCK_LAUNCH_ARGS);   // the lcp_lemke kinds with 1024 < n <= 2048: 1024 threads, two rows per lane (mh_lcp_blkx.hip)
MH_HIDDEN hipError_t mh_launch_lcp_blky(MH_LCP_BLOCK_LAUNCH_ARGS);   // the lcp_lemke kinds with 512 < n <= 1024 and more tasks than CUs: 256 threads, four rows per lane (mh_lcp_blky.hip)
#define MH_BLKY_MIN_TASKS_PER_CU 2
#define MH_BLKX_MIN_N 1025
#define MH_BLKX_MAX_N 2048
#define MH_BLK2_MIN_PER_CU 4      /* problems (worlds of a ladder launch) per CU from which the lcp_lemke kinds take the 128-thread geometry: four problems share
                                     a CU there (measured, 16-box stacks: 363 k pivots/s on a full chip against 262 k with 256 threads at two per CU and 323 k
                                     with the round-3 right-looking LU at three; below that the 256-thread geometry finishes a problem sooner) */
#define MH_BLK1_MIN_PER_CU 1000000   /* problems per CU from which the one-wave geometry is chosen by itself (tuned by measurement; off until then) */
MH_HIDDEN hipError_t mh_launch_lcp_blk1(MH_LCP_BLOCK_LAUNCH_ARGS);
MH_HIDDEN hipError_t mh_launch_lcp_blk2(MH_LCP_BLOCK_LAUNCH_ARGS);   // two wavefronts per problem (lcp_lemke kinds, n <= 512, large batches)   // one wavefront per problem (lcp_lemke kinds, n <= 512, large batches)

// the three size variants of the many-worlds kernel, one translation unit each (mh_world_{small,wheel,large}.hip)
typedef void (*mh_world_kernel)(const mh_scene*, int, double, int, double*, mh_world_aux*, double*, int, double*, int, unsigned long long*, const int*);
struct mh_world_variant {
  mh_world_kernel kernel;
  int ph_count;                                        // per-phase cycle accumulators of the profiling launch
  hipError_t (*upload_tables)(const void* fric, size_t fric_bytes, const void* pow10, size_t pow10_bytes);
};
MH_HIDDEN const mh_world_variant* mh_world_variant_small();
MH_HIDDEN const mh_world_variant* mh_world_variant_wheel();
MH_HIDDEN const mh_world_variant* mh_world_variant_large();
MH_HIDDEN const mh_world_variant* mh_world_variant_small_prof();     // the same kernels with the phase profiler's stamps compiled in (mh_world_*_prof.hip)
MH_HIDDEN const mh_world_variant* mh_world_variant_wheel_prof();
MH_HIDDEN const mh_world_variant* mh_world_variant_large_prof();

// mh_debug_set keys 1 / 2 (test hooks; defined in mh_capi.hip)
extern MH_HIDDEN int mh_g_debug_ka;
extern MH_HIDDEN int mh_g_debug_blk;
extern MH_HIDDEN int mh_g_debug_fastgeom;
extern MH_HIDDEN int mh_g_debug_artic_pack;          // mh_debug_set(9, v): the articulated stepper with two worlds per wavefront (k_artic_step_p2)
extern MH_HIDDEN int mh_g_debug_reglu;               // mh_debug_set(10, v): lcp_fast's register-resident dense LU (mh_lu_reg.inc)

}  // extern "C"
