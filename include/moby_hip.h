/* moby_hip.h -- C ABI of libmoby_hip.so, the MI355X (gfx950) many-worlds
 * contact-dynamics core that stands in for Moby's LCP / impact-handling hot
 * path.  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * The reference has no plugin boundary around this path; each
 * entry point below names the C++ seam of /root/reference it replaces.
 * INTEGRATION.md shows the reference-side adapter (a Moby::LCP-shaped C++
 * class, moby_amd/cpp/MobyHipLCP.h) that a maintainer links instead of
 * src/LCP.cpp.
 *
 * Conventions
 *   - all matrices are column-major doubles (Ravelin::MatrixNd::data()/
 *     leading_dim()), all vectors contiguous doubles;
 *   - functions return MH_OK (0) or a negative MH_ERR_* code and never throw;
 *     mh_last_error() returns a thread-local message for the last failure;
 *   - `_dev` entry points take DEVICE pointers and enqueue on `stream`
 *     (a hipStream_t, NULL = default stream) without synchronising;
 *     the unsuffixed ones take HOST pointers, copy, run and synchronise;
 *   - per-problem outputs: status[b] = 1 where the reference solver would
 *     return true, 0 where it would return false (it never throws,
 *     include/Moby/LCP.h:21-27);
 *   - every world carries its own libc rand() stream (32 uint32 words,
 *     mh_rand_seed(state, 1) == a fresh process of the reference).
 */
#ifndef MOBY_HIP_H
#define MOBY_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_INVALID_ARG   (-1)
#define MH_ERR_UNSUPPORTED_N (-2)  /* n outside the range the kernels cover */
#define MH_ERR_HIP           (-3)  /* a HIP runtime call failed */
#define MH_ERR_NO_DEVICE     (-4)

#define MH_RAND_WORDS 32
#define MH_LCP_MAX_N_WAVE 64       /* one-wavefront-per-problem solver (M in LDS); also the many-worlds limit */
#define MH_LCP_MAX_N_BLOCK 4096    /* n above MH_LCP_MAX_N_WAVE: one 256-thread workgroup per problem, M read in
                                      place from HBM, LU scratch in an HBM workspace of B (n^2 + 5n) doubles */

/* solver selector: include/Moby/LCP.h:21-27 */
#define MH_LCP_FAST       0  /* LCP::lcp_fast              src/LCP.cpp:41   */
#define MH_LCP_FAST_REG   1  /* LCP::lcp_fast_regularized  src/LCP.cpp:212  */
#define MH_LCP_LEMKE      2  /* LCP::lcp_lemke (dense)     src/LCP.cpp:545  */
#define MH_LCP_LEMKE_REG  3  /* LCP::lcp_lemke_regularized src/LCP.cpp:353  */

/* trace encoding (optional pivot trace, one int32 stream per problem)
 *   lcp_fast : +(i+1) variable i moved basic->nonbasic, -(i+1) the reverse
 *   lcp_lemke: (entering+1), (leaving+1) per pivot
 *   0x40000000|k : start of attempt k of a regularised wrapper (0 = lambda 0) */
#define MH_TRACE_ATTEMPT 0x40000000

typedef struct mh_lcp_opts {
  int      min_exp;   /* regularised wrappers: first exponent (default -20)   */
  unsigned step_exp;  /* exponent step (lcp_fast_regularized default 4)       */
  int      max_exp;   /* exclusive upper exponent                             */
  double   piv_tol;   /* <= 0: solver default (LCP.cpp:761)                   */
  double   zero_tol;  /* <= 0: solver default (LCP.cpp:58,228,570)            */
} mh_lcp_opts;

/* library / device ------------------------------------------------------- */
/* 100 * major + minor.  101: mh_impact_batch_lu_work / mh_big_batch_lu_work fill B x 4 doubles per call (100: B x 2) -- a caller built against the
 * two-column layout must check for >= 101 and size its buffer accordingly (#define MH_VERSION is what this header describes). */
#define MH_VERSION 101
int         mh_version(void);
const char* mh_last_error(void);
int         mh_device_count(void);
/* Devices.  One process may drive every GPU of a node: worlds are independent, so a batch of B worlds is split
 * into one batch per device and nothing but per-interval counters crosses devices (moby_amd/cpp/example_multi_gpu.cpp: one
 * batch + stream per device, the counters reduced with RCCL's ncclAllReduce).
 *   - a batch (mh_world_batch, mh_big_batch, mh_impact_batch, mh_artic_batch) lives on the device that is CURRENT in the calling
 *     thread when its `create` runs (mh_device_set, or hipSetDevice in a HIP program); `mh_*_batch_device` returns it;
 *   - every later entry point that takes the batch runs on THAT device, whatever device the calling thread has current at the
 *     time: the library switches to the batch's device for the call and restores the caller's before it returns;
 *   - a `stream` argument must belong to the batch's device (HIP refuses the launch otherwise: MH_ERR_HIP); device pointers
 *     handed to a batch's entry points (traj_dev, ids_dev, `device_ptrs` results) are that device's;
 *   - the stateless entries (mh_lcp_solve_batch[_dev], mh_world_step_batch, mh_impact_process_batch) run on the current device. */
int         mh_device_get(void);        /* the calling thread's current device, or MH_ERR_NO_DEVICE */
int         mh_device_set(int device);  /* 0 <= device < mh_device_count(); MH_ERR_INVALID_ARG otherwise, MH_ERR_NO_DEVICE without a GPU */

/* glibc srand(seed) state for one world (host helper, no GPU needed) */
void        mh_rand_seed(uint32_t* state32, uint32_t seed);
/* next rand() of that stream (host helper; used by adapters and tests) */
int         mh_rand_next(uint32_t* state32);

/* B independent dense LCPs  w = M z + q, w,z >= 0, w'z = 0  of equal size n.
 * Replaces Moby::LCP::{lcp_fast, lcp_fast_regularized, lcp_lemke,
 * lcp_lemke_regularized} (include/Moby/LCP.h:21-27; callers
 * src/ImpactConstraintHandlerQP.cpp:219,224, src/ConstraintStabilization.cpp:954-955,
 * src/ImpactConstraintHandler.cpp:1239-1281).
 *
 *   M       B matrices, problem b at M + b*strideM, leading dimension ld >= n
 *   q       B*n
 *   z       B*n, in/out: warm start in (only read where z_size_in[b] == n),
 *           solution out
 *   z_size_in   B ints or NULL (= all n): z.size() on entry.  lcp_fast
 *           warm-starts iff it equals n (LCP.cpp:65); lcp_lemke draws n rand()
 *           values iff it differs from n (LCP.cpp:564-567,611-621)
 *   z_size_out  B ints or NULL: z.size() on return (2n after some Lemke
 *           failures, LCP.cpp:840-903)
 *   rng     B*32 words in/out (mh_rand_seed)
 *   status  B ints; pivots B unsigned (LCP::pivots) or NULL
 *   trace   B*trace_cap int32 or NULL; trace_len B ints or NULL
 */
int mh_lcp_solve_batch_dev(void* stream, int kind, int B, int n,
                           const double* M, int ld, long strideM,
                           const double* q, double* z,
                           const int* z_size_in, int* z_size_out,
                           uint32_t* rng, int* status, unsigned* pivots,
                           int32_t* trace, int trace_cap, int* trace_len,
                           const mh_lcp_opts* opts);

int mh_lcp_solve_batch(int kind, int B, int n,
                       const double* M, int ld, long strideM,
                       const double* q, double* z,
                       const int* z_size_in, int* z_size_out,
                       uint32_t* rng, int* status, unsigned* pivots,
                       int32_t* trace, int trace_cap, int* trace_len,
                       const mh_lcp_opts* opts);


/* ------------------------------------------------------------------------
 * Many-worlds stepping: B independent instances of one scene topology.
 * Replaces, per world, TimeSteppingSimulator::step (src/TimeSteppingSimulator.cpp:
 * 52-222), ConstraintSimulator::{broad_phase, calc_pairwise_distances,
 * find_unilateral_constraints, calc_impacting_unilateral_constraint_forces}
 * (src/ConstraintSimulator.cpp:298-537), ImpactConstraintHandler::
 * process_constraints (src/ImpactConstraintHandler.cpp:75) with the Drumwright-
 * Shell QP->LCP model (src/ImpactConstraintHandlerQP.cpp:94-497) and
 * ConstraintStabilization::stabilize (src/ConstraintStabilization.cpp:167).
 *
 * Scene scope of this build: up to MH_MAX_BODIES free rigid bodies with sphere,
 * box (against the plane only) or rimless-wheel spokes geometry plus one static
 * plane (the closed-form pairs of CCD.inl:804-886, 1164-1207 and of example/
 * rimless-wheel/coldet-plugin.cpp), gravity, per-pair ContactParameters; impact
 * models: Drumwright-Shell QP->LCP and, for islands whose contacts all have
 * mu-coulomb >= 100, the no-slip model (ImpactConstraintHandler.cpp:1009-1417).
 * Body ids are 0..nb-1 in the order the reference sorts them (by id string); the
 * ground plane, when present, has id nb.  Pair p enumerates (i<j) lexicographically.
 */
#define MH_MAX_BODIES 8
#define MH_MAX_PAIRS  36            /* C(MH_MAX_BODIES + 1, 2) */
#define MH_BODY_STATE 13            /* x(3) quat xyzw(4) v(3) omega(3), world axes, at the COM */
#define MH_GEOM_SPHERE 0
#define MH_GEOM_SPOKES 1            /* rimless wheel: N point "spoke tips" at radius R in the body's x-z plane
                                       (example/rimless-wheel/coldet-plugin.cpp:104-137, params.h:4-6); it is
                                       only ever tested against the ground plane, and always (plugin :53-74) */
#define MH_GEOM_BOX 2               /* BoxPrimitive, tested against the ground plane only (vertex-plane contacts,
                                       CCD.inl:848-886); box-box / box-sphere pairs must be disabled */
#define MH_GEOM_PIN 3               /* a body point (geom_dim, body frame) held at the global origin by six frictionless contacts with
                                       normals +-y, +-z, +-x: the collision plugin of example/contact-constrained-pendulum
                                       (contact-constrained-pendulum-coldet-plugin.cpp:53-146).  Large-world stepper (moby_hip_stack.h)
                                       and oracle only; the pair (body, static world) is always a candidate */
#define MH_MAX_SPOKES 8
#define MH_NOSLIP_MAX 16            /* largest no-slip LCP (contacts of one island) whose warm start is kept */

typedef struct mh_scene {
  int    nb;                               /* enabled rigid bodies */
  int    has_ground;                       /* static Plane primitive present */
  int    geom_type[MH_MAX_BODIES];         /* MH_GEOM_* */
  double geom_dim[MH_MAX_BODIES][3];       /* sphere: radius,-,- ; spokes: R, number of spokes, - ; box: xlen, ylen, zlen */
  double mass[MH_MAX_BODIES];
  double inertia[MH_MAX_BODIES][3];        /* body-frame principal inertia (SpherePrimitive.cpp:138-155) */
  double plane_R[9];                       /* row-major rotation of the plane frame; its +Y is the normal (PlanePrimitive) */
  double plane_o[3];
  double gravity[3];                       /* GravityForce accel (GravityForce.cpp:33-69) */
  int    pair_enabled[MH_MAX_PAIRS];       /* 0 = <DisabledPair> */
  double cp_epsilon[MH_MAX_PAIRS];         /* ContactParameters.cpp:98-135 */
  double cp_mu_coulomb[MH_MAX_PAIRS];
  double cp_mu_viscous[MH_MAX_PAIRS];
  double cp_compliance[MH_MAX_PAIRS];
  int    cp_nk[MH_MAX_PAIRS];              /* friction-cone-edges (>= 4) */
  double min_step_size;                    /* TimeSteppingSimulator.cpp:48  (sqrt eps) */
  double contact_dist_thresh;              /* ConstraintSimulator.cpp:56    (1e-6)     */
  double cstab_eps;                        /* ConstraintStabilization.cpp:59 (sqrt eps) */
  unsigned cstab_max_iterations;           /* ConstraintStabilization.cpp:56 (reference: UINT_MAX; here MH_CSTAB_DEFAULT_MAX_ITERATIONS) */
  int    lcp_n_max;                        /* largest LCP the wave solver must hold in LDS (0 = 64); worlds
                                              that exceed it get MH_WORLD_UNSUPPORTED */
} mh_scene;

/* status bits of mh_world_aux.status */
#define MH_WORLD_OK            0
#define MH_WORLD_LCP_FAILED    1   /* LCPSolverException (ImpactConstraintHandlerQP.cpp:225) */
#define MH_WORLD_IMPACT_TOL    2   /* ImpactToleranceException (warned only, ConstraintSimulator.cpp:342) */
#define MH_WORLD_UNSUPPORTED   4   /* island larger than the wave solver covers / unsupported model */
#define MH_WORLD_STAB_FAILED   8   /* update_q gave up (ConstraintStabilization.cpp:230-234) */
#define MH_WORLD_STALLED       16   /* > 100000 zero-length mini-steps in one step, or MH_CSTAB_HARD_CAP stabilisation
                                       iterations in one call (the reference would not return) */
#define MH_CSTAB_HARD_CAP 10000u
/* Default of mh_scene.cstab_max_iterations in EVERY entry path (mh_scene_defaults, the XML loader when the attribute
 * constraint-stabilization-max-iterations is absent, moby_amd/scene.py).  The reference's default is UINT_MAX
 * (ConstraintStabilization.cpp:56); with the restated arithmetic its loop 2-cycles on resting stacks (DESIGN.md 2,
 * deviation 1) and would not return, so the documented default is a cap the loop reaches about once in 1000 steps.
 * A scene that sets the attribute explicitly (e.g. ur10.xml: 0) is honoured as written. */
#define MH_CSTAB_DEFAULT_MAX_ITERATIONS 10u
#define MH_CA_HARD_CAP 10000000u     /* conservative-advancement sub-steps of one mini-step before MH_WORLD_STALLED */

/* persistent per-world solver state (what the reference keeps in the
 * simulator / handler / libc between steps) + counters */
typedef struct mh_world_aux {
  uint32_t rng[MH_RAND_WORDS];     /* libc rand() stream                                   */
  double   time;                   /* Simulator::current_time                              */
  double   zlast[MH_LCP_MAX_N_WAVE];  /* ImpactConstraintHandler::_zlast (ICH-QP:158-162,233) */
  double   zbuf[MH_LCP_MAX_N_WAVE];   /* storage of ImpactConstraintHandler::_z            */
  int      zlast_size;
  int      zbuf_size;              /* _z.size()                                            */
  int      zbuf_cap;               /* entries of zbuf ever written (Ravelin keeps them)    */
  int      status;                 /* MH_WORLD_* bits, sticky                              */
  double   vns[MH_NOSLIP_MAX];     /* ImpactConstraintHandler::_v, the no-slip LCP's z (ICH:1239) */
  int      vns_size;               /* _v.size()                                            */
  int      pad0;
  unsigned long long steps;        /* step() calls                                         */
  unsigned long long mini_steps;   /* do_mini_step calls                                   */
  unsigned long long lcp_solves;   /* impact + stabilisation LCPs solved                   */
  unsigned long long lcp_rows;     /* sum of their dimensions (BASELINE metric "rows")     */
  unsigned long long lcp_pivots;
  unsigned long long stab_iters;
  unsigned long long lcp_alg_bytes;/* sum of 8 (n^2 + 2n): bytes the same solves move through the LCP entry */
  unsigned long long stab_rows;    /* the part of lcp_rows that ConstraintStabilization::determine_dq solved (impact rows = lcp_rows - stab_rows) */
} mh_world_aux;

/* test hooks: key 1 = edge of the LDS-resident LU block of the world kernel (0..64, clamped to what the
 * kernel variant holds: 12 or 16; 0 sends every LU factorisation through the HBM workspace path) */
int  mh_debug_set(int key, int value);   /* key 2: block LCP solver (n > 64) thread geometry -- 0 choose by n and B,
                                            1 = 256 threads per problem, 2 = 1024 threads per problem, 3 / 4 = 64 / 128 threads per problem
                                            (the lcp_lemke kinds with n <= 512 only; the lcp_fast kinds keep the choice of 0);
                                            key 3: lcp_lemke's bases (n > 64) through the structure-exploiting LU (1, default) or the dense one (0);
                                            key 4: the Lemke ladder of the island pipeline in sequence (0), as (world, attempt) tasks (1),
                                                   tasks started beside lcp_fast when n >= 256 (2); 3 (default) = 2, and when the batch fills the chip
                                                   with LCPs of 384..512 rows the tasks follow lcp_fast's launch on a second stream, behind a gate that
                                                   opens once its last workgroup has started; 4 = 3 for LCPs of any size;
                                            key 5: lcp_fast (n > 64) skips the repetitions of a repeating pivot sequence (1, default) or
                                                   runs every iteration (0);
                                            key 6: the structure-exploiting LU reuses the factors of the unchanged leading columns from one Lemke
                                                   pivot to the next (1, default) or factorises every basis from scratch (0).
                                            key 7: ladder tasks started after lcp_fast are handed out by need (1, default) or by block index (0).
                                            key 8: thread geometry of the lcp_fast kinds alone for n <= 512 (0 choose, 1-4 as key 2);
                                            key 9: fixed-base CRB articulated bodies without spheres / stabiliser stepped two worlds per wavefront (1; default 0:
                                                   measured no faster, profiles/r04_a_artic_issue.json).
                                            key 10: lcp_fast in the 1024-thread geometry solves a nonbasic system of up to 191 rows in the registers of
                                                   its sixteen waves (1, default) or through the HBM workspace (0).
                                            None of the switches changes a result */
void mh_scene_defaults(mh_scene* s);   /* zero + the reference's default tolerances */
void mh_world_aux_init(mh_world_aux* a, uint32_t seed);

/* A batch of B worlds resident on the GPU (device copies of the scene, the
 * body states and the per-world solver state).  This is what a batched
 * TimeSteppingSimulator subclass holds instead of B simulator objects.
 *   create    allocates device memory, uploads the scene, initialises every
 *             world's aux record (rand() stream at srand(1))
 *   upload    state: B * nb * MH_BODY_STATE doubles (host); aux: B records or NULL
 *   step      enqueues ONE launch that advances every world by nsteps steps of dt
 *             (TimeSteppingSimulator::step called nsteps times) on `stream`
 *             (hipStream_t, NULL = default); does not synchronise.
 *             traj_dev: optional DEVICE buffer B * nsteps * nb * 7 doubles receiving
 *             the generalized coordinates (x, quat xyzw) after each step -- the rows
 *             programs/regress.cpp:82-93 prints -- or NULL
 *   download  synchronises the device and copies state / aux back (either may be NULL)
 *   device_ptrs  raw device pointers for zero-copy interop (e.g. torch tensors)
 */
typedef struct mh_world_batch mh_world_batch;
int mh_world_batch_create(const mh_scene* scene, int B, mh_world_batch** out);
int mh_world_batch_destroy(mh_world_batch* wb);
/* The per-interval reduction of a multi-GPU run, device side.  Enqueues on `stream` a pass over the batch's solver records
 * that leaves, in DEVICE memory, the MH_COUNTERS-element vectors a collective then reduces over the devices of the node (RCCL:
 * ncclAllReduce(sums, ncclUint64, ncclSum) and (maxs, ncclUint64, ncclMax); moby_amd/cpp/example_multi_gpu.cpp):
 *   [0] steps  [1] lcp_rows  [2] lcp_pivots  [3] worlds with an error bit (MH_WORLD_IMPACT_TOL, a warning, not counted)
 *   [4] lcp_solves  [5] mini_steps  [6] stab_iters  [7] lcp_alg_bytes
 * sums = the totals over this batch's worlds, maxs = the largest value any one world holds (the slowest world of the interval). */
#define MH_COUNTERS 8
int mh_world_batch_counters_dev(mh_world_batch* wb, void* stream, unsigned long long* sums_dev, unsigned long long* maxs_dev);
int mh_world_batch_device(const mh_world_batch* wb);   /* the device the batch lives on (see Devices above) */
int mh_world_batch_upload(mh_world_batch* wb, const double* state, const mh_world_aux* aux);
int mh_world_batch_step(mh_world_batch* wb, void* stream, double dt, int nsteps, double* traj_dev);
/* step of a SUBSET: nsteps x step(dt) of the worlds ids_dev[0 .. count) (a DEVICE array of world indices) on `stream`.  Worlds are
 * independent: a batch may be split over streams, e.g. the few worlds whose solver chain runs to its pivot caps every
 * step in a launch of their own, so that the next interval of the others does not wait for them.  Concurrent launches must not
 * share a world, and the ids of ONE launch must be unique (two wavefronts on one world race); an id outside [0, B) is ignored. */
int mh_world_batch_step_ids(mh_world_batch* wb, void* stream, double dt, int nsteps, const int* ids_dev, int count);
int mh_world_batch_download(mh_world_batch* wb, double* state, mh_world_aux* aux);
int mh_world_batch_device_ptrs(mh_world_batch* wb, double** state_dev, mh_world_aux** aux_dev);
/* diagnostic build of the same launch with in-kernel s_memtime stamps: mean cycles per world spent in
 * each phase (order: broad phase + CA, position integration, forward dynamics, contact generation,
 * islands, problem data, LCP matrix build, LCP solve, impulse application, stabilisation); after the per-phase entries: the slowest
 * and the fastest world's total, the mean world's total and the total of the world at the 99th percentile (if nphase leaves room) */
int mh_world_batch_occupancy(mh_world_batch* wb);   /* diagnostic: resident workgroups per CU (runtime query) */
int mh_world_batch_profile(mh_world_batch* wb, double dt, int nsteps, double* phase_cycles, int nphase);
int mh_world_profile_phase_count(void);            /* entries per world of the stamped launch (the four totals follow them) */

/* Host convenience: create + upload + step + download (+ trajectory) + destroy. */
int mh_world_step_batch(const mh_scene* scene, int B, double dt, int nsteps,
                        double* state, mh_world_aux* aux, double* traj);

#ifdef __cplusplus
}
#endif
#endif
