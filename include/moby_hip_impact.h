/* moby_hip_impact.h -- C ABI of the batched impact handler (seam B2 of SURVEY.md 8b).
 *
 * Replaces, for B independent worlds that each hand over an explicit list of contact
 * constraints,
 *   ImpactConstraintHandler::process_constraints      include/Moby/ImpactConstraintHandler.h:47,
 *                                                      src/ImpactConstraintHandler.cpp:75-168
 *   ::apply_model_to_connected_constraints             src/ImpactConstraintHandler.cpp:530-626
 *   ::compute_problem_data / add_contact_*_to_Jacobian src/ImpactConstraintHandler.cpp:1817-2166
 *   ::solve_qp_work / setup_QP (Drumwright-Shell)      src/ImpactConstraintHandlerQP.cpp:94-497
 *   ::update_from_stacked, apply_restitution,
 *     update_constraint_velocities_from_impulses        src/ImpactConstraintHandler.cpp:298-491
 * for islands of ANY size the LCP entry covers (n = 6 nc + nc nk/2 <= MH_LCP_MAX_N_BLOCK): the caller
 * is Moby's ConstraintSimulator::calc_impacting_unilateral_constraint_forces
 * (src/ConstraintSimulator.cpp:298-355), which owns the contact list -- so contact generation for
 * pairs the many-worlds stepper does not cover (box-box, box-sphere: v-clip on a qhull polyhedron)
 * stays with the caller.  This is the path of BASELINE config 4 (box stacks: nc ~ 200-256, n ~ 2048).
 *
 * Scope of this build: free rigid bodies; contacts only (no joint limits, no bilateral rows).  A world's contacts may
 * form any number of islands (UnilateralConstraint::determine_connected_constraints, src/UnilateralConstraint.cpp:940-1194);
 * they are processed one after the other as ImpactConstraintHandler.cpp:105-151 does, each with the Drumwright-Shell model
 * or -- when every contact of the island has mu >= 100 -- the no-slip model (ImpactConstraintHandler.cpp:134-135, 236-295,
 * 1009-1417; islands of up to MH_NOSLIP_MAX contacts, larger ones are flagged MH_WORLD_UNSUPPORTED -- never approximated).
 */
#ifndef MOBY_HIP_IMPACT_H
#define MOBY_HIP_IMPACT_H
#include "moby_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* one UnilateralConstraint of type eContact, as CollisionDetection::create_contact
 * (src/CollisionDetection.cpp:57-95) and ConstraintSimulator::preprocess_constraint
 * (src/ConstraintSimulator.cpp:390-416) leave it */
typedef struct mh_contact {
  double point[3];        /* contact_point, world frame                                   */
  double normal[3];       /* contact_normal, unit, points from body2 toward body1          */
  int    body1, body2;    /* contact_geom1 / contact_geom2 body: 0..nb-1, anything else = static */
  double mu_coulomb, mu_viscous, epsilon, compliance;   /* ContactParameters              */
  int    nk;              /* friction-cone-edges (NK, even, >= 4)                          */
  int    pad;
} mh_contact;

/* B worlds x (nb bodies, nc contacts) resident on the GPU, plus what the reference's handler object
 * keeps between calls: _zlast / _z (ImpactConstraintHandlerQP.cpp:158-162, 233) and the rand() stream.
 *   create    mass: nb, inertia: nb x 3 body-frame principal moments (shared by all worlds); every
 *             contact of every call must use the same nk (so that all LCPs have the same n)
 *   upload    state: B*nb*MH_BODY_STATE, contacts: B*nc (host)
 *   process   enqueues process_constraints for every world on `stream`; does not synchronise
 *   download  synchronises; state (velocities changed), impulses B*nc*3 = accumulated (cn, cs, ct) of each
 *             contact in the caller's contact order, status B (MH_WORLD_* bits, sticky), pivots B
 *             (sum of LCP::pivots over the call's solves), solves B (LCPs solved in the last call);
 *             any pointer may be NULL
 */
typedef struct mh_impact_batch mh_impact_batch;
int mh_impact_batch_create(int B, int nb, int nc, int nk, const double* mass, const double* inertia,
                           mh_impact_batch** out);
int mh_impact_batch_destroy(mh_impact_batch* ib);
int mh_impact_batch_device(const mh_impact_batch* ib);   /* the device the batch lives on (moby_hip.h, Devices) */
int mh_impact_batch_upload(mh_impact_batch* ib, const double* state, const mh_contact* contacts);
int mh_impact_batch_process(mh_impact_batch* ib, void* stream);
int mh_impact_batch_download(mh_impact_batch* ib, double* state, double* impulses, int* status,
                             unsigned* pivots, int* solves);
/* the dimension n of every world's impact LCP, and (debug / parity tests) device -> host copies of the
 * assembled _MM (B*n*n, column-major) and _qq (B*n) of the last call */
int mh_impact_batch_lcp_size(const mh_impact_batch* ib);
int mh_impact_batch_debug_lcp(mh_impact_batch* ib, double* MM, double* qq);
/* Measurement aid: per world, the work SURVEY 8(d) prices the solver chain at -- every factorisation of the workgroup-per-problem
 * solver (one per pivot: src/LCP.cpp:120 for lcp_fast's k x k nonbasic block, :837-838 for lcp_lemke's n x n basis) counted as a
 * dense dgesv, 2/3 k^3 flops over 8 k^2 bytes, whatever the kernels skip of it.  work: B x 4 doubles, accumulated since the batch
 * was created or last reset (reset != 0 zeroes the counters after the copy): [0] those model flops, [1] those model bytes, [2] the
 * flops the factorisation routines really issue (the structure-exploiting LU of Lemke's bases performs a small part of a dgesv), [3] the
 * seconds workgroups spent on the world's problems, every attempt of the Lemke ladder included (summed over a batch and divided by
 * resident workgroups x wall time: how busy the solver kept the chip).  LCPs of at most 64 rows are not counted. */
int mh_impact_batch_lu_work(mh_impact_batch* ib, double* work, int reset);
/* checkpoint / resume: what the handler keeps between calls -- _zlast (B*n), its size (B), the rand()
 * streams (B*MH_RAND_WORDS) and the sticky status bits (B).  The reference's XML pickle drops _zlast, so a resumed run
 * there diverges in its pivot sequence; a batch restored with load_solver_state continues bit for bit. */
int mh_impact_batch_save_solver_state(mh_impact_batch* ib, double* zlast, int* zlast_size, uint32_t* rng, int* status);
int mh_impact_batch_load_solver_state(mh_impact_batch* ib, const double* zlast, const int* zlast_size, const uint32_t* rng,
                                      const int* status);
/* ... and the no-slip model's warm start ImpactConstraintHandler::_v (src/ImpactConstraintHandler.cpp:1239): v is
 * B x MH_NOSLIP_MAX doubles, v_size B ints.  A batch whose islands never take the no-slip model keeps size 0. */
int mh_impact_batch_save_noslip_state(mh_impact_batch* ib, double* v, int* v_size);
int mh_impact_batch_load_noslip_state(mh_impact_batch* ib, const double* v, const int* v_size);
/* Which model the islands with finite friction take (src/ImpactConstraintHandler.cpp:139-146): the reference chooses at
 * BUILD time -- Drumwright-Shell (apply_model_to_connected_constraints, the default) or, with -DUSE_AP (CMakeLists.txt:19,
 * 77-79), Anitescu-Potra (src/ImpactConstraintHandlerLCP.cpp:36-370: LCP [UL UR; LL 0] of 5 nc + NK_DIRS rows,
 * lcp_lemke_regularized(-20, 1, -2) on a fresh z, impulses applied through the contacts' accumulated wrenches).  Here it
 * is a property of the batch; islands whose contacts all have mu >= 100 take the no-slip model either way. */
#define MH_IMPACT_MODEL_DS 0
#define MH_IMPACT_MODEL_AP 1
int mh_impact_batch_set_model(mh_impact_batch* ib, int model);
/* raw device pointers for zero-copy interop.  Contacts written straight to contacts_dev skip upload()'s host
 * checks; the device repeats them (nk of the batch, unit normal, two distinct bodies with at least one dynamic) and
 * flags a world with a malformed contact MH_WORLD_UNSUPPORTED instead of processing it.  The restitution round
 * (epsilon > 0, src/ImpactConstraintHandler.cpp:578-602) is decided on the device per world, not from upload(). */
int mh_impact_batch_device_ptrs(mh_impact_batch* ib, double** state_dev, mh_contact** contacts_dev);

/* Host convenience: create + upload + process + download + destroy (fresh handler state). */
int mh_impact_process_batch(int B, int nb, int nc, int nk, const double* mass, const double* inertia,
                            double* state, const mh_contact* contacts, double* impulses,
                            int* status, unsigned* pivots, int* solves);

#ifdef __cplusplus
}
#endif
#endif
