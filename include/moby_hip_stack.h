/* moby_hip_stack.h -- C ABI of the many-worlds stepper for LARGE worlds (BASELINE config 4: box stacks,
 * tens of bodies, impact LCPs of 1000+ rows), seams B5 and B3 of SURVEY.md 8b for worlds the one-wavefront
 * kernel of moby_hip.h (<= 8 bodies, <= 64 LCP rows) does not hold.
 *
 * Replaces, per world,
 *   TimeSteppingSimulator::step / do_mini_step / calc_next_CA_Euler_step   src/TimeSteppingSimulator.cpp:52-222, 272-331
 *   ConstraintSimulator::{broad_phase, calc_pairwise_distances, find_unilateral_constraints,
 *     calc_impacting_unilateral_constraint_forces}                          src/ConstraintSimulator.cpp:298-537
 *   CCD::broad_phase, calc_CA_Euler_step_generic, calc_next_CA_Euler_step_generic / _polyhedron_plane,
 *     find_contacts_plane_generic                                           src/CCD.cpp:169-468, 702-876; include/Moby/CCD.inl:848-886
 *   ImpactConstraintHandler::process_constraints with its island loop       src/ImpactConstraintHandler.cpp:75-168, 530-626
 *   ConstraintStabilization::stabilize (+ compute_problem_data, determine_dq, update_q, ridders_unilateral)
 *                                                                           src/ConstraintStabilization.cpp:167-254, 347-492, 932-970, 1056-1379
 *   Simulator::calc_fwd_dyn for free bodies, and for islands of bodies tied by implicit joints the KKT solve of
 *     Simulator::solve with Simulator::find_islands                         src/Simulator.cpp:482-602, 608-805, 956-1045
 *   ImpactConstraintHandler::compute_X / get_full_rank_implicit_constraints (used by the stabiliser)
 *                                                                           src/ImpactConstraintHandler.cpp:1590-1739
 * Every world is stepped by its own workgroup (geometry, islands, problem data, impulse application, Ridders line
 * search) and its LCPs go through the LCP entry of moby_hip.h with per-island sizes; nothing runs on the host but
 * the loop that asks "does any world have another mini-step / island / stabilisation iteration".
 *
 * Geometry scope: boxes and spheres against ONE static plane (the closed forms of moby_hip.h), plus box-on-box pairs
 * declared MH_PAIR_VERTEX_FACE: the pair is handled exactly like box-plane (CCD.inl:848-886, PlanePrimitive.cpp:338-376,
 * CCD.cpp:383-397, 410-468), the plane being the +Y face of the lower-id box ("support") and the polyhedron the
 * higher-id box, with the polyhedron's velocity taken relative to the support.  The reference's box-box path (v-clip on
 * a qhull polyhedron, contact order qhull-dependent: SURVEY a18) is not reproducible and is not built; the model
 * is the build's documented order for stacked pairs and is valid while the upper box's vertices project inside the
 * support face (the stack generator shrinks boxes with height).  Any other box-box / box-sphere pair must be left out of
 * the candidate list (= a <DisabledPair>).
 */
#ifndef MOBY_HIP_STACK_H
#define MOBY_HIP_STACK_H
#include "moby_hip.h"
#include "moby_hip_impact.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MH_PAIR_CLOSED_FORM 0   /* sphere-sphere, sphere-plane, box-plane */
#define MH_PAIR_VERTEX_FACE 1   /* box (higher id) on the +Y face of a box (lower id) */

#define MH_IJOINT_SPHERICAL 0     /* 3 equations (Ravelin::SphericalJointd)                              */
#define MH_IJOINT_REVOLUTE  1    /* 5: the axis a_0 = a_1 (inboard) stays orthogonal to b_0, b_1 (outboard) */
#define MH_IJOINT_FIXED     2     /* 6: a_k . b_k = 0 for three pairs of orthogonal axes                  */
#define MH_IJOINT_PLANAR    3     /* 3 (Ravelin::PlanarJointd): ONE position row along the plane normal a_2 (inboard frame),
                                     then a_0 . b_0 = 0, a_1 . b_1 = 0 with a_0, a_1 the in-plane directions (inboard) and
                                     b_0 = b_1 the normal in the outboard frame: the body slides on the plane and turns about its normal */
#define MH_IJOINT_UNIVERSAL 4     /* 4 (Ravelin::UniversalJointd): the joint point + a_0 . b_0 = 0 for the two axes of the cross,
                                     a_0 fixed in the inboard frame, b_0 in the outboard frame */
#define MH_IJOINT_PRISMATIC 5     /* 5 (Ravelin::PrismaticJointd): TWO position rows along a_0, a_1 (inboard frame, orthogonal to
                                     the sliding axis a_2), then a_k . b_k = 0, k = 0, 1, 2 (b = the outboard images of a_1, a_2, a_0):
                                     the outboard link slides along the axis without turning */
#define MH_IJOINT_MAX_BODIES 16   /* bodies of one jointed island (forward-dynamics KKT system of up to 96 coordinates) */
#define MH_IJOINT_MAX_JOINTS 16   /* joints of one jointed island                                           */
#define MH_IJOINT_MAX_EQNS   48   /* constraint equations of one jointed island (J iM J' and its factor live in LDS) */

#define MH_BIG_MAX_JOINTS 64      /* implicit joints per scene                                               */
#define MH_BIG_MAX_JOINT_ROWS (6 * MH_BIG_MAX_JOINTS)

#define MH_BIG_MAX_BODIES 128
#define MH_BIG_MAX_PAIRS  256
#define MH_BIG_MAX_CONTACTS 512   /* per world, impact or stabilisation list */

/* A scene of any size: body tables + the candidate pairs (everything not listed is a <DisabledPair>).  Pairs are
 * (a < b) body ids, the ground plane is id nb, sorted lexicographically (the canonical order of SURVEY 7 / a18).
 * All contacts of a scene share one friction-cone-edges value nk (so a world's LCP size is 6 nc + nc nk/2). */
typedef struct mh_big_scene {
  int nb, has_ground;
  const int*    geom_type;      /* nb: MH_GEOM_SPHERE / MH_GEOM_BOX                 */
  const double* geom_dim;       /* nb x 3                                            */
  const double* mass;           /* nb                                                */
  const double* inertia;        /* nb x 3 body-frame principal moments               */
  double plane_R[9], plane_o[3], gravity[3];
  int npairs;
  const int*    pair_a;         /* npairs                                            */
  const int*    pair_b;
  const int*    pair_model;     /* MH_PAIR_*                                         */
  const double* cp_epsilon;     /* npairs each: ContactParameters                    */
  const double* cp_mu_coulomb;
  const double* cp_mu_viscous;
  const double* cp_compliance;
  int    nk;                    /* friction-cone-edges of every pair (even, >= 4)    */
  double min_step_size, contact_dist_thresh, cstab_eps;
  unsigned cstab_max_iterations;
  int    lcp_n_max;             /* capacity of the handlers' LCPs (0 = as large as the pair list allows, <= MH_LCP_MAX_N_BLOCK);
                                   an island that needs more flags its world MH_WORLD_UNSUPPORTED */
  int    impact_model;          /* MH_IMPACT_MODEL_DS (0, the reference's default build) or MH_IMPACT_MODEL_AP (its -DUSE_AP build):
                                   moby_hip_impact.h */
  /* Implicit (bilateral) joints between free bodies, or between a body and the static world (id nb): the simulator's
   * <ImplicitConstraint> list (ConstraintSimulator.cpp, Simulator::implicit_joints).  Forward dynamics of a jointed island
   * is the KKT solve of Simulator::solve (src/Simulator.cpp:608-805) and joints connect constraint islands
   * (src/UnilateralConstraint.cpp:993-1008).  A joint is 3 position rows (the joint point of both bodies coincides, global
   * axes; the planar joint: one row along its normal) plus 0 / 2 / 3 orientation rows "a_k . b_k = 0" with a_k fixed in the inboard and b_k in the outboard frame.
   * ConstraintStabilization closes the joints too (CStab:133-160, 197, 1132-1192): islands no contact touches take the
 * bilateral step alone (CStab:462-486, 531-700), contact islands that hold jointed bodies get the general
 * ImpactConstraintHandler::compute_X (X = iM - 2G + G'MG over the full-rank joint rows, ICH:1590-1739).  The impact handler
 * itself never sees joint rows -- the reference leaves island_ijoints empty there.  Sizes: MH_IJOINT_MAX_* per island. */
  int njoints;
  const int*    joint_type;       /* njoints: MH_IJOINT_*                                                          */
  const int*    joint_inboard;    /* body id, or nb for the static world                                           */
  const int*    joint_outboard;
  const double* joint_anchor_in;  /* njoints x 3: the joint point in the inboard body's frame (global if the world)  */
  const double* joint_anchor_out; /* njoints x 3                                                                    */
  const double* joint_vec_in;     /* njoints x 9: a_0, a_1, a_2 in the inboard frame                                */
  const double* joint_vec_out;    /* njoints x 9: b_0, b_1, b_2 in the outboard frame                               */
} mh_big_scene;

/* B worlds resident on the GPU.  mh_world_aux carries the rand() stream, time, status bits and counters; the
 * handlers' _zlast / _z vectors (up to lcp_n_max doubles each) live in the batch (solver_state below).
 *   step        nsteps x TimeSteppingSimulator::step(dt) for every world (seam B5); host loop + device kernels,
 *               returns when the device work is enqueued AND the last control read-back is done (the loop needs them)
 *   stabilize   ConstraintStabilization::stabilize alone on the resident states (seam B3,
 *               include/Moby/ConstraintStabilization.h:24): configurations change, velocities are restored
 */
typedef struct mh_big_batch mh_big_batch;
int mh_big_batch_create(const mh_big_scene* scene, int B, mh_big_batch** out);
int mh_big_batch_destroy(mh_big_batch* bb);
int mh_big_batch_device(const mh_big_batch* bb);   /* the device the batch lives on (moby_hip.h, Devices) */
/* upload: state (B x nb x 13) and / or aux (B records; NULL leaves them).  aux carries the rand() state, clocks, counters, flags and
 * the handler's warm-start SIZES (zlast_size, zbuf_size, zbuf_cap): a zeroed aux makes the next solve cold.  The vectors behind those
 * sizes are load_solver_state's: a resume is download + save_solver_state, then upload + load_solver_state. */
int mh_big_batch_upload(mh_big_batch* bb, const double* state, const mh_world_aux* aux);
int mh_big_batch_step(mh_big_batch* bb, void* stream, double dt, int nsteps);
int mh_big_batch_stabilize(mh_big_batch* bb, void* stream);
int mh_big_batch_download(mh_big_batch* bb, double* state, mh_world_aux* aux);
int mh_big_batch_lcp_capacity(const mh_big_batch* bb);
/* the handlers' solver chain priced as SURVEY 8(d) does (see mh_impact_batch_lu_work): B x 4 doubles */
int mh_big_batch_lu_work(mh_big_batch* bb, double* work, int reset);
/* checkpoint / resume: _zlast, _z (B x capacity each) and their sizes (B each: zlast_size, zbuf_size, zbuf_cap) */
int mh_big_batch_save_solver_state(mh_big_batch* bb, double* zlast, double* zbuf, int* sizes3);
int mh_big_batch_load_solver_state(mh_big_batch* bb, const double* zlast, const double* zbuf, const int* sizes3);

#ifdef __cplusplus
}
#endif
#endif
