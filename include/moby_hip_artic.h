/* moby_hip_artic.h -- C ABI of the many-worlds stepper for ARTICULATED bodies (BASELINE config 5: the ur10 arm of
 * example/ur10/model.sdf x8192 initial states): reduced-coordinate forward dynamics by the composite-rigid-body
 * algorithm + Cholesky, and joint limits as unilateral constraints.
 *
 * Replaces, per world (one Moby::RCArticulatedBody with 1-DOF revolute / prismatic joints and a fixed base -- or a floating one carried by six virtual joints,
 * mh_artic_model.floating_base -- which is
 * what SDFReader::read_model builds: eCRB + eLinkCOM, src/SDFReader.cpp:934-935),
 *   TimeSteppingSimulator::step / do_mini_step                        src/TimeSteppingSimulator.cpp:52-222
 *   Ravelin::RCArticulatedBodyd::calc_fwd_dyn (CRB or FSAB, mh_artic_model.algorithm)  [seam B4]
 *                                                                      call site src/Simulator.cpp:552 -- Ravelin's source is
 *                                                                      NOT in the reference tree: the algorithm is
 *                                                                      Featherstone's (CRBA for H, RNEA for the bias), "parity unpinned"
 *   ArticulatedBody::find_limit_constraints                            include/Moby/ArticulatedBody.inl:9-43
 *   ImpactConstraintHandler::compute_problem_data / compute_X /
 *     compute_limit_components for limit rows                          src/ImpactConstraintHandler.cpp:1590-1695, 1755-1781
 *   ::apply_no_slip_model[_to_connected_constraints] with NC = 0       src/ImpactConstraintHandler.cpp:236-295, 1009-1417
 *     (an island without contacts has all_inf == true, ICH:123-135: LCP  L X L' l + L v >= 0  with X = H^-1)
 *   ::update_from_stacked, update_constraint_velocities_from_impulses, apply_restitution   src/ImpactConstraintHandler.cpp:298-525
 * One wavefront per world; H, its Cholesky factor, H^-1 and the limit LCP live in LDS.
 *
 * MH_WORLD_LCP_FAILED is the end of a world's run here as everywhere (moby_hip.h): the exception it stands for -- LCPSolverException, a generalized inertia that is not
 * positive definite, a failed compute_X -- is caught nowhere in the reference; the state stays where the throw left it and later steps pass the world over.
 *
 * Scope: config 5 is the robot alone (self-collision disabled as in ur10.xml:12: no contact rows, one mini-step per step);
 * optionally sphere primitives on links against a static plane (mh_artic_model.nspheres, no-slip contacts), no actuator torques (controller plugins stay on the host side of the seam: add them
 * through qdd = H^-1 (tau - C) by passing tau).  Constraint stabilisation with joint-limit rows and, for bodies with link spheres, contact
 * rows: mh_artic_model.cstab_max_iterations (ur10.xml:11 sets constraint-stabilization-max-iterations = 0 = off).
 */
#ifndef MOBY_HIP_ARTIC_H
#define MOBY_HIP_ARTIC_H
#include "moby_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MH_ARTIC_MAX_JOINTS 16
#define MH_JOINT_REVOLUTE 0
#define MH_JOINT_PRISMATIC 1
#define MH_ARTIC_CRB  0
#define MH_ARTIC_FSAB 1
#define MH_ARTIC_MAX_SPHERES 4    /* sphere primitives carried by links (contacts against the one static plane) */

/* Joint i carries link i; joints are listed parents first.  A link may be massless as long as its joint moves mass (something outboard of it has mass).  All quantities are LOCAL (constant), as a reader of
 * model.sdf derives them once at q = 0 (mh_io_load_sdf, moby_amd/host/mh_io.cpp):
 *   Rrel, trel   pose of link i's frame in its parent link's frame at q = 0 (row-major rotation; parent -1 = model frame)
 *   axis         joint axis in link i's frame, unit (revolute: rotation about it through the link origin; prismatic:
 *                translation along it)
 *   com, inertia centre of mass in link i's frame; inertia about the COM in link i's axes (row-major 3 x 3, symmetric) */
typedef struct mh_artic_model {
  int    nj;
  int    parent[MH_ARTIC_MAX_JOINTS];
  int    jtype[MH_ARTIC_MAX_JOINTS];
  double Rrel[MH_ARTIC_MAX_JOINTS][9];
  double trel[MH_ARTIC_MAX_JOINTS][3];
  double axis[MH_ARTIC_MAX_JOINTS][3];
  double com[MH_ARTIC_MAX_JOINTS][3];
  double inertia[MH_ARTIC_MAX_JOINTS][9];
  double mass[MH_ARTIC_MAX_JOINTS];
  double lolimit[MH_ARTIC_MAX_JOINTS];     /* Joint::lolimit / hilimit (src/Joint.cpp:201-262); +-DBL_MAX = none */
  double hilimit[MH_ARTIC_MAX_JOINTS];
  double limit_restitution[MH_ARTIC_MAX_JOINTS];   /* restitution-coeff, default 0 (src/Joint.cpp:33) */
  double gravity[3];
  int    algorithm;      /* RCArticulatedBody::algorithm_type (include/Moby/RCArticulatedBody.h: eFeatherstone / eCRB; the SDF
                            reader sets eCRB, src/SDFReader.cpp:934): MH_ARTIC_CRB (0) or MH_ARTIC_FSAB -- forward dynamics by
                            Featherstone's articulated-body recursion; the impact handler's X = H^-1 is the generalized inertia's
                            inverse either way (ICH:1600-1607) */
  int    floating_base;  /* RCArticulatedBody floating-base="true" (src/RCArticulatedBody.cpp:172-175), carried by VIRTUAL joints: 1 = joints 0..2 are prismatic
                            along the global x, y, z (Rrel = identity; trel of joint 0 = the base link's COM at q = 0) and joints 3..5 revolute about the base
                            link's own x, y, z through that COM, links 0..4 massless, link 5 the base link; the body's own joints follow (what
                            mh_io_load_xml_artic builds).  The dynamics, calc_jacobian and the contact rows need no special case -- six more 1-DOF columns;
                            the one place that reads the flag is conservative advancement, which adds the base's linear velocity along the direction of
                            approach as CCD::calc_max_dist does for a moving base (CCD.cpp:547-555).  NOT Ravelin's base coordinates (spatial velocity +
                            unit quaternion): the orientation is integrated in three angles, singular when joint 4 reaches +-pi/2.  0 = fixed base */
  /* Collision geometry (optional; nspheres = 0 is the robot alone: no pairs, one mini-step per step).  Sphere primitives fixed to
   * links against ONE static plane -- the closed-form pair of CCD.inl:804-847; the body's own pairs are disabled as ur10.xml:12
   * does.  With spheres the step is TimeSteppingSimulator::step in full: conservative advancement over the pairs
   * (CCD::calc_CA_Euler_step_sphere, the ARTICULATED CCD::calc_max_dist, CCD.cpp:545-583), mini-steps, contact + limit rows in one
   * island.  Contact rows are [d, r x d] . calc_jacobian(link) (ICH:1817-1895).  Impact models as the reference picks them
   * (ICH:123-146): every contact mu-coulomb >= 100 (what ur10.xml:19 gives the robot's contacts) -> the no-slip model
   * (ICH:1009-1417) over NC + NL <= MH_NOSLIP_MAX rows; otherwise the Drumwright-Shell QP -> LCP with contact AND limit variables
   * (ICH-QP:94-497: n = 6 NC + NC nk/2 + 2 NL <= MH_LCP_MAX_N_WAVE rows, lcp_fast_regularized(-20, 4, -8) on the persistent _z /
   * _zlast, then the Lemke ladder). */
  int    nspheres;
  int    sphere_link[MH_ARTIC_MAX_SPHERES];
  double sphere_center[MH_ARTIC_MAX_SPHERES][3];   /* link frame */
  double sphere_radius[MH_ARTIC_MAX_SPHERES];
  double plane_R[9];                               /* as mh_scene: row-major rotation of the plane frame, its +Y is the normal */
  double plane_o[3];
  double cp_epsilon, cp_mu_coulomb;                /* ContactParameters of the (robot, plane) pair */
  double min_step_size;                            /* TimeSteppingSimulator.cpp:48 (sqrt eps) */
  double contact_dist_thresh;                      /* ConstraintSimulator.cpp:56 (1e-6) */
  double cp_mu_viscous, cp_compliance;             /* used by the Drumwright-Shell model only (mu_coulomb < 100) */
  int    cp_nk;                                    /* friction-cone-edges (>= 4, even; ur10.xml:19 has 4); 0 is read as 4 */
  int    cstab_max_iterations;                     /* ConstraintStabilization::max_iterations ("constraint-stabilization-max-iterations"; ur10.xml:11
                                                      sets 0 = off; the reference's own default is UINT_MAX, see MH_CSTAB_DEFAULT_MAX_ITERATIONS) */
  /* ConstraintStabilization::stabilize after every step (TSS:97) with the body's JOINT LIMITS as rows (CStab:257-304 add_limit_constraints:
   * one row per finite limit, signed_violation = its distance; L_v = violation - |eps| - NEAR_ZERO, CStab:434-441; MM = L X L',
   * CStab:932-970; the line search of update_q over the limit slacks, CStab:1056-1216, 1322-1379).  Bodies WITH sphere primitives add a
   * contact row per (sphere, plane) pair (CStab:306-345: the synthetic contact "between separated bodies" when the signed distance is
   * at least NEAR_ZERO, find_contacts' otherwise; Cn_v = distance - |eps| - NEAR_ZERO, CStab:431) and the mixed LCP
   * MM = [Cn X Cn'  Cn X L'; .  L X L'] (CStab:705-904, 932-970); update_q's line search then evaluates the sphere distances too. */
  double cstab_eps;                                /* ConstraintStabilization::eps ("unilateral-stabilization-tol"), default NEAR_ZERO (CStab:59) */
} mh_artic_model;

/* B worlds resident on the GPU: joint positions q and velocities qd (B x nj each) + mh_world_aux (rand() stream, time,
 * status, counters; vns / vns_size hold ImpactConstraintHandler::_v, the warm start of the limit LCP).
 *   step      nsteps x TimeSteppingSimulator::step(dt) in ONE launch
 *   fwd_dyn   seam B4: qdd = H(q)^-1 (tau - C(q, qd)) for the resident states (tau: B x nj device-side copy of host
 *             values, or NULL = 0); H_out (B x nj x nj, row-major) optional -- the generalized inertia
 *             (get_generalized_inertia, used by compute_X)
 */
typedef struct mh_artic_batch mh_artic_batch;
int mh_artic_batch_create(const mh_artic_model* model, int B, mh_artic_batch** out);
int mh_artic_batch_destroy(mh_artic_batch* ab);
int mh_artic_batch_device(const mh_artic_batch* ab);   /* the device the batch lives on (moby_hip.h, Devices) */
int mh_artic_batch_upload(mh_artic_batch* ab, const double* q, const double* qd, const mh_world_aux* aux);
int mh_artic_batch_step(mh_artic_batch* ab, void* stream, double dt, int nsteps);
int mh_artic_batch_fwd_dyn(mh_artic_batch* ab, const double* tau, double* qdd_out, double* H_out);
int mh_artic_batch_download(mh_artic_batch* ab, double* q, double* qd, mh_world_aux* aux);
/* link poses of the resident states (B x nj x 12: row-major R (9), origin (3), model frame): what a viewer or a
 * collision front end on the host needs */
int mh_artic_batch_link_poses(mh_artic_batch* ab, double* poses);
/* RCArticulatedBodyd::calc_jacobian(frame at a point, link, J) of the resident states: the 6 x nj map from joint velocities to
 * the velocity of link `link` at the given points (B x 3, model frame) -- rows 0..2 the linear velocity of the point, rows 3..5
 * the angular velocity, global axes; column j is zero unless joint j lies between the link and the base.  What a contact on a
 * link multiplies its direction row [d, r x d] with (ImpactConstraintHandler::add_contact_dir_to_Jacobian, ICH:1817-1845).
 * J_out: B x 6 x nj, row-major. */
int mh_artic_batch_jacobian(mh_artic_batch* ab, int link, const double* points, double* J_out);

#ifdef __cplusplus
}
#endif
#endif
