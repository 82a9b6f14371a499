/* moby_hip_io.h -- host-side scene and trajectory I/O around libmoby_hip (C ABI).
 *
 * The compatibility layer SURVEY 8(f)2 asks for, so that Moby's own scene files
 * and regression harness drive the GPU stepper unchanged:
 *
 *   mh_io_load_xml        the subset of XMLReader::read (src/XMLReader.cpp:151-204) the many-
 *                         worlds stepper covers: <Sphere> <Box> <Plane> <GravityForce> <RigidBody>
 *                         (<CollisionGeometry>, <InertiaFromPrimitive>) <TimeSteppingSimulator>
 *                         (<DynamicBody> <RecurrentForce> <DisabledPair> <ContactParameters>)
 *                         <CollisionDetectionPlugin> (the rimless-wheel plugin only) and
 *                         <DRIVER step-size>.  Attribute meaning follows the reference loaders:
 *                         src/RigidBody.cpp:132-369, src/Primitive.cpp:244-300, src/SpherePrimitive.cpp:
 *                         138-155,360-380, src/BoxPrimitive.cpp:640-712, src/GravityForce.cpp:74-90,
 *                         src/ContactParameters.cpp:46-135, src/ConstraintSimulator.cpp:540-708,
 *                         src/TimeSteppingSimulator.cpp:463-476.
 *   mh_io_format_row      one row of programs/regress.cpp:82-93: current_time, then the 7 Euler
 *                         coordinates of every enabled body in id order (ostream default format)
 *   mh_io_compare_trajs   programs/compare-trajs.cpp: max over rows of the L-inf difference,
 *                         last line = timing; returns 0 when max_diff <= tol
 *
 * Pure host code (libxml2); no GPU needed.  Library: moby_amd/libmoby_hip_io.so.
 */
#ifndef MOBY_HIP_IO_H
#define MOBY_HIP_IO_H
#include "moby_hip.h"
#include "moby_hip_artic.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MH_IO_ID_LEN 64

typedef struct mh_io_scene {
  mh_scene scene;
  double   state[MH_MAX_BODIES * MH_BODY_STATE];   /* initial state of the enabled bodies, id order */
  char     body_id[MH_MAX_BODIES + 1][MH_IO_ID_LEN]; /* ids of the enabled bodies, then the ground's */
  double   step_size;                              /* <DRIVER step-size>, 0 if absent */
} mh_io_scene;

/* 0 on success; on failure returns nonzero and mh_io_last_error() says what was not understood
 * (unsupported elements are errors, never silently dropped) */
int mh_io_load_xml(const char* path, mh_io_scene* out);
const char* mh_io_last_error(void);

/* SDF models (example/ur10/model.sdf): the subset of SDFReader::read_model / read_link / read_joint / read_inertial
 * (src/SDFReader.cpp:880-1000, 432-610, 1288-1330) an articulated-body batch needs -- <link> (pose, inertial: pose, mass,
 * inertia), <joint type="revolute|prismatic"> (parent, child, axis: xyz, use_parent_model_frame, limit lower / upper).
 * As in the reference a joint whose parent is "world" makes the body fixed-base, a joint sits at its child link's origin
 * (no <pose> under <joint> in the scope here), and the axis is given in the parent LINK's frame when
 * use_parent_model_frame is 1 (SDFReader.cpp:559-562), else in the child link's.  Joints are ordered parents first
 * (file order within a level); names of the links the joints carry come back in link_id.  Collision geometry, visuals,
 * surfaces and plugins are skipped (out of scope: SURVEY 2).  0 on success. */
typedef struct mh_io_artic {
  mh_artic_model model;
  char link_id[MH_ARTIC_MAX_JOINTS][MH_IO_ID_LEN];
  char joint_id[MH_ARTIC_MAX_JOINTS][MH_IO_ID_LEN];
} mh_io_artic;
int mh_io_load_sdf(const char* path, const double gravity[3], mh_io_artic* out);

/* URDF robots (example/urdf/pendulum.urdf): the subset of URDFReader::read (src/URDFReader.cpp:65-141, 144-203, 299-403, 456-566, 599-636,
 * 943-1043) an articulated-body batch needs -- <robot name>, <link name> (<inertial>: origin xyz / rpy, mass value, inertia ixx..izz given in the
 * inertial frame), <joint name type="revolute|continuous|prismatic|fixed"> (parent link, child link, origin xyz / rpy RELATIVE TO THE PARENT LINK's
 * frame, axis xyz in the joint's frame, default (1, 0, 0); limit lower / upper).  As in the reference the child link's frame IS the joint's frame, a
 * revolute joint without limits gets -pi/2 .. pi/2, continuous and prismatic ones -10000 .. 10000 (URDFReader.cpp:326-345), a <limit> element counts
 * only if it carries effort, lower or upper (:549), <dynamics damping friction> land in Joint::mu_fv / mu_fc, which a reduced-coordinate body never
 * reads (only src/MCArticulatedBody.cpp:419-420 does) and are therefore accepted and without effect, and floating / planar joints are not read.
 * The base is the one link that is no joint's child; it is FIXED (RCArticulatedBody floating-base="false", example/urdf/pendulum-urdf.xml:23), its
 * frame is the model frame.  The batch knows 1-DOF joints only, so a FIXED joint welds its child to the link that carries it: masses add up, the
 * COM and the tensor about it follow the parallel-axis theorem, joints hanging from the welded link hang from the carrier -- one rigid body, which is
 * what a zero-DOF joint is in reduced coordinates.  A moving link must end up with mass > 0 (the reference disables a link without: :613-616).
 * Collision geometry is skipped here as in mh_io_load_sdf; mh_io_load_xml_artic (urdf-filename) keeps the <collision> spheres.  0 on success. */
int mh_io_load_urdf(const char* path, const double gravity[3], mh_io_artic* out);

/* A Moby XML file with ONE <RCArticulatedBody> -- fixed base, or floating-base="true" (RCArticulatedBody.cpp:172-175): then six VIRTUAL joints come first in the model
 * (mh_artic_model.floating_base, include/moby_hip_artic.h: sliders along the global x, y, z and hinges about the base link's own x, y, z through its COM; link 5 is the base
 * link, "<body id>.base-tx" ... "-rz" the joint ids), q0[0..5] = 0 is the pose the file states, qd0[0..2] the base link's linear-velocity and qd0[3..5] its angular-velocity in
 * its own axes, translate="x y z" moves the whole body (:176-199; taken with a floating base only), the model frame is the global frame and the base link's collision geometry
 * moves like any link's.  <FixedJoint> (either kind of base): the batch knows 1-DOF joints only, so the outboard link is WELDED onto the link that carries it, as
 * mh_io_load_urdf does for a URDF "fixed" joint -- masses add up, the COM and the (then full) tensor about it follow the parallel-axis theorem, the welded link's
 * collision geometry and the joints hanging from it ride on the carrier; its id stays known to <DisabledPair> / <ContactParameters> --
 * (the files of example/joint-limits, example/reduced-coords: RigidBody links
 * with InertiaFromPrimitive | mass / inertia, <RevoluteJoint> / <PrismaticJoint> with location / axis in the global frame, lower-limits,
 * upper-limits, restitution-coeff, q, qd -- RCArticulatedBody.cpp:162-260, Joint.cpp:184-345, RevoluteJoint.cpp:39-55,
 * PrismaticJoint.cpp; or urdf-filename="..." (ArticulatedBody.cpp:250-273: links and joints from the URDF file next to the XML file, as
 * mh_io_load_urdf reads them, q = qd = 0, the first <collision> sphere of a link as its link sphere) -> mh_artic_model at q = 0 (the link poses the file states; the base link is the link no joint carries), the
 * joints' q / qd as the initial state (q0, qd0: MH_ARTIC_MAX_JOINTS doubles each, joint order = out->joint_id), <DRIVER step-size>,
 * the simulator's GravityForce, fdyn-algorithm crb / fsab.
 * Collision geometry: a link's <CollisionGeometry> with a <Sphere> primitive becomes a link sphere; ONE disabled <RigidBody> with a
 * <Plane> primitive becomes the plane, with the <ContactParameters> that name it (object ids: the plane body and the articulated body
 * or one of its links; one set for every link, the first found).  Any other collision geometry must be unable to collide (no other
 * body in the simulator and every link-link pair under <DisabledPair>, as those example files do) -- otherwise the file is rejected.
 * 0 on success. */
int mh_io_load_xml_artic(const char* path, mh_io_artic* out, double* q0, double* qd0, double* step_size);

/* writes at most cap bytes (NUL-terminated) and returns the length the full row needs */
int mh_io_format_row(double t, const double* state, int nb, char* buf, int cap);

/* *max_diff = max over all rows but the last of the L-inf difference; the last line of each file is
 * its timing (returned in timing[0], timing[1] if timing != NULL).  Returns 0 if max_diff <= tol,
 * 1 if larger, -1 on I/O / shape errors (as the reference tool's exit code) */
int mh_io_compare_trajs(const char* file1, const char* file2, double tol, double* max_diff, double* timing);

#ifdef __cplusplus
}
#endif
#endif
