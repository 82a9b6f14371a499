// MobyHipStackSimulator.h -- reference-side adapter for LARGE worlds (include/moby_hip_stack.h): the whole-step seam B5
// (Moby::TimeSteppingSimulator::step, include/Moby/TimeSteppingSimulator.h:36) and the stabilisation seam B3
// (Moby::ConstraintStabilization::stabilize, include/Moby/ConstraintStabilization.h:24; call site
// src/TimeSteppingSimulator.cpp:97) for B copies of one scene of up to 128 bodies.
//
//   MobyHip::BatchedStackSimulator sim(scene, B, state);
//   sim.step(1e-3);               // Simulator::step(dt) on every world; returns dt
//   sim.stabilize();              // cstab.stabilize(sim) alone: configurations change, velocities are restored
//
// Conventions as in MobyHipSimulator.h: bodies in id order, x y z qx qy qz qw + eSpatial velocity per body, exceptions
// of the reference as sticky status bits, rand() of every world at srand(1).
#ifndef MOBY_HIP_STACK_SIMULATOR_ADAPTER_H
#define MOBY_HIP_STACK_SIMULATOR_ADAPTER_H
#include <cmath>
#include <stdexcept>
#include <vector>
#include "../../include/moby_hip_stack.h"

namespace MobyHip {

// The simulator's <ImplicitConstraint> list as mh_big_scene wants it.  add() takes what a Moby::Joint holds at the links'
// reference poses -- the joint location and axis in the GLOBAL frame (Joint::get_location(GLOBAL), RevoluteJoint::get_axis /
// PrismaticJoint::get_axis / PlanarJoint::get_normal / UniversalJoint::get_axis(eAxis1, eAxis2)) and the two links' poses
// (x y z qx qy qz qw; NULL = the static world) -- and stores it in the links' frames.  attach() points a scene at the tables.
class ImplicitJoints {
 public:
  void add(int type, int inboard, const double* inboard_pose, int outboard, const double* outboard_pose, const double location[3],
           const double axis[3] = NULL, const double axis2[3] = NULL) {
    double Ri[9], Ro[9], xi[3] = {0, 0, 0}, xo[3] = {0, 0, 0};
    rot(inboard_pose, Ri, xi); rot(outboard_pose, Ro, xo);
    double a[3] = {0, 0, 1};
    if (axis) { const double n = std::sqrt(axis[0]*axis[0] + axis[1]*axis[1] + axis[2]*axis[2]); for (int k = 0; k < 3; k++) a[k] = axis[k] / n; }
    double t1[3], t2[3]; basis(a, t1, t2);
    double vin[9] = {0}, vout[9] = {0};
    if (type == MH_IJOINT_REVOLUTE) { tmul(Ri, a, vin); tmul(Ri, a, vin + 3); tmul(Ro, t1, vout); tmul(Ro, t2, vout + 3); }
    else if (type == MH_IJOINT_UNIVERSAL) {
      double b[3] = { t1[0], t1[1], t1[2] };
      if (axis2) { const double d = a[0]*axis2[0] + a[1]*axis2[1] + a[2]*axis2[2]; double n = 0; for (int k = 0; k < 3; k++) { b[k] = axis2[k] - a[k] * d; n += b[k] * b[k]; } n = std::sqrt(n); for (int k = 0; k < 3; k++) b[k] /= n; }
      tmul(Ri, a, vin); tmul(Ro, b, vout);
    } else if (type == MH_IJOINT_PLANAR) { tmul(Ri, t1, vin); tmul(Ri, t2, vin + 3); tmul(Ri, a, vin + 6); tmul(Ro, a, vout); tmul(Ro, a, vout + 3); }
    else if (type == MH_IJOINT_PRISMATIC) { const double* tri[3] = { t1, t2, a }; for (int k = 0; k < 3; k++) { tmul(Ri, tri[k], vin + 3 * k); tmul(Ro, tri[(k + 1) % 3], vout + 3 * k); } }
    else if (type == MH_IJOINT_FIXED) { for (int k = 0; k < 3; k++) { double e[3] = {0, 0, 0}, f[3] = {0, 0, 0}; e[k] = 1.0; f[(k + 1) % 3] = 1.0; tmul(Ri, e, vin + 3 * k); tmul(Ro, f, vout + 3 * k); } }
    const double di[3] = { location[0] - xi[0], location[1] - xi[1], location[2] - xi[2] }, d_o[3] = { location[0] - xo[0], location[1] - xo[1], location[2] - xo[2] };
    double ai[3], ao[3]; tmul(Ri, di, ai); tmul(Ro, d_o, ao);
    _type.push_back(type); _in.push_back(inboard); _out.push_back(outboard);
    _ain.insert(_ain.end(), ai, ai + 3); _aout.insert(_aout.end(), ao, ao + 3);
    _vin.insert(_vin.end(), vin, vin + 9); _vout.insert(_vout.end(), vout, vout + 9);
  }
  void attach(mh_big_scene& sc) const {
    sc.njoints = (int)_type.size();
    sc.joint_type = _type.data(); sc.joint_inboard = _in.data(); sc.joint_outboard = _out.data();
    sc.joint_anchor_in = _ain.data(); sc.joint_anchor_out = _aout.data(); sc.joint_vec_in = _vin.data(); sc.joint_vec_out = _vout.data();
  }
 private:
  static void rot(const double* pose, double R[9], double x[3]) {      // Ravelin's quaternion -> matrix form (diagonal 2 (w^2 + q_i^2) - 1)
    if (!pose) { for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0; return; }
    for (int k = 0; k < 3; k++) x[k] = pose[k];
    const double qx = pose[3], qy = pose[4], qz = pose[5], w = pose[6];
    R[0] = 2 * (w * w + qx * qx) - 1; R[1] = 2 * (qx * qy - qz * w); R[2] = 2 * (qx * qz + qy * w);
    R[3] = 2 * (qx * qy + qz * w); R[4] = 2 * (w * w + qy * qy) - 1; R[5] = 2 * (qy * qz - qx * w);
    R[6] = 2 * (qx * qz - qy * w); R[7] = 2 * (qy * qz + qx * w); R[8] = 2 * (w * w + qz * qz) - 1;
  }
  static void tmul(const double R[9], const double v[3], double out[3]) { for (int c = 0; c < 3; c++) out[c] = R[c] * v[0] + R[3 + c] * v[1] + R[6 + c] * v[2]; }   // R' v
  static void basis(const double n[3], double s[3], double t[3]) {     // Vector3d::determine_orthonormal_basis, as the oracle pins it
    const double ax = std::fabs(n[0]), ay = std::fabs(n[1]), az = std::fabs(n[2]);
    double e[3] = {0, 0, 0};
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    s[0] = n[1] * e[2] - n[2] * e[1]; s[1] = n[2] * e[0] - n[0] * e[2]; s[2] = n[0] * e[1] - n[1] * e[0];
    const double l = std::sqrt(s[0]*s[0] + s[1]*s[1] + s[2]*s[2]); for (int k = 0; k < 3; k++) s[k] /= l;
    t[0] = n[1] * s[2] - n[2] * s[1]; t[1] = n[2] * s[0] - n[0] * s[2]; t[2] = n[0] * s[1] - n[1] * s[0];
  }
  std::vector<int> _type, _in, _out; std::vector<double> _ain, _aout, _vin, _vout;
};

class BatchedStackSimulator {
 public:
  double current_time;
  BatchedStackSimulator(const mh_big_scene& scene, int B, const double* state /* B x nb x 13 */)
      : current_time(0.0), _nb(scene.nb), _B(B), _bb(NULL), _dirty(false)
  {
    if (mh_big_batch_create(&scene, B, &_bb) != MH_OK) throw std::runtime_error(mh_last_error());
    _state.assign(state, state + (size_t)B * _nb * MH_BODY_STATE);
    _aux.resize((size_t)B);
    if (mh_big_batch_upload(_bb, _state.data(), NULL) != MH_OK) throw std::runtime_error(mh_last_error());
  }
  ~BatchedStackSimulator() { if (_bb) mh_big_batch_destroy(_bb); }
  double step(double dt, int nsteps = 1) {
    if (mh_big_batch_step(_bb, NULL, dt, nsteps) != MH_OK) throw std::runtime_error(mh_last_error());
    _dirty = true; current_time += dt * nsteps;
    return dt;
  }
  void stabilize() { if (mh_big_batch_stabilize(_bb, NULL) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = true; }
  void get_generalized_coordinates_euler(int w, int b, double q[7]) { sync(); for (int k = 0; k < 7; k++) q[k] = _state[((size_t)w * _nb + b) * MH_BODY_STATE + k]; }
  void get_generalized_velocity(int w, int b, double v[6]) { sync(); for (int k = 0; k < 6; k++) v[k] = _state[((size_t)w * _nb + b) * MH_BODY_STATE + 7 + k]; }
  int status(int w) { sync(); return _aux[(size_t)w].status; }
  const mh_world_aux& counters(int w) { sync(); return _aux[(size_t)w]; }
 private:
  BatchedStackSimulator(const BatchedStackSimulator&);
  BatchedStackSimulator& operator=(const BatchedStackSimulator&);
  void sync() { if (!_dirty) return; if (mh_big_batch_download(_bb, _state.data(), _aux.data()) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = false; }
  int _nb, _B; mh_big_batch* _bb; bool _dirty;
  std::vector<double> _state; std::vector<mh_world_aux> _aux;
};

} // namespace MobyHip
#endif
