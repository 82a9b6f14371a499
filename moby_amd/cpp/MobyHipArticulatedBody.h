// MobyHipArticulatedBody.h -- reference-side adapter for fixed-base articulated bodies (include/moby_hip_artic.h): the
// forward-dynamics seam B4 (Ravelin::RCArticulatedBodyd::calc_fwd_dyn / get_generalized_inertia as Moby calls them:
// src/Simulator.cpp:552, src/ImpactConstraintHandler.cpp:1600-1607) and the whole-step seam for B copies of one model.
//
//   mh_io_artic io; mh_io_load_sdf("model.sdf", gravity, &io);        // or fill mh_artic_model from Moby's Joint / RigidBody objects
//   MobyHip::BatchedArticulatedBody robots(io.model, B, q, qd);
//   robots.calc_fwd_dyn(tau, qdd);                 // qdd = H^-1 (tau - C) for every copy (tau may be NULL)
//   robots.get_generalized_inertia(H);             // B x nj x nj, row-major
//   robots.step(5e-4, 200);                        // TimeSteppingSimulator::step x 200 in one launch (joint limits included)
//
// Collision geometry (sphere primitives on links against one static plane, contacts with mu-coulomb >= 100 as ur10.xml:19 has
// them) is part of the model: fill it before constructing the batch, e.g.
//   MobyHip::add_link_sphere(io.model, finger_link, centre_in_link_frame, 0.03);     // <Sphere> CollisionGeometry on a link
//   MobyHip::set_ground_plane(io.model, normal, point_on_plane, /*epsilon*/ 0.0, /*mu_coulomb*/ 100.0);
// step() then runs TimeSteppingSimulator::step in full: conservative advancement, mini-steps, contact + limit rows in one LCP.
//
// Generalized coordinates / velocities are the joint positions / velocities in the model's joint order (parents first),
// as get_generalized_coordinates_euler / get_generalized_velocity return them for a fixed-base RCArticulatedBody.
#ifndef MOBY_HIP_ARTICULATED_ADAPTER_H
#define MOBY_HIP_ARTICULATED_ADAPTER_H
#include <cmath>
#include <stdexcept>
#include <vector>
#include "../../include/moby_hip_artic.h"

namespace MobyHip {

// CollisionGeometry of a SpherePrimitive on link `link` (centre in the link frame)
inline void add_link_sphere(mh_artic_model& m, int link, const double centre[3], double radius)
{
  if (m.nspheres >= MH_ARTIC_MAX_SPHERES) throw std::runtime_error("MobyHip::add_link_sphere: more than MH_ARTIC_MAX_SPHERES spheres");
  const int s = m.nspheres++;
  m.sphere_link[s] = link; m.sphere_radius[s] = radius;
  for (int k = 0; k < 3; k++) m.sphere_center[s][k] = centre[k];
}
// The static PlanePrimitive (its +Y axis is the normal, PlanePrimitive.cpp) through `point`, and the ContactParameters of the
// (robot, plane) pair; simulator constants at the reference's defaults (TimeSteppingSimulator.cpp:48, ConstraintSimulator.cpp:56)
inline void set_ground_plane(mh_artic_model& m, const double normal[3], const double point[3], double epsilon, double mu_coulomb)
{
  double n[3] = { normal[0], normal[1], normal[2] };
  const double len = std::sqrt(n[0]*n[0] + n[1]*n[1] + n[2]*n[2]);
  for (int k = 0; k < 3; k++) n[k] /= len;
  int a = 0; for (int k = 1; k < 3; k++) if (std::fabs(n[k]) < std::fabs(n[a])) a = k;
  double e[3] = { 0.0, 0.0, 0.0 }; e[a] = 1.0;
  double x[3] = { n[1]*e[2] - n[2]*e[1], n[2]*e[0] - n[0]*e[2], n[0]*e[1] - n[1]*e[0] };
  const double xl = std::sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]);
  for (int k = 0; k < 3; k++) x[k] /= xl;
  const double z[3] = { x[1]*n[2] - x[2]*n[1], x[2]*n[0] - x[0]*n[2], x[0]*n[1] - x[1]*n[0] };
  for (int k = 0; k < 3; k++) { m.plane_R[3*k] = x[k]; m.plane_R[3*k+1] = n[k]; m.plane_R[3*k+2] = z[k]; m.plane_o[k] = point[k]; }
  m.cp_epsilon = epsilon; m.cp_mu_coulomb = mu_coulomb; m.cp_mu_viscous = 0.0; m.cp_compliance = 0.0; m.cp_nk = 4;   // set the last three directly for the D-S model
  m.min_step_size = 1.4901161193847656e-08; m.contact_dist_thresh = 1e-6;
}

class BatchedArticulatedBody {
 public:
  BatchedArticulatedBody(const mh_artic_model& model, int B, const double* q, const double* qd) : _nj(model.nj), _B(B), _ab(NULL), _dirty(false)
  {
    if (mh_artic_batch_create(&model, B, &_ab) != MH_OK) throw std::runtime_error(mh_last_error());
    _q.assign(q, q + (size_t)B * _nj); _qd.assign(qd, qd + (size_t)B * _nj); _aux.resize((size_t)B);
    if (mh_artic_batch_upload(_ab, _q.data(), _qd.data(), NULL) != MH_OK) throw std::runtime_error(mh_last_error());
  }
  ~BatchedArticulatedBody() { if (_ab) mh_artic_batch_destroy(_ab); }
  void set_generalized_coordinates_euler(const double* q) { _q.assign(q, q + _q.size()); if (mh_artic_batch_upload(_ab, _q.data(), NULL, NULL) != MH_OK) throw std::runtime_error(mh_last_error()); }
  void set_generalized_velocity(const double* qd) { _qd.assign(qd, qd + _qd.size()); if (mh_artic_batch_upload(_ab, NULL, _qd.data(), NULL) != MH_OK) throw std::runtime_error(mh_last_error()); }
  void calc_fwd_dyn(const double* tau /* B x nj or NULL */, double* qdd /* B x nj */) { if (mh_artic_batch_fwd_dyn(_ab, tau, qdd, NULL) != MH_OK) throw std::runtime_error(mh_last_error()); }
  void get_generalized_inertia(double* H /* B x nj x nj */) { std::vector<double> qdd((size_t)_B * _nj); if (mh_artic_batch_fwd_dyn(_ab, NULL, qdd.data(), H) != MH_OK) throw std::runtime_error(mh_last_error()); }
  double step(double dt, int nsteps = 1) { if (mh_artic_batch_step(_ab, NULL, dt, nsteps) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = true; return dt; }
  const std::vector<double>& q() { sync(); return _q; }
  const std::vector<double>& qd() { sync(); return _qd; }
  int status(int w) { sync(); return _aux[(size_t)w].status; }
  const mh_world_aux& aux(int w) { sync(); return _aux[(size_t)w]; }   // counters: mini_steps, lcp_solves, lcp_rows ...
  int num_joints() const { return _nj; }
 private:
  BatchedArticulatedBody(const BatchedArticulatedBody&);
  BatchedArticulatedBody& operator=(const BatchedArticulatedBody&);
  void sync() { if (!_dirty) return; if (mh_artic_batch_download(_ab, _q.data(), _qd.data(), _aux.data()) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = false; }
  int _nj, _B; mh_artic_batch* _ab; bool _dirty;
  std::vector<double> _q, _qd; std::vector<mh_world_aux> _aux;
};

} // namespace MobyHip
#endif
