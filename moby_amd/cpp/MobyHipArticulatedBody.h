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
// Generalized coordinates / velocities are the joint positions / velocities in the model's joint order (parents first),
// as get_generalized_coordinates_euler / get_generalized_velocity return them for a fixed-base RCArticulatedBody.
#ifndef MOBY_HIP_ARTICULATED_ADAPTER_H
#define MOBY_HIP_ARTICULATED_ADAPTER_H
#include <stdexcept>
#include <vector>
#include "../../include/moby_hip_artic.h"

namespace MobyHip {

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
