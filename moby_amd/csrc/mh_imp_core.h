// The island pipeline shared by the impact-handler entry (mh_impact.hip, seam B2) and the large-world stepper
// (mh_big.hip, seams B5 / B3): device buffers of B worlds x up to ncmax contacts x up to nb bodies, and the host
// functions that run ImpactConstraintHandler::process_constraints (mode IMPACT) or ConstraintStabilization's
// compute_problem_data + determine_dq (mode STAB) over every island of every world.  Host-only header: the kernels
// live in mh_impact.hip.
#pragma once
#include "mh_host.h"
#include "../../include/moby_hip_impact.h"

enum { MH_CORE_IMPACT = 0, MH_CORE_STAB = 1 };

struct mh_imp_core {
  int B, nb, ncmax, nk, kh, nmax, islmax;
  // inputs owned by the caller of the pipeline
  const double* mass; const double* inertia;        // nb, nb x 3 (shared by all worlds)
  double* state;                                    // B x nb x 13
  const mh_contact* contacts;                       // B x ncmax
  const int* ncount;                                // B contacts in use, or NULL (= ncmax)
  const double* cdist;                              // MH_CORE_STAB: signed_violation of each contact (B x ncmax)
  double stab_eps;                                  // ConstraintStabilization::eps
  uint32_t* rng;                                    // B x MH_RAND_WORDS
  int* status;                                      // B, MH_WORLD_* bits (sticky)
  // An exception of the reference's handler (LCPSolverException ICH-QP:225, runtime_error ICH:1282, std::exception ICH-AP:334, the assert of ICH:1184-1186) is
  // caught nowhere up to main(): it unwinds process_constraints.  thrown[b] = 1 from the kernel that sets MH_WORLD_LCP_FAILED in mode IMPACT; k_unwind, after
  // the round, then takes the world's remaining islands away (nisl = 0: no later round, no tolerance check).  Zeroed by k_prep: one call = one process_constraints.
  int* thrown;
  // island tables and per-contact problem data, in island order
  int* order; int* cbody; double* cpar; double* W; double* XJ; double* Cv; double* xinv;
  int* nisl; int* isl_start; int* isl_len; int* isl_model; int* maxisl;   // isl_model: 0 Drumwright-Shell, 1 no-slip (ICH:123-135); maxisl: max over worlds of nisl
  // the LCP of the current round
  double* G; double* MM; double* qq; double* z; int* zsz; int* ncur;
  // what the handler object keeps between solves: _zlast and the storage of _z (ICH-QP:158-162, 233)
  double* zlast; double* zbuf; int* zlast_size; int* zbuf_size; int* zbuf_cap;
  double* vns; int* vns_size;                       // ImpactConstraintHandler::_v, the no-slip LCP's z (ICH:1239): B x MH_NOSLIP_MAX, B
  int* run; int* need2; int* again; int* lst1; int* lst2; unsigned* piv1; unsigned* piv2;
  double* imp;                                      // B x ncmax x 3 accumulated (cn, cs, ct), caller order
  unsigned long long* cnt;                          // B x 5: LCPs solved, rows, pivots, LCP-entry bytes 8 (n^2 + 2n), stabilisation rows
  const double* fcos; const double* fsin;           // kh each (host libm)
  // Anitescu-Potra model (ImpactConstraintHandlerLCP.cpp; the reference's -DUSE_AP build): ap != 0 sends every island with
  // finite friction through it.  nk4 = NK_DIRS rows per contact, apcos / apsin their polygon directions (host libm),
  // apw = the contacts' accumulated impulse wrenches in the global frame (B x ncmax x 6, island order)
  int ap, nk4; const double* apcos; const double* apsin; double* apw;
  // implicit joints of the scene (0 / NULL without): their dynamic links are nodes of the island search and an edge
  // joins the two (UC:993-1008); the handler itself never sees their rows (island_ijoints stays empty in ICH)
  int nj; const int* jin; const int* jout;
  const unsigned char* jointed;                     // nb flags: body belongs to an island with a joint
  // MH_CORE_STAB, contact islands that hold jointed bodies (isl_model 3): set_unilateral_constraint_data with implicit joints
  // (CStab:705-904) and compute_X's general case (ICH:1590-1695).  Joint tables (scene), the island's sorted bodies, and per
  // world (one such island per round): dense X and its factors' scratch (ngc <= 96), X Cn' rows, the bilateral step's data
  const int* jtype; const double* janchor_in; const double* janchor_out; const double* jvec_in; const double* jvec_out;
  int* isl_nbod; int* isl_bod;                      // B x islmax, B x islmax x MH_IJOINT_MAX_BODIES
  double* bT1; double* bT2; double* bT3; double* bT4;   // B x 96 x 96 each: H'J iM then X; G; M G; H'
  double* bXCn;                                     // B x ncmax x 96
  double* bJiM; double* blam; int* bact; int* bk;   // B x 48 x 96, B x 48, B x 48, B
  double* ws_d; int* ws_i;                          // block-solver workspace (nmax > 64)
  // task mode of the Lemke ladder (one workgroup per (world, attempt), mh_lcp_block.h): per-task workspace, z, status, pivots, sizes,
  // rand() scratch, model work; solved_at per world.  Allocated on first use for t_cap tasks.
  void* s2; void* ev0; void* ev1;                   // second stream + events: the ladder's tasks beside lcp_fast (speculation, core_solve_round)
  double* t_wsd; int* t_wsi; double* t_z; int* t_st; unsigned* t_piv; int* t_zsz; uint32_t* t_rng; double* t_work; int* solved_at; long t_cap; long t_slots;     // t_cap tasks (z, status, pivots, sizes, rand() scratch, work), t_slots LU workspaces (mh_task_slots: the persistent workgroups)
  double* work;                                     // B x MH_WORK (mh_host.h): SURVEY 8(d)'s model work of the block solver's factorisations (2/3 k^3 flops, 8 k^2 bytes), accumulated
  int* hmax;                                        // pinned host copy of maxisl
  void* allocs[64]; int nallocs;
};

extern "C" {
// allocates every buffer the pipeline owns (everything but mass / inertia / state / contacts / ncount / cdist / rng /
// status, which the caller points at its own memory) and uploads the friction-polygon table
MH_HIDDEN int mh_imp_core_create(mh_imp_core* c, int B, int nb, int ncmax, int nk, int nmax);
MH_HIDDEN void mh_imp_core_destroy(mh_imp_core* c);
// after create, for scenes with implicit joints: the joint tables (device pointers, the caller's) and the scratch of the
// stabiliser's general compute_X
MH_HIDDEN int mh_imp_core_enable_joints(mh_imp_core* c, int nj, const int* jtype, const int* jin, const int* jout, const double* janchor_in,
                                        const double* janchor_out, const double* jvec_in, const double* jvec_out, const unsigned char* jointed);
// prep (islands, rows, X C^T, C v) + one round per island (gram, LCP matrix, solver chain, impulse application,
// restitution / second solve) + the impact-tolerance check.  Synchronises `stream` once, after the island search, to
// learn how many rounds the batch needs.
MH_HIDDEN int mh_imp_core_process(mh_imp_core* c, void* stream, int mode);
// host copy of c->work (B x 2 doubles), optionally zeroing the device counters afterwards
MH_HIDDEN int mh_imp_core_lu_work(mh_imp_core* c, double* work, int reset);
}
