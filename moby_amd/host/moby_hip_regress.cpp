// moby-hip-regress: the reference's regression runner (programs/regress.cpp) on top of the
// MI355X stepper.  Same command line, same options file, same output:
//
//   moby-hip-regress <options file> <xml file> <output file>
//
// options (whitespace-separated words in the options file, programs/regress.cpp:205-233):
//   -s=<step size>  -mt=<max time>  -mi=<max iterations>  -oi  -or  -p=<plugin>
// Of the plugins only librimless-wheel-init.so is emulated (example/rimless-wheel/init.cpp:
// 166-191: theta = 0, z = 0.866025403784439, rates from $RIMLESS_WHEEL_THETAD).
// Output: one row per step BEFORE it is taken -- current_time, then the 7 Euler coordinates of
// every enabled body in id order -- and the elapsed CPU seconds as the last line
// (regress.cpp:82-93, 274-277).  Extra: -B=<n> steps n identical worlds and writes world 0
// (for timing the batch path); -chunk=<steps per launch> (default 256).
// A file with one fixed-base <RCArticulatedBody> (example/joint-limits/*.xml; mh_io_load_xml_artic) runs on the articulated stepper:
// the row is current_time and the joint positions (get_generalized_coordinates_euler of a fixed-base reduced-coordinate body).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <sstream>
#include "../../include/moby_hip_io.h"
#include "../../include/moby_hip_artic.h"

int main(int argc, char** argv)
{
  if (argc != 4) { std::cerr << "syntax: moby-hip-regress <options file> <xml file> <output file>" << std::endl; return -1; }
  std::ifstream oin(argv[1]);
  if (oin.fail()) { std::cerr << "regress: error opening options file " << argv[1] << std::endl; return -1; }
  double step_size = 0.001, max_time = 1e300; unsigned long max_iter = ~0ul; int B = 1, chunk = 256;
  bool out_iter = false, out_rate = false, wheel_init = false;
  std::string w;
  while (oin >> w) {
    if (w.find("-oi") != std::string::npos) out_iter = true;
    else if (w.find("-or") != std::string::npos) out_rate = true;
    else if (w.find("-s=") != std::string::npos) step_size = std::atof(w.substr(3).c_str());
    else if (w.find("-mi=") != std::string::npos) max_iter = std::strtoul(w.substr(4).c_str(), nullptr, 10);
    else if (w.find("-mt=") != std::string::npos) max_time = std::atof(w.substr(4).c_str());
    else if (w.find("-B=") != std::string::npos) B = std::atoi(w.substr(3).c_str());
    else if (w.find("-chunk=") != std::string::npos) chunk = std::atoi(w.substr(7).c_str());
    else if (w.find("-p=") != std::string::npos) {
      if (w.find("rimless-wheel-init") != std::string::npos) wheel_init = true;
      else { std::cerr << "regress: plugin " << w.substr(3) << " is not emulated" << std::endl; return -1; }
    }
  }
  if (!(step_size > 0.0 && step_size < 1.0) || B < 1 || chunk < 1) { std::cerr << "regress: bad options" << std::endl; return -1; }
  mh_io_scene io;
  if (mh_io_load_xml(argv[2], &io) != 0) {
    const std::string why = mh_io_last_error();
    mh_io_artic ar; double q0[MH_ARTIC_MAX_JOINTS], qd0[MH_ARTIC_MAX_JOINTS], file_dt = 0.0;
    if (mh_io_load_xml_artic(argv[2], &ar, q0, qd0, &file_dt) != 0) {
      const std::string why2 = mh_io_last_error();                // a file without an RCArticulatedBody: the first reader's complaint is the relevant one
      std::cerr << "regress: " << (why2.find("expected exactly one <RCArticulatedBody>") != std::string::npos ? why : why2) << std::endl; return -1;
    }
    // ---- one fixed-base articulated body ----
    const int nj = ar.model.nj;
    std::vector<double> q((size_t)B * nj), qd((size_t)B * nj);
    for (int b = 0; b < B; b++) for (int i = 0; i < nj; i++) { q[(size_t)b * nj + i] = q0[i]; qd[(size_t)b * nj + i] = qd0[i]; }
    mh_artic_batch* ab = nullptr;
    if (mh_artic_batch_create(&ar.model, B, &ab) != MH_OK || mh_artic_batch_upload(ab, q.data(), qd.data(), nullptr) != MH_OK) { std::cerr << "regress: " << mh_last_error() << std::endl; return -1; }
    std::ofstream out(argv[3]);
    if (out.fail()) { std::cerr << "regress: cannot open " << argv[3] << std::endl; return -1; }
    const clock_t start = clock();
    double t_now = 0.0, total_t = 0.0; unsigned long iter = 0;
    std::vector<mh_world_aux> aux((size_t)B);
    while (true) {
      std::ostringstream o; o << t_now; for (int i = 0; i < nj; i++) o << " " << q[i];          // world 0, before the step
      out << o.str() << std::endl;
      if (out_iter) std::cout << "iteration: " << iter << "  simulation time: " << t_now << std::endl;
      const clock_t pre = clock();
      if (mh_artic_batch_step(ab, nullptr, step_size, 1) != MH_OK || mh_artic_batch_download(ab, q.data(), qd.data(), aux.data()) != MH_OK) { std::cerr << "regress: " << mh_last_error() << std::endl; return -1; }
      total_t += (clock() - pre) / (double)CLOCKS_PER_SEC;
      iter++; t_now += step_size;
      if (out_rate) std::cout << "time to compute last iteration: " << total_t / iter << " (" << total_t / iter << "s/iter, " << total_t / t_now << "s/step)" << std::endl;
      if (iter >= max_iter || t_now > max_time) break;
    }
    if (aux[0].status & ~MH_WORLD_IMPACT_TOL) std::cerr << "regress: world 0 finished with status bits " << aux[0].status << std::endl;
    out << (clock() - start) / (double)CLOCKS_PER_SEC << std::endl;
    mh_artic_batch_destroy(ab);
    return 0;
  }
  const int nb = io.scene.nb, nst = nb * MH_BODY_STATE;
  if (wheel_init) {
    const char* thd = std::getenv("RIMLESS_WHEEL_THETAD");
    if (!thd) { std::cerr << "RIMLESS_WHEEL_THETAD not defined!" << std::endl; return -1; }
    const double theta_dot = std::atof(thd), R = 1.0;
    const double dist_per_sec = (2 * M_PI * R) * (theta_dot / (M_PI * 2.0));
    for (int b = 0; b < nb; b++) if (io.scene.geom_type[b] == MH_GEOM_SPOKES) {
      double* s = io.state + MH_BODY_STATE * b;
      s[0] = 0.0; s[1] = 0.0; s[2] = 0.866025403784439; s[3] = s[4] = s[5] = 0.0; s[6] = 1.0;
      // the plugin's SVelocityd has the GLOBAL pose: its linear part is the velocity of the body point at the origin
      s[7] = dist_per_sec + theta_dot * s[2]; s[8] = 0.0; s[9] = 0.0; s[10] = 0.0; s[11] = theta_dot; s[12] = 0.0;
    }
  }
  std::vector<double> st((size_t)B * nst);
  for (int b = 0; b < B; b++) std::memcpy(&st[(size_t)b * nst], io.state, sizeof(double) * nst);
  std::vector<mh_world_aux> aux((size_t)B);
  for (int b = 0; b < B; b++) mh_world_aux_init(&aux[b], 1);     // rand() as in a fresh process (srand(1))
  std::ofstream out(argv[3]);
  if (out.fail()) { std::cerr << "regress: cannot open " << argv[3] << std::endl; return -1; }
  // the stepper advances `chunk` steps per launch and hands back the coordinates after every step
  // (traj: B x n x nb x 7); the solver state (aux) travels with the body state between launches
  std::vector<double> traj((size_t)B * chunk * nb * 7);
  std::vector<double> row(io.state, io.state + nst);
  const clock_t start = clock();
  double total_t = 0.0, t_now = 0.0;
  unsigned long iter = 0;
  char buf[4096];
  bool go = true;
  while (go) {
    // steps this launch may take: the reference stops after the step that makes ITER >= MAX_ITER or time > MAX_TIME
    int n = 0;
    { double tt = t_now; unsigned long it = iter;
      while (n < chunk) { n++; it++; tt += step_size; if (it >= max_iter || tt > max_time) break; } }
    const clock_t pre = clock();
    if (mh_world_step_batch(&io.scene, B, step_size, n, st.data(), aux.data(), traj.data()) != MH_OK) { std::cerr << "regress: " << mh_last_error() << std::endl; return -1; }
    total_t += (clock() - pre) / (double)CLOCKS_PER_SEC;
    for (int s = 0; s < n; s++) {
      mh_io_format_row(t_now, row.data(), nb, buf, (int)sizeof(buf));
      out << buf << std::endl;
      if (out_iter) std::cout << "iteration: " << iter << "  simulation time: " << t_now << std::endl;
      for (int b = 0; b < nb; b++) for (int k = 0; k < 7; k++) row[MH_BODY_STATE * b + k] = traj[((size_t)s * nb + b) * 7 + k];   // world 0
      iter++; t_now += step_size;
      if (out_rate) std::cout << "time to compute last iteration: " << total_t / iter << " (" << total_t / iter << "s/iter, " << total_t / t_now << "s/step)" << std::endl;
      if (iter >= max_iter || t_now > max_time) { go = false; break; }
    }
  }
  if (aux[0].status & ~MH_WORLD_IMPACT_TOL) std::cerr << "regress: world 0 finished with status bits " << aux[0].status << std::endl;
  out << (clock() - start) / (double)CLOCKS_PER_SEC << std::endl;
  out.close();
  return 0;
}
