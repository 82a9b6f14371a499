// moby-hip-compare-trajs <file1> <file2> <tol>: programs/compare-trajs.cpp on mh_io_compare_trajs
#include <cstdlib>
#include <iostream>
#include "../../include/moby_hip_io.h"
int main(int argc, char** argv)
{
  if (argc < 4) return -1;
  double md = 0.0, tm[2] = {0.0, 0.0};
  const int rc = mh_io_compare_trajs(argv[1], argv[2], std::atof(argv[3]), &md, tm);
  if (rc < 0) { std::cerr << mh_io_last_error() << std::endl; return -1; }
  std::cout << "maximum difference: " << md << std::endl;
  std::cout << "reference timing: " << tm[0] << "  new timing: " << tm[1] << std::endl;
  return rc ? -1 : 0;
}
