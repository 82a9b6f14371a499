// Definitions shared by every size variant of the world kernel (included once).
#pragma once
#include "mh_lcp_wave.h"

namespace mh {

struct V3 { double x, y, z; };
MH_DEV V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MH_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MH_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MH_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
MH_DEV V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
MH_DEV V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
MH_DEV double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MH_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MH_DEV double norm(V3 a) { return sqrt(dot(a, a)); }


#define MHW_INF 1.7976931348623157e+308
#define MHW_KA 16      /* upper bound of the LDS LU block edge over the variants (MHW_KA_V) */

// friction polygon directions cos/sin(j/(kh-1) * pi/2), filled by the host's libm
// (ImpactConstraintHandlerQP.cpp:466-468) so that device and oracle agree bit for bit
struct FricTable { double c[33][32]; double s[33][32]; };
__constant__ FricTable c_fric;
__constant__ Pow10Table c_pow10;

} // namespace mh
