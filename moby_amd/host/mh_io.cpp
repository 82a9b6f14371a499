// libmoby_hip_io.so: scene loader + regress-row / compare-trajs helpers (include/moby_hip_io.h).
// Host only (g++ + libxml2).  The loader is written against the attribute semantics of the
// reference's load_from_xml methods (cited per element below); it is a new parser over libxml2's
// tree, not the reference's XMLTree/XMLReader.
#include <libxml/parser.h>
#include <libxml/tree.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <strings.h>
#include <vector>
#include "../../include/moby_hip_io.h"

namespace {

thread_local char g_err[512] = "";
int fail(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); return 1; }

struct Attrs { std::map<std::string, std::string> kv;
  bool has(const char* k) const { return kv.count(k) != 0; }
  const std::string& str(const char* k) const { static const std::string e; auto i = kv.find(k); return i == kv.end() ? e : i->second; } };

Attrs attrs_of(xmlNode* n) {
  Attrs a;
  for (xmlAttr* p = n->properties; p; p = p->next) {
    xmlChar* v = xmlNodeListGetString(n->doc, p->children, 1);
    a.kv[(const char*)p->name] = v ? (const char*)v : "";
    if (v) xmlFree(v);
  }
  return a;
}
// numbers separated by blanks, commas or semicolons (XMLAttrib::get_*_value accept all three)
std::vector<double> numbers(const std::string& s) {
  std::vector<double> v; std::string t = s;
  for (char& c : t) if (c == ',' || c == ';') c = ' ';
  std::istringstream in(t); std::string w;
  while (in >> w) v.push_back(std::atof(w.c_str()));
  return v;
}
bool boolean(const std::string& s) { return !(strcasecmp(s.c_str(), "false") == 0 || s == "0"); }

// Quatd::rpy as Ravelin forms it (half-angle products, ZYX), x y z w: every rpy attribute becomes a quaternion first
void rpy_to_quat(double roll, double pitch, double yaw, double q[4]) {
  const double cr = std::cos(roll * 0.5), sr = std::sin(roll * 0.5), cp = std::cos(pitch * 0.5), sp = std::sin(pitch * 0.5);
  const double cy = std::cos(yaw * 0.5), sy = std::sin(yaw * 0.5);
  q[0] = sr * cp * cy - cr * sp * sy; q[1] = cr * sp * cy + sr * cp * sy; q[2] = cr * cp * sy - sr * sp * cy; q[3] = cr * cp * cy + sr * sp * sy;
}
// rotation matrix of a STATIC pose (plane / primitive poses) in the form Ravelin uses, diagonal 2 (w^2 + q_i^2) - 1: pinned by
// the round-off fingerprint of regress/sphere-stack.dat (moby_amd/scene.py::quat_to_R, tests/test_oracle_world.py)
void quat_to_R_static(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 2 * (w*w + x*x) - 1; R[1] = 2 * (x*y - z*w); R[2] = 2 * (x*z + y*w);
  R[3] = 2 * (x*y + z*w); R[4] = 2 * (w*w + y*y) - 1; R[5] = 2 * (y*z - x*w);
  R[6] = 2 * (x*z - y*w); R[7] = 2 * (y*z + x*w); R[8] = 2 * (w*w + z*z) - 1;
}
void rpy_to_R(double roll, double pitch, double yaw, double R[9]) { double q[4]; rpy_to_quat(roll, pitch, yaw, q); quat_to_R_static(q, R); }
void quat_to_R(const double q[4], double R[9]) {     // q = x y z w, row-major R (the kernels' / oracle's rot())
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y*y + z*z); R[1] = 2 * (x*y - z*w); R[2] = 2 * (x*z + y*w);
  R[3] = 2 * (x*y + z*w); R[4] = 1 - 2 * (x*x + z*z); R[5] = 2 * (y*z - x*w);
  R[6] = 2 * (x*z - y*w); R[7] = 2 * (y*z + x*w); R[8] = 1 - 2 * (x*x + y*y);
}

struct Prim { int type; double dim[3]; double mass; double J[3]; bool posed; double R[9]; double o[3]; };   // type: 0 sphere, 2 box, 100 plane
struct Body { std::string id; bool enabled = true; double x[3] = {0, 0, 0}; double q[4] = {0, 0, 0, 1}; double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
              bool rotated = false; double v[3] = {0, 0, 0}, w[3] = {0, 0, 0}; double mass = 0; double J[3] = {0, 0, 0}; std::string geom; bool has_cg = false; };
struct CP { std::string a, b; double eps = 0, mu = 0, muv = 0, comp = 0; int nk = 4; };

void collect(xmlNode* n, const char* name, std::vector<xmlNode*>& out) {
  for (xmlNode* c = n; c; c = c->next) {
    if (c->type == XML_ELEMENT_NODE) { if (strcmp((const char*)c->name, name) == 0) out.push_back(c); collect(c->children, name, out); }
  }
}
xmlNode* first(xmlNode* root, const char* name) { std::vector<xmlNode*> v; collect(root, name, v); return v.empty() ? nullptr : v[0]; }

// Primitive::load_from_xml (Primitive.cpp:244-300): mass | density, pose
int prim_common(const Attrs& a, Prim& p, double volume, const char* what) {
  p.mass = 0.0;
  if (a.has("mass")) p.mass = std::atof(a.str("mass").c_str());
  else if (a.has("density")) p.mass = std::atof(a.str("density").c_str()) * volume;
  p.posed = false;
  for (int i = 0; i < 9; i++) p.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  p.o[0] = p.o[1] = p.o[2] = 0.0;
  if (a.has("quat") || a.has("aangle")) return fail("%s: quat / aangle poses of primitives are not supported", what);
  if (a.has("rpy")) { const std::vector<double> r = numbers(a.str("rpy")); if (r.size() != 3) return fail("%s: bad rpy", what); rpy_to_R(r[0], r[1], r[2], p.R); p.posed = true; }
  if (a.has("position")) { const std::vector<double> r = numbers(a.str("position")); if (r.size() != 3) return fail("%s: bad position", what);
    for (int i = 0; i < 3; i++) { p.o[i] = r[i]; if (r[i] != 0.0) p.posed = true; } }
  return 0;
}

} // namespace

// ---- SDF (src/SDFReader.cpp) -----------------------------------------------------------------------------
namespace {
xmlNode* child_named(xmlNode* n, const char* name) {
  for (xmlNode* c = n ? n->children : nullptr; c; c = c->next)
    if (c->type == XML_ELEMENT_NODE && strcasecmp((const char*)c->name, name) == 0) return c;
  return nullptr;
}
std::string text_of(xmlNode* n) {
  if (!n) return std::string();
  xmlChar* v = xmlNodeGetContent(n);
  std::string s = v ? (const char*)v : "";
  if (v) xmlFree(v);
  const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return (a == std::string::npos) ? std::string() : s.substr(a, b - a + 1);
}
struct Pose { double R[9]; double x[3]; };
Pose identity_pose() { Pose p; for (int i = 0; i < 9; i++) p.R[i] = (i % 4 == 0) ? 1.0 : 0.0; p.x[0] = p.x[1] = p.x[2] = 0.0; return p; }
// SDFReader::read_pose (SDFReader.cpp:1272-1286): x y z roll pitch yaw, Quatd::rpy
bool read_pose(xmlNode* parent, Pose& p) {
  xmlNode* n = child_named(parent, "pose");
  p = identity_pose();
  if (!n) return true;
  const std::vector<double> v = numbers(text_of(n));
  if (v.size() != 6) return false;
  for (int i = 0; i < 3; i++) p.x[i] = v[i];
  rpy_to_R(v[3], v[4], v[5], p.R);
  return true;
}
void mat3mul(const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3*i+j] = A[3*i] * B[j] + A[3*i+1] * B[3+j] + A[3*i+2] * B[6+j]; }
void mat3Tmul(const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3*i+j] = A[i] * B[j] + A[3+i] * B[3+j] + A[6+i] * B[6+j]; }
void mat3vec(const double* A, const double* v, double* y) { for (int i = 0; i < 3; i++) y[i] = A[3*i] * v[0] + A[3*i+1] * v[1] + A[3*i+2] * v[2]; }
void mat3Tvec(const double* A, const double* v, double* y) { for (int i = 0; i < 3; i++) y[i] = A[i] * v[0] + A[3+i] * v[1] + A[6+i] * v[2]; }
struct SdfLink { std::string name; Pose pose; Pose inertial; double mass; double I[9]; };
struct SdfJoint { std::string name, parent, child; int type; double xyz[3]; bool parent_frame; double lo, hi; };
}  // namespace

extern "C" int mh_io_load_sdf(const char* path, const double gravity[3], mh_io_artic* out)
{
  if (!path || !out) return fail("null argument");
  xmlDoc* doc = xmlReadFile(path, nullptr, XML_PARSE_NONET | XML_PARSE_NOERROR | XML_PARSE_NOWARNING);
  if (!doc) return fail("cannot parse %s", path);
  struct Guard { xmlDoc* d; ~Guard() { xmlFreeDoc(d); } } guard{doc};
  std::vector<xmlNode*> models; collect(xmlDocGetRootElement(doc), "model", models);
  if (models.size() != 1) return fail("%s: expected exactly one <model>, found %zu", path, models.size());
  xmlNode* model = models[0];
  std::vector<SdfLink> links; std::vector<SdfJoint> joints;
  for (xmlNode* c = model->children; c; c = c->next) {
    if (c->type != XML_ELEMENT_NODE) continue;
    if (strcasecmp((const char*)c->name, "link") == 0) {                       // read_link / read_inertial
      SdfLink L; L.name = attrs_of(c).str("name");
      if (!read_pose(c, L.pose)) return fail("link %s: bad <pose>", L.name.c_str());
      xmlNode* in = child_named(c, "inertial");
      if (!in) return fail("link %s: no <inertial>", L.name.c_str());
      if (!read_pose(in, L.inertial)) return fail("link %s: bad inertial <pose>", L.name.c_str());
      L.mass = std::atof(text_of(child_named(in, "mass")).c_str());
      xmlNode* im = child_named(in, "inertia");
      if (!im) return fail("link %s: no <inertia>", L.name.c_str());
      auto g = [&](const char* k) { return std::atof(text_of(child_named(im, k)).c_str()); };
      const double ixx = g("ixx"), ixy = g("ixy"), ixz = g("ixz"), iyy = g("iyy"), iyz = g("iyz"), izz = g("izz");
      const double I[9] = { ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz };
      // the tensor is given in the inertial frame: bring it into the link's axes (about the COM)
      double T[9]; mat3mul(L.inertial.R, I, T);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L.I[3*i+j] = T[3*i] * L.inertial.R[3*j] + T[3*i+1] * L.inertial.R[3*j+1] + T[3*i+2] * L.inertial.R[3*j+2];
      links.push_back(L);
    } else if (strcasecmp((const char*)c->name, "joint") == 0) {                // read_joint
      SdfJoint J; const Attrs a = attrs_of(c); J.name = a.str("name");
      std::string type = a.str("type"); for (char& ch : type) ch = (char)tolower(ch);
      if (type == "revolute") J.type = MH_JOINT_REVOLUTE; else if (type == "prismatic") J.type = MH_JOINT_PRISMATIC;
      else return fail("joint %s: type '%s' is not supported (revolute, prismatic)", J.name.c_str(), type.c_str());
      J.parent = text_of(child_named(c, "parent")); J.child = text_of(child_named(c, "child"));
      if (child_named(c, "pose")) return fail("joint %s: a <pose> under <joint> is not supported", J.name.c_str());
      xmlNode* ax = child_named(c, "axis");
      if (!ax) return fail("joint %s: no <axis>", J.name.c_str());
      const std::vector<double> v = numbers(text_of(child_named(ax, "xyz")));
      if (v.size() != 3) return fail("joint %s: bad <xyz>", J.name.c_str());
      for (int i = 0; i < 3; i++) J.xyz[i] = v[i];
      xmlNode* pf = child_named(ax, "use_parent_model_frame");
      J.parent_frame = pf && boolean(text_of(pf));
      J.lo = -1.7976931348623157e308; J.hi = 1.7976931348623157e308;            // Joint::lolimit / hilimit defaults (Joint.cpp:33-54)
      if (xmlNode* lim = child_named(ax, "limit")) {
        if (xmlNode* l = child_named(lim, "lower")) J.lo = std::atof(text_of(l).c_str());
        if (xmlNode* u = child_named(lim, "upper")) J.hi = std::atof(text_of(u).c_str());
      }
      // SDFReader::read_joint keeps <dynamics><damping> / <friction> as Joint::mu_fv / mu_fc (SDFReader.cpp:575-588); joint friction is not
      // part of this build's forward dynamics, so a model that asks for it is refused rather than simulated without it
      if (xmlNode* dyn = child_named(ax, "dynamics")) {
        for (const char* tag : { "damping", "friction" })
          if (xmlNode* d = child_named(dyn, tag)) if (std::atof(text_of(d).c_str()) != 0.0)
            return fail("joint %s: <dynamics><%s> = %s: joint friction / damping is not supported", J.name.c_str(), tag, text_of(d).c_str());
      }
      if (child_named(c, "axis2")) return fail("joint %s: a second axis (<axis2>) is not supported", J.name.c_str());
      joints.push_back(J);
    }
  }
  if (joints.empty()) return fail("%s: a model without joints is a single rigid body, not an articulated one", path);
  if ((int)joints.size() > MH_ARTIC_MAX_JOINTS) return fail("%zu joints > %d", joints.size(), MH_ARTIC_MAX_JOINTS);
  std::map<std::string, int> link_of;
  for (size_t i = 0; i < links.size(); i++) link_of[links[i].name] = (int)i;
  // parents first: repeatedly take, in file order, the joints whose parent is the world or already placed
  std::vector<int> order; std::map<std::string, int> joint_of_link;             // child link -> position in `order`
  std::vector<char> used(joints.size(), 0);
  bool fixed_base = false;
  while (order.size() < joints.size()) {
    bool progress = false;
    for (size_t j = 0; j < joints.size(); j++) {
      if (used[j]) continue;
      const bool world = strcasecmp(joints[j].parent.c_str(), "world") == 0;
      if (!world && !joint_of_link.count(joints[j].parent)) continue;
      if (!link_of.count(joints[j].child)) return fail("joint %s: child link '%s' not found", joints[j].name.c_str(), joints[j].child.c_str());
      if (joint_of_link.count(joints[j].child)) return fail("link %s is the child of two joints (closed chains are not supported)", joints[j].child.c_str());
      if (world) fixed_base = true;
      joint_of_link[joints[j].child] = (int)order.size();
      order.push_back((int)j); used[j] = 1; progress = true;
    }
    if (!progress) return fail("%s: joints that do not hang from the world (floating bases are not supported)", path);
  }
  if (!fixed_base) return fail("%s: no joint attaches the model to the world", path);
  if (joint_of_link.size() != links.size()) return fail("%s: %zu links but %zu are carried by joints", path, links.size(), joint_of_link.size());
  std::memset(out, 0, sizeof(*out));
  mh_artic_model& m = out->model;
  m.cstab_eps = std::sqrt(2.220446049250313e-16);          // (stabilisation itself off: cstab_max_iterations = 0, as example/ur10/ur10.xml:11)
  m.nj = (int)order.size();
  for (int k = 0; k < 3; k++) m.gravity[k] = gravity ? gravity[k] : 0.0;
  for (int i = 0; i < m.nj; i++) {
    const SdfJoint& J = joints[order[i]];
    const SdfLink& L = links[link_of[J.child]];
    const bool world = strcasecmp(J.parent.c_str(), "world") == 0;
    const Pose Pp = world ? identity_pose() : links[link_of[J.parent]].pose;
    m.parent[i] = world ? -1 : joint_of_link[J.parent];
    m.jtype[i] = J.type;
    mat3Tmul(Pp.R, L.pose.R, m.Rrel[i]);                                        // R_p' R_c
    const double d[3] = { L.pose.x[0] - Pp.x[0], L.pose.x[1] - Pp.x[1], L.pose.x[2] - Pp.x[2] };
    mat3Tvec(Pp.R, d, m.trel[i]);
    double ag[3];                                                                // the axis in the model frame at q = 0 ...
    if (J.parent_frame) mat3vec(Pp.R, J.xyz, ag); else mat3vec(L.pose.R, J.xyz, ag);
    double al[3]; mat3Tvec(L.pose.R, ag, al);                                    // ... and in the child link's frame
    const double nrm = std::sqrt(al[0]*al[0] + al[1]*al[1] + al[2]*al[2]);
    if (!(nrm > 0.0)) return fail("joint %s: zero axis", J.name.c_str());
    for (int k = 0; k < 3; k++) m.axis[i][k] = al[k] / nrm;
    for (int k = 0; k < 3; k++) m.com[i][k] = L.inertial.x[k];
    for (int k = 0; k < 9; k++) m.inertia[i][k] = L.I[k];
    m.mass[i] = L.mass;
    m.lolimit[i] = J.lo; m.hilimit[i] = J.hi; m.limit_restitution[i] = 0.0;
    snprintf(out->link_id[i], MH_IO_ID_LEN, "%s", L.name.c_str());
    snprintf(out->joint_id[i], MH_IO_ID_LEN, "%s", J.name.c_str());
  }
  return 0;
}


// ---- URDF (src/URDFReader.cpp) ----------------------------------------------------------------------------
namespace {
struct UrdfLink { std::string name; bool has_inertial = false; bool bad_inertia = false; Pose inertial; double mass = 0.0; double I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                  int cg = 0; Pose cg_pose; double cg_radius = 0.0; std::string cg_kind; };      // cg: 0 none, 1 sphere, 2 box / cylinder
struct UrdfJoint { std::string name, parent, child; int type = 0; Pose origin; double axis[3] = {1.0, 0.0, 0.0}; double lo = 0.0, hi = 0.0; };
constexpr int URDF_FIXED = 2;
// URDFReader::read_origin (URDFReader.cpp:943-972): the first <origin>, xyz and rpy attributes, Quatd::rpy
bool urdf_origin(xmlNode* parent, Pose& p) {
  p = identity_pose();
  xmlNode* n = child_named(parent, "origin");
  if (!n) return true;
  const Attrs a = attrs_of(n);
  if (a.has("xyz")) { const std::vector<double> v = numbers(a.str("xyz")); if (v.size() != 3) return false; for (int i = 0; i < 3; i++) p.x[i] = v[i]; }
  if (a.has("rpy")) { const std::vector<double> v = numbers(a.str("rpy")); if (v.size() != 3) return false; rpy_to_R(v[0], v[1], v[2], p.R); }
  return true;
}
// the robot as a fixed-base tree of 1-DOF joints; a FixedJoint's outboard link rides on the link that carries it
struct UrdfGeom { int link; std::string id; bool sphere; double center[3]; double radius; std::string kind; };   // link -1: the base
struct UrdfRobot { std::string name, base; mh_io_artic art; std::vector<UrdfGeom> geoms; std::vector<std::string> link_names; };

int parse_urdf(const char* path, UrdfRobot& rob)
{
  xmlDoc* doc = xmlReadFile(path, nullptr, XML_PARSE_NONET | XML_PARSE_NOERROR | XML_PARSE_NOWARNING);
  if (!doc) return fail("cannot parse %s", path);
  struct Guard { xmlDoc* d; ~Guard() { xmlFreeDoc(d); } } guard{doc};
  xmlNode* root = xmlDocGetRootElement(doc);
  if (!root || strcasecmp((const char*)root->name, "robot") != 0) return fail("%s: the root element of a URDF file is <robot>", path);   // URDFReader.cpp:124-135
  { const Attrs a = attrs_of(root); if (!a.has("name")) return fail("%s: <robot> without a name", path); rob.name = a.str("name"); }   // :147-160
  std::vector<UrdfLink> links; std::vector<UrdfJoint> joints;
  for (xmlNode* c = root->children; c; c = c->next) {
    if (c->type != XML_ELEMENT_NODE) continue;
    const Attrs a = attrs_of(c);
    if (strcasecmp((const char*)c->name, "link") == 0) {                        // read_link / read_inertial / read_collision (:180-203, 599-636, 781-811)
      UrdfLink L; if (!a.has("name")) return fail("%s: a <link> without a name", path);
      L.name = a.str("name"); L.inertial = identity_pose(); L.cg_pose = identity_pose();
      if (xmlNode* in = child_named(c, "inertial")) {
        L.has_inertial = true;
        if (!urdf_origin(in, L.inertial)) return fail("link %s: bad inertial <origin>", L.name.c_str());
        if (xmlNode* mn = child_named(in, "mass")) { const Attrs ma = attrs_of(mn); if (ma.has("value")) L.mass = std::atof(ma.str("value").c_str()); }
        double I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (xmlNode* im = child_named(in, "inertia")) { const Attrs ia = attrs_of(im);
          auto g = [&](const char* k) { return ia.has(k) ? std::atof(ia.str(k).c_str()) : 0.0; };
          I[0] = g("ixx"); I[4] = g("iyy"); I[8] = g("izz"); I[1] = I[3] = g("ixy"); I[2] = I[6] = g("ixz"); I[5] = I[7] = g("iyz"); }
        // read_inertial (:613-616) disables a link whose tensor is not SPD just as it disables one without mass; this build has no disabled links
        // (a massless link is refused below), so the same refusal: Sylvester's criterion on the tensor as given
        { const double d1 = I[0], d2 = I[0] * I[4] - I[1] * I[3],
                       d3 = I[0] * (I[4] * I[8] - I[5] * I[7]) - I[1] * (I[3] * I[8] - I[5] * I[6]) + I[2] * (I[3] * I[7] - I[4] * I[6]);
          if (child_named(in, "inertia") && !(d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) L.bad_inertia = true; }
        double Tm[9]; mat3mul(L.inertial.R, I, Tm);                               // the tensor is given in the inertial frame: into the link's axes
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L.I[3*i+j] = Tm[3*i] * L.inertial.R[3*j] + Tm[3*i+1] * L.inertial.R[3*j+1] + Tm[3*i+2] * L.inertial.R[3*j+2];
      }
      if (xmlNode* col = child_named(c, "collision")) {                            // the first <collision> only, as the reference
        if (xmlNode* ge = child_named(col, "geometry")) {
          xmlNode* sp = child_named(ge, "sphere");
          // read_primitive tries box, cylinder, sphere in that order (:813-836)
          if (child_named(ge, "box")) { L.cg = 2; L.cg_kind = "box"; }
          else if (child_named(ge, "cylinder")) { L.cg = 2; L.cg_kind = "cylinder"; }
          else if (sp && attrs_of(sp).has("radius")) { L.cg = 1; L.cg_kind = "sphere"; L.cg_radius = std::atof(attrs_of(sp).str("radius").c_str()); }
          if (L.cg && !urdf_origin(col, L.cg_pose)) return fail("link %s: bad collision <origin>", L.name.c_str());
        }
      }
      links.push_back(L);
    } else if (strcasecmp((const char*)c->name, "joint") == 0) {                 // read_joint (:299-403)
      UrdfJoint J; if (!a.has("name") || !a.has("type")) return fail("%s: a <joint> needs a name and a type", path);
      J.name = a.str("name");
      const std::string type = a.str("type");
      if (strcasecmp(type.c_str(), "revolute") == 0) { J.type = MH_JOINT_REVOLUTE; J.lo = -M_PI_2; J.hi = M_PI_2; }
      else if (strcasecmp(type.c_str(), "continuous") == 0) { J.type = MH_JOINT_REVOLUTE; J.lo = -10000.0; J.hi = 10000.0; }
      else if (strcasecmp(type.c_str(), "prismatic") == 0) { J.type = MH_JOINT_PRISMATIC; J.lo = -10000.0; J.hi = 10000.0; }
      else if (strcasecmp(type.c_str(), "fixed") == 0) J.type = URDF_FIXED;
      else return fail("joint %s: type '%s' is not supported (revolute, continuous, prismatic, fixed; the reference drops floating and planar joints too)", J.name.c_str(), type.c_str());
      xmlNode* pn = child_named(c, "parent"); xmlNode* cn = child_named(c, "child");
      if (!pn || !attrs_of(pn).has("link")) return fail("joint %s: no <parent link=...>", J.name.c_str());
      if (!cn || !attrs_of(cn).has("link")) return fail("joint %s: no <child link=...>", J.name.c_str());
      J.parent = attrs_of(pn).str("link"); J.child = attrs_of(cn).str("link");
      if (!urdf_origin(c, J.origin)) return fail("joint %s: bad <origin>", J.name.c_str());
      if (xmlNode* ax = child_named(c, "axis")) { const Attrs xa = attrs_of(ax);  // read_axis (:456-496): default (1, 0, 0), in the joint's frame
        if (xa.has("xyz")) { const std::vector<double> v = numbers(xa.str("xyz")); if (v.size() != 3) return fail("joint %s: bad <axis>", J.name.c_str()); for (int i = 0; i < 3; i++) J.axis[i] = v[i]; } }
      if (J.type != URDF_FIXED)                                                 // read_limits (:532-566): the FIRST <limit> child that carries effort, lower or upper
        for (xmlNode* lim = c->children; lim; lim = lim->next) {                // (one with a velocity only is passed over and the search goes on)
          if (lim->type != XML_ELEMENT_NODE || strcasecmp((const char*)lim->name, "limit") != 0) continue;
          const Attrs la = attrs_of(lim);
          if (!(la.has("effort") || la.has("lower") || la.has("upper"))) continue;
          if (la.has("lower")) J.lo = std::atof(la.str("lower").c_str());
          if (la.has("upper")) J.hi = std::atof(la.str("upper").c_str());
          break;                                                                  // "multiple tags unsupported" (:561-562)
        }
      // <dynamics damping friction> become Joint::mu_fv / mu_fc (:499-529), which only MCArticulatedBody reads (MCArticulatedBody.cpp:419-420,
      // 564-565): a reduced-coordinate body steps without them in the reference, and so here
      joints.push_back(J);
    }
  }
  if (links.empty()) return fail("%s: no links", path);
  std::map<std::string, int> link_of; for (size_t i = 0; i < links.size(); i++) { if (link_of.count(links[i].name)) return fail("%s: two links named %s", path, links[i].name.c_str()); link_of[links[i].name] = (int)i; }
  std::map<std::string, int> inner;                                               // child link -> its joint
  for (size_t j = 0; j < joints.size(); j++) {
    if (!link_of.count(joints[j].parent) || !link_of.count(joints[j].child)) return fail("joint %s: unknown link", joints[j].name.c_str());
    if (inner.count(joints[j].child)) return fail("link %s is the child of two joints (closed chains are not supported)", joints[j].child.c_str());
    inner[joints[j].child] = (int)j;
  }
  int nbase = 0; for (const UrdfLink& L : links) if (!inner.count(L.name)) { rob.base = L.name; nbase++; }
  if (nbase != 1) return fail("%s: %d links are carried by no joint (exactly one base link expected)", path, nbase);
  for (const UrdfLink& L : links) if (L.bad_inertia && L.name != rob.base) return fail("link %s: the inertia tensor is not positive definite (the reference disables such a link, URDFReader.cpp:613-616)", L.name.c_str());
  // parents first, file order within a level.  carrier[l]: the model link whose frame link l is rigidly attached to (-1 = base) + its pose there
  struct Placed { int carrier; Pose pose; };
  std::map<std::string, Placed> placed; placed[rob.base] = Placed{ -1, identity_pose() };
  struct Part { double mass; double com[3]; double I[9]; };
  std::vector<std::vector<Part> > parts;                                          // per model link: its own inertia, then what fixed joints hang on it
  std::memset(&rob.art, 0, sizeof(rob.art));
  mh_artic_model& m = rob.art.model;
  std::vector<char> used(joints.size(), 0);
  size_t done = 0;
  auto add_part = [&](int carrier, const Pose& P, const UrdfLink& L) {
    if (carrier < 0 || !L.has_inertial) return;
    Part p; p.mass = L.mass;
    double c[3]; mat3vec(P.R, L.inertial.x, c); for (int k = 0; k < 3; k++) p.com[k] = c[k] + P.x[k];
    double Tm[9]; mat3mul(P.R, L.I, Tm);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p.I[3*i+j] = Tm[3*i] * P.R[3*j] + Tm[3*i+1] * P.R[3*j+1] + Tm[3*i+2] * P.R[3*j+2];
    parts[carrier].push_back(p);
  };
  auto add_geom = [&](int carrier, const Pose& P, const UrdfLink& L) {
    if (!L.cg) return;
    UrdfGeom g; g.link = carrier; g.id = L.name; g.sphere = (L.cg == 1); g.radius = L.cg_radius; g.kind = L.cg_kind;
    double c[3]; mat3vec(P.R, L.cg_pose.x, c); for (int k = 0; k < 3; k++) g.center[k] = c[k] + P.x[k];
    rob.geoms.push_back(g);
  };
  add_geom(-1, identity_pose(), links[link_of[rob.base]]);
  while (done < joints.size()) {
    bool progress = false;
    for (size_t j = 0; j < joints.size(); j++) {
      if (used[j] || !placed.count(joints[j].parent)) continue;
      const UrdfJoint& J = joints[j]; const Placed& PP = placed[J.parent]; const UrdfLink& L = links[link_of[J.child]];
      Pose F;                                                                      // the joint's (= the child link's) frame in the carrier's frame
      mat3mul(PP.pose.R, J.origin.R, F.R);
      { double tt[3]; mat3vec(PP.pose.R, J.origin.x, tt); for (int k = 0; k < 3; k++) F.x[k] = tt[k] + PP.pose.x[k]; }
      if (J.type == URDF_FIXED) {
        placed[J.child] = Placed{ PP.carrier, F };
        add_part(PP.carrier, F, L); add_geom(PP.carrier, F, L);
      } else {
        if (m.nj >= MH_ARTIC_MAX_JOINTS) return fail("%s: more than %d moving joints", path, MH_ARTIC_MAX_JOINTS);
        const int i = m.nj++;
        m.parent[i] = PP.carrier; m.jtype[i] = J.type;
        for (int k = 0; k < 9; k++) m.Rrel[i][k] = F.R[k];
        for (int k = 0; k < 3; k++) m.trel[i][k] = F.x[k];
        const double nrm = std::sqrt(J.axis[0]*J.axis[0] + J.axis[1]*J.axis[1] + J.axis[2]*J.axis[2]);
        if (!(nrm > 0.0)) return fail("joint %s: zero axis", J.name.c_str());
        for (int k = 0; k < 3; k++) m.axis[i][k] = J.axis[k] / nrm;
        m.lolimit[i] = J.lo; m.hilimit[i] = J.hi; m.limit_restitution[i] = 0.0;
        snprintf(rob.art.link_id[i], MH_IO_ID_LEN, "%s", L.name.c_str());
        snprintf(rob.art.joint_id[i], MH_IO_ID_LEN, "%s", J.name.c_str());
        parts.push_back(std::vector<Part>());
        placed[J.child] = Placed{ i, identity_pose() };
        add_part(i, identity_pose(), L); add_geom(i, identity_pose(), L);
      }
      used[j] = 1; done++; progress = true;
    }
    if (!progress) return fail("%s: joints that do not hang from the base link %s", path, rob.base.c_str());
  }
  if (m.nj == 0) return fail("%s: no moving joint: a single rigid body, not an articulated one", path);
  for (int i = 0; i < m.nj; i++) {
    const std::vector<Part>& ps = parts[i];
    if (ps.size() == 1) {                                                          // the link alone: its numbers as the file states them
      m.mass[i] = ps[0].mass; for (int k = 0; k < 3; k++) m.com[i][k] = ps[0].com[k]; for (int k = 0; k < 9; k++) m.inertia[i][k] = ps[0].I[k];
    } else {                                                                       // + the links fixed to it: one rigid body (parallel-axis theorem about the common COM)
      double M = 0.0, c[3] = { 0.0, 0.0, 0.0 };
      for (const Part& p : ps) { M += p.mass; for (int k = 0; k < 3; k++) c[k] += p.mass * p.com[k]; }
      if (M > 0.0) for (int k = 0; k < 3; k++) c[k] /= M;
      double I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (const Part& p : ps) { const double d[3] = { p.com[0] - c[0], p.com[1] - c[1], p.com[2] - c[2] }; const double dd = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) I[3*r+q] += p.I[3*r+q] + p.mass * (((r == q) ? dd : 0.0) - d[r] * d[q]); }
      m.mass[i] = M; for (int k = 0; k < 3; k++) m.com[i][k] = c[k]; for (int k = 0; k < 9; k++) m.inertia[i][k] = I[k];
    }
    // URDFReader::read_inertial disables a link without mass or with a tensor that is not positive definite (:613-616); a disabled link
    // inside the tree has no dynamics in the reference either, so the model is refused
    if (!(m.mass[i] > 0.0)) return fail("link %s: no mass (the reference disables such a link)", rob.art.link_id[i]);
  }
  for (const UrdfLink& L : links) rob.link_names.push_back(L.name);
  m.cstab_eps = std::sqrt(2.220446049250313e-16);
  return 0;
}
}  // namespace

extern "C" int mh_io_load_urdf(const char* path, const double gravity[3], mh_io_artic* out)
{
  if (!path || !out) return fail("null argument");
  UrdfRobot rob;
  if (parse_urdf(path, rob)) return 1;
  *out = rob.art;
  for (int k = 0; k < 3; k++) out->model.gravity[k] = gravity ? gravity[k] : 0.0;
  return 0;
}


extern "C" {

const char* mh_io_last_error(void) { return g_err; }

int mh_io_load_xml(const char* path, mh_io_scene* out)
{
  if (!path || !out) return fail("null argument");
  xmlDoc* doc = xmlReadFile(path, nullptr, XML_PARSE_NONET | XML_PARSE_NOERROR | XML_PARSE_NOWARNING);
  if (!doc) return fail("cannot parse %s", path);
  struct Guard { xmlDoc* d; ~Guard() { xmlFreeDoc(d); } } guard{doc};
  xmlNode* root = xmlDocGetRootElement(doc);
  std::memset(out, 0, sizeof(*out));
  mh_scene& sc = out->scene;
  // defaults (mh_scene_defaults, duplicated here so that this library does not need the HIP one)
  sc.min_step_size = std::sqrt(2.220446049250313e-16); sc.contact_dist_thresh = 1e-6; sc.cstab_eps = std::sqrt(2.220446049250313e-16);
  sc.cstab_max_iterations = MH_CSTAB_DEFAULT_MAX_ITERATIONS; sc.plane_R[0] = sc.plane_R[4] = sc.plane_R[8] = 1.0;
  for (int p = 0; p < MH_MAX_PAIRS; p++) { sc.pair_enabled[p] = 1; sc.cp_nk[p] = 4; }

  // ---- primitives ----
  std::map<std::string, Prim> prims;
  { std::vector<xmlNode*> v; collect(root, "Sphere", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = MH_GEOM_SPHERE;
      const double r = a.has("radius") ? std::atof(a.str("radius").c_str()) : 0.0;   // SpherePrimitive.cpp:368
      p.dim[0] = r; p.dim[1] = p.dim[2] = 0.0;
      if (prim_common(a, p, M_PI * r * r * r * 4.0 / 3.0, "Sphere")) return 1;       // SpherePrimitive.cpp:144-146
      const double j = r * r * p.mass * 2.0 / 5.0;                                  // SpherePrimitive.cpp:149
      p.J[0] = p.J[1] = p.J[2] = j;
      prims[a.str("id")] = p; } }
  { std::vector<xmlNode*> v; collect(root, "Box", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = MH_GEOM_BOX;
      const double x = std::atof(a.str("xlen").c_str()), y = std::atof(a.str("ylen").c_str()), z = std::atof(a.str("zlen").c_str());
      p.dim[0] = x; p.dim[1] = y; p.dim[2] = z;
      if (prim_common(a, p, x * y * z, "Box")) return 1;                             // BoxPrimitive.cpp:692-712
      const double M = p.mass / 12.0;
      p.J[0] = M * (y * y + z * z); p.J[1] = M * (x * x + z * z); p.J[2] = M * (x * x + y * y);
      prims[a.str("id")] = p; } }
  { std::vector<xmlNode*> v; collect(root, "Plane", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = 100; p.dim[0] = p.dim[1] = p.dim[2] = 0.0; p.J[0] = p.J[1] = p.J[2] = 0.0;
      if (prim_common(a, p, 0.0, "Plane")) return 1;
      prims[a.str("id")] = p; } }
  // ---- gravity (GravityForce.cpp:74-90) ----
  std::map<std::string, std::vector<double> > gravs;
  { std::vector<xmlNode*> v; collect(root, "GravityForce", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); std::vector<double> g = numbers(a.str("accel")); if (g.size() != 3) return fail("GravityForce %s: bad accel", a.str("id").c_str()); gravs[a.str("id")] = g; } }
  // ---- collision detection plugin: only the rimless wheel's ----
  std::map<std::string, std::string> plugins;
  { std::vector<xmlNode*> v; collect(root, "CollisionDetectionPlugin", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); plugins[a.str("id")] = a.str("plugin"); } }
  // ---- rigid bodies (RigidBody.cpp:132-369) ----
  std::map<std::string, Body> bodies;
  { std::vector<xmlNode*> v; collect(root, "RigidBody", v);
    for (xmlNode* n : v) {
      const Attrs a = attrs_of(n); Body b; b.id = a.str("id");
      if (a.has("enabled")) b.enabled = boolean(a.str("enabled"));
      if (a.has("compliant") && boolean(a.str("compliant"))) return fail("RigidBody %s: compliant bodies are not supported", b.id.c_str());
      if (a.has("articulated-body-id")) return fail("RigidBody %s: articulated bodies are not supported", b.id.c_str());
      if (a.has("mass")) b.mass = std::atof(a.str("mass").c_str());
      if (a.has("inertia")) { const std::vector<double> J = numbers(a.str("inertia")); if (J.size() != 9) return fail("RigidBody %s: inertia needs 9 numbers", b.id.c_str());
        if (J[1] != 0 || J[2] != 0 || J[3] != 0 || J[5] != 0 || J[6] != 0 || J[7] != 0) return fail("RigidBody %s: only diagonal body-frame inertias are supported", b.id.c_str());
        b.J[0] = J[0]; b.J[1] = J[4]; b.J[2] = J[8]; }
      if (a.has("position")) { const std::vector<double> p = numbers(a.str("position")); if (p.size() != 3) return fail("RigidBody %s: bad position", b.id.c_str()); for (int i = 0; i < 3; i++) b.x[i] = p[i]; }
      if (a.has("quat")) { const std::vector<double> q = numbers(a.str("quat")); if (q.size() != 4) return fail("RigidBody %s: bad quat", b.id.c_str());
        // the attribute is written and read as w x y z (XMLTree.cpp:89-94, 407-419); the state keeps x y z w
        const double nrm = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
        if (!(nrm > 0.0)) return fail("RigidBody %s: zero quat", b.id.c_str());
        b.q[0] = q[1] / nrm; b.q[1] = q[2] / nrm; b.q[2] = q[3] / nrm; b.q[3] = q[0] / nrm;
        quat_to_R(b.q, b.R);
        b.rotated = !(b.q[0] == 0.0 && b.q[1] == 0.0 && b.q[2] == 0.0); }
      else if (a.has("rpy")) { const std::vector<double> r = numbers(a.str("rpy")); if (r.size() != 3) return fail("RigidBody %s: bad rpy", b.id.c_str()); rpy_to_quat(r[0], r[1], r[2], b.q); quat_to_R_static(b.q, b.R); b.rotated = (r[0] != 0 || r[1] != 0 || r[2] != 0); }
      else if (a.has("aangle")) {                                    // axis x y z, angle (RigidBody.cpp:213-219)
        const std::vector<double> r = numbers(a.str("aangle")); if (r.size() != 4) return fail("RigidBody %s: bad aangle", b.id.c_str());
        const double nrm = std::sqrt(r[0]*r[0] + r[1]*r[1] + r[2]*r[2]);
        if (!(nrm > 0.0)) return fail("RigidBody %s: aangle with a zero axis", b.id.c_str());
        const double sh = std::sin(0.5 * r[3]), ch = std::cos(0.5 * r[3]);
        b.q[0] = r[0] / nrm * sh; b.q[1] = r[1] / nrm * sh; b.q[2] = r[2] / nrm * sh; b.q[3] = ch;
        quat_to_R(b.q, b.R);
        b.rotated = (r[3] != 0.0);
      }
      if (a.has("linear-velocity")) { const std::vector<double> p = numbers(a.str("linear-velocity")); if (p.size() != 3) return fail("RigidBody %s: bad linear-velocity", b.id.c_str()); for (int i = 0; i < 3; i++) b.v[i] = p[i]; }
      if (a.has("angular-velocity")) { const std::vector<double> p = numbers(a.str("angular-velocity")); if (p.size() != 3) return fail("RigidBody %s: bad angular-velocity", b.id.c_str()); for (int i = 0; i < 3; i++) b.w[i] = p[i]; }
      bool from_prim = false; double m = 0, J[3] = {0, 0, 0};
      for (xmlNode* c = n->children; c; c = c->next) {
        if (c->type != XML_ELEMENT_NODE) continue;
        const Attrs ca = attrs_of(c);
        if (strcmp((const char*)c->name, "CollisionGeometry") == 0) {
          if (b.has_cg) return fail("RigidBody %s: more than one CollisionGeometry", b.id.c_str());
          b.has_cg = true; b.geom = ca.str("primitive-id");
          if (ca.has("rel-origin") || ca.has("rel-rpy") || ca.has("rel-quat")) return fail("RigidBody %s: offset collision geometries are not supported", b.id.c_str());
        } else if (strcmp((const char*)c->name, "InertiaFromPrimitive") == 0) {           // RigidBody.cpp:246-321: additive, from zero
          auto it = prims.find(ca.str("primitive-id"));
          if (it == prims.end()) return fail("RigidBody %s: unknown primitive %s", b.id.c_str(), ca.str("primitive-id").c_str());
          if (ca.has("relative-origin") || ca.has("relative-rpy") || ca.has("relative-aangle")) return fail("RigidBody %s: offset inertias are not supported", b.id.c_str());
          from_prim = true; m += it->second.mass; for (int i = 0; i < 3; i++) J[i] += it->second.J[i];
        }
      }
      if (from_prim) { b.mass = m; for (int i = 0; i < 3; i++) b.J[i] = J[i]; }
      bodies[b.id] = b;
    } }
  // ---- the simulator (ConstraintSimulator.cpp:540-708, TimeSteppingSimulator.cpp:463-476, Simulator.cpp:826-950) ----
  xmlNode* sim = first(root, "TimeSteppingSimulator");
  if (!sim) return fail("%s: no TimeSteppingSimulator", path);
  if (first(root, "RCArticulatedBody") || first(root, "MCArticulatedBody")) return fail("articulated bodies are not supported");
  const Attrs sa = attrs_of(sim);
  if (sa.has("min-step-size")) sc.min_step_size = std::atof(sa.str("min-step-size").c_str());
  if (sa.has("contact-dist-thresh")) sc.contact_dist_thresh = std::atof(sa.str("contact-dist-thresh").c_str());
  if (sa.has("unilateral-stabilization-tol")) sc.cstab_eps = std::atof(sa.str("unilateral-stabilization-tol").c_str());
  if (sa.has("constraint-stabilization-max-iterations")) sc.cstab_max_iterations = (unsigned)std::strtoul(sa.str("constraint-stabilization-max-iterations").c_str(), nullptr, 10);
  bool wheel_plugin = false;
  if (sa.has("collision-detection-plugin")) {
    const std::string& pl = plugins[sa.str("collision-detection-plugin")];
    if (pl.find("rimless-wheel-coldet-plugin") == std::string::npos) return fail("collision detection plugin '%s' is not supported", pl.c_str());
    wheel_plugin = true;
  }
  std::vector<std::string> dyn; std::string ground; std::vector<CP> cps; std::vector<std::pair<std::string, std::string> > disabled;
  bool have_g = false; double g[3] = {0, 0, 0};
  for (xmlNode* c = sim->children; c; c = c->next) {
    if (c->type != XML_ELEMENT_NODE) continue;
    const Attrs ca = attrs_of(c); const char* nm = (const char*)c->name;
    if (strcmp(nm, "DynamicBody") == 0) {
      auto it = bodies.find(ca.str("dynamic-body-id"));
      if (it == bodies.end()) return fail("simulator: unknown body %s", ca.str("dynamic-body-id").c_str());
      if (it->second.enabled) dyn.push_back(it->first);
      else { if (!ground.empty()) return fail("more than one disabled body"); ground = it->first; }
    } else if (strcmp(nm, "RecurrentForce") == 0) {
      auto it = gravs.find(ca.str("recurrent-force-id"));
      if (it == gravs.end()) return fail("simulator: only GravityForce recurrent forces are supported (%s)", ca.str("recurrent-force-id").c_str());
      if (have_g) return fail("more than one gravity force");
      have_g = true; for (int i = 0; i < 3; i++) g[i] = it->second[i];
    } else if (strcmp(nm, "DisabledPair") == 0) disabled.push_back(std::make_pair(ca.str("object1-id"), ca.str("object2-id")));
    else if (strcmp(nm, "ContactParameters") == 0) {                                   // ContactParameters.cpp:46-135
      CP p; p.a = ca.str("object1-id"); p.b = ca.str("object2-id");
      if (ca.has("epsilon")) p.eps = std::atof(ca.str("epsilon").c_str());
      if (ca.has("mu-coulomb")) p.mu = std::atof(ca.str("mu-coulomb").c_str());
      if (ca.has("mu-viscous")) p.muv = std::atof(ca.str("mu-viscous").c_str());
      if (ca.has("compliance")) p.comp = std::atof(ca.str("compliance").c_str());
      if (ca.has("friction-cone-edges")) { p.nk = std::atoi(ca.str("friction-cone-edges").c_str()); if (p.nk < 4) p.nk = 4; }   // :128-135
      cps.push_back(p);
    } else if (strcmp(nm, "ImplicitConstraint") == 0 || strcmp(nm, "ExplicitConstraint") == 0) return fail("joints are not supported");
  }
  std::sort(dyn.begin(), dyn.end());                                                    // programs/regress.cpp:66-69
  const int nb = (int)dyn.size();
  if (nb < 1 || nb > MH_MAX_BODIES) return fail("%d enabled bodies (supported: 1..%d)", nb, MH_MAX_BODIES);
  sc.nb = nb; sc.has_ground = ground.empty() ? 0 : 1;
  for (int i = 0; i < 3; i++) sc.gravity[i] = g[i];
  std::map<std::string, int> index;
  for (int b = 0; b < nb; b++) {
    const Body& B = bodies[dyn[b]];
    index[B.id] = b;
    snprintf(out->body_id[b], MH_IO_ID_LEN, "%s", B.id.c_str());
    sc.mass[b] = B.mass; for (int i = 0; i < 3; i++) sc.inertia[b][i] = B.J[i];
    if (!(B.mass > 0.0) || !(B.J[0] > 0.0) || !(B.J[1] > 0.0) || !(B.J[2] > 0.0)) return fail("RigidBody %s: mass / inertia missing", B.id.c_str());
    if (B.has_cg && !B.geom.empty()) {
      auto it = prims.find(B.geom);
      if (it == prims.end()) return fail("RigidBody %s: unknown primitive %s", B.id.c_str(), B.geom.c_str());
      const Prim& P = it->second;
      if (P.type == 100) return fail("RigidBody %s: an enabled body with plane geometry is not supported", B.id.c_str());
      if (P.posed) return fail("RigidBody %s: primitive %s has a pose of its own (not supported)", B.id.c_str(), B.geom.c_str());
      sc.geom_type[b] = P.type; for (int i = 0; i < 3; i++) sc.geom_dim[b][i] = P.dim[i];
    } else if (wheel_plugin) {
      sc.geom_type[b] = MH_GEOM_SPOKES; sc.geom_dim[b][0] = 1.0; sc.geom_dim[b][1] = 6.0; sc.geom_dim[b][2] = 0.0;   // params.h:4-6
    } else return fail("RigidBody %s: no collision geometry", B.id.c_str());
    double* s = out->state + MH_BODY_STATE * b;
    for (int i = 0; i < 3; i++) { s[i] = B.x[i]; s[7 + i] = B.v[i]; s[10 + i] = B.w[i]; }
    for (int i = 0; i < 4; i++) s[3 + i] = B.q[i];
  }
  if (sc.has_ground) {
    const Body& G = bodies[ground];
    index[G.id] = nb;
    snprintf(out->body_id[nb], MH_IO_ID_LEN, "%s", G.id.c_str());
    auto it = prims.find(G.geom);
    if (it == prims.end() || it->second.type != 100) return fail("disabled body %s must carry a Plane", G.id.c_str());
    if (G.rotated && it->second.posed) return fail("ground %s: pose on both the body and the plane primitive is not supported", G.id.c_str());
    const double* R = G.rotated ? G.R : it->second.R;
    for (int i = 0; i < 9; i++) sc.plane_R[i] = R[i];
    for (int i = 0; i < 3; i++) sc.plane_o[i] = G.x[i] + it->second.o[i];
  }
  const int ntot = nb + sc.has_ground;
  auto pidx = [&](int i, int j) { if (i > j) std::swap(i, j); return i * ntot - (i * (i + 1)) / 2 + (j - i - 1); };
  for (const CP& p : cps) {
    if (!index.count(p.a) || !index.count(p.b)) continue;     // parameters for bodies outside the simulator (wheel.xml:40 has one commented out)
    const int k = pidx(index[p.a], index[p.b]);
    sc.cp_epsilon[k] = p.eps; sc.cp_mu_coulomb[k] = p.mu; sc.cp_mu_viscous[k] = p.muv; sc.cp_compliance[k] = p.comp; sc.cp_nk[k] = p.nk;
  }
  for (const auto& d : disabled) {
    if (!index.count(d.first) || !index.count(d.second) || d.first == d.second) continue;
    sc.pair_enabled[pidx(index[d.first], index[d.second])] = 0;
  }
  if (xmlNode* drv = first(root, "DRIVER")) { const Attrs da = attrs_of(drv); if (da.has("step-size")) out->step_size = std::atof(da.str("step-size").c_str()); }
  return 0;
}

int mh_io_load_xml_artic(const char* path, mh_io_artic* out, double* q0, double* qd0, double* step_size)
{
  if (!path || !out) return fail("null argument");
  xmlDoc* doc = xmlReadFile(path, nullptr, XML_PARSE_NONET | XML_PARSE_NOERROR | XML_PARSE_NOWARNING);
  if (!doc) return fail("cannot parse %s", path);
  struct Guard { xmlDoc* d; ~Guard() { xmlFreeDoc(d); } } guard{doc};
  xmlNode* root = xmlDocGetRootElement(doc);
  std::vector<xmlNode*> abs; collect(root, "RCArticulatedBody", abs);
  if (abs.size() != 1) return fail("%s: expected exactly one <RCArticulatedBody>, found %zu", path, abs.size());
  xmlNode* ab = abs[0];
  const Attrs aa = attrs_of(ab);
  // floating-base="true" (RCArticulatedBody.cpp:172-175): the base link rides on SIX VIRTUAL 1-DOF JOINTS -- three prismatic ones along the global axes, then three
  // revolute ones about the base link's own x, y, z -- carried by massless links (mh_artic_model joints 0..5; the base link is link 5, the file's joints follow).  At
  // q = 0 they are at the pose the file states, and their rates are the base's linear velocity (global axes, at its COM) and its angular velocity in its own axes.  The
  // batch then needs nothing new: CRBA / the articulated-body recursion, calc_jacobian and the contact rows see six more columns.  What this is NOT: Ravelin's base
  // coordinates (a spatial velocity and a unit quaternion, Ravelin absent from the tree) -- the rotation is integrated in three angles, so a trajectory agrees with a
  // quaternion integrator's to O(dt) in the orientation and the middle angle must stay away from +-90 degrees of the START orientation.
  const bool floating = aa.has("floating-base") && boolean(aa.str("floating-base"));
  if (aa.has("rpy")) return fail("RCArticulatedBody %s: rpy is not supported", aa.str("id").c_str());
  double shift[3] = { 0.0, 0.0, 0.0 };                                       // translate="x,y,z" moves the whole body (RCArticulatedBody.cpp:176-199); taken for floating bases only
  if (aa.has("translate")) {
    if (!floating) return fail("RCArticulatedBody %s: translate is not supported on a fixed base", aa.str("id").c_str());
    const std::vector<double> t = numbers(aa.str("translate")); if (t.size() != 3) return fail("RCArticulatedBody %s: bad translate", aa.str("id").c_str());
    for (int i = 0; i < 3; i++) shift[i] = t[i];
  }
  // ArticulatedBody::load_from_xml (ArticulatedBody.cpp:250-273): with urdf-filename the links and joints come from the URDF file (found relative
  // to the XML file, XMLReader changes into its directory) and nothing else under the element is read
  const bool from_urdf = aa.has("urdf-filename");
  if (from_urdf && floating) return fail("RCArticulatedBody %s: a floating base with urdf-filename is not supported", aa.str("id").c_str());
  UrdfRobot rob;
  if (from_urdf) {
    std::string up = aa.str("urdf-filename");
    if (!up.empty() && up[0] != '/') { const std::string xp = path; const size_t sl = xp.find_last_of('/'); if (sl != std::string::npos) up = xp.substr(0, sl + 1) + up; }
    if (parse_urdf(up.c_str(), rob)) return 1;
  }
  // ---- primitives: mass properties (SpherePrimitive.cpp:138-155, BoxPrimitive.cpp:692-712, CylinderPrimitive.cpp:524-547) ----
  std::map<std::string, Prim> prims;
  { std::vector<xmlNode*> v; collect(root, "Sphere", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = MH_GEOM_SPHERE; const double r = std::atof(a.str("radius").c_str());
      p.dim[0] = r; p.dim[1] = p.dim[2] = 0.0; if (prim_common(a, p, M_PI * r * r * r * 4.0 / 3.0, "Sphere")) return 1;
      p.J[0] = p.J[1] = p.J[2] = r * r * p.mass * 2.0 / 5.0; prims[a.str("id")] = p; } }
  { std::vector<xmlNode*> v; collect(root, "Box", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = MH_GEOM_BOX;
      const double x = std::atof(a.str("xlen").c_str()), y = std::atof(a.str("ylen").c_str()), z = std::atof(a.str("zlen").c_str());
      p.dim[0] = x; p.dim[1] = y; p.dim[2] = z; if (prim_common(a, p, x * y * z, "Box")) return 1;
      const double M = p.mass / 12.0; p.J[0] = M * (y * y + z * z); p.J[1] = M * (x * x + z * z); p.J[2] = M * (x * x + y * y); prims[a.str("id")] = p; } }
  { std::vector<xmlNode*> v; collect(root, "Cylinder", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = 101;
      const double r = std::atof(a.str("radius").c_str()), h = std::atof(a.str("height").c_str());
      p.dim[0] = r; p.dim[1] = h; p.dim[2] = 0.0; if (prim_common(a, p, M_PI * r * r * h, "Cylinder")) return 1;
      const double nl = (1.0 / 12.0) * p.mass * (h * h + 3.0 * (r * r)); p.J[0] = nl; p.J[1] = 0.5 * p.mass * (r * r); p.J[2] = nl; prims[a.str("id")] = p; } }
  { std::vector<xmlNode*> v; collect(root, "Plane", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); Prim p; p.type = 100; p.dim[0] = p.dim[1] = p.dim[2] = 0.0; p.J[0] = p.J[1] = p.J[2] = 0.0;
      if (prim_common(a, p, 0.0, "Plane")) return 1;
      prims[a.str("id")] = p; } }
  // ---- links and joints of the body ----
  struct XWeldGeom { std::string id, prim; double x[3]; double R[9]; };      // the collision geometry of a link a FixedJoint welded onto this one: its link's pose at q = 0
  struct XLink { std::string id; double x[3]; double R[9]; double mass; double J[3]; std::string geom; bool has_cg; double v[3], w[3];
                 double cx[3];                                               // the COM: x, unless links were welded on (x stays where the link's own frame and geometry are)
                 bool composite; double Jf[9];                               // links were welded on: cx is the common COM, Jf the full tensor about it in this link's axes
                 std::vector<XWeldGeom> wg; };
  struct XFixed { std::string id, in, out; };
  std::vector<XFixed> fixeds;
  struct XJoint { std::string id, in, out; int type; double loc[3], axis[3], lo, hi, q, qd, resti; };
  std::vector<XLink> links; std::vector<XJoint> joints;
  for (xmlNode* c = from_urdf ? nullptr : ab->children; c; c = c->next) {
    if (c->type != XML_ELEMENT_NODE) continue;
    const Attrs a = attrs_of(c);
    const std::string nm = (const char*)c->name;
    if (nm == "RigidBody") {
      XLink L; L.id = a.str("id"); L.mass = 0.0; L.J[0] = L.J[1] = L.J[2] = 0.0; L.has_cg = false;
      for (int i = 0; i < 3; i++) L.x[i] = L.v[i] = L.w[i] = 0.0;
      for (int i = 0; i < 9; i++) L.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
      if (a.has("position")) { const std::vector<double> p = numbers(a.str("position")); if (p.size() != 3) return fail("link %s: bad position", L.id.c_str()); for (int i = 0; i < 3; i++) L.x[i] = p[i]; }
      for (int i = 0; i < 3; i++) { L.x[i] += shift[i]; L.cx[i] = L.x[i]; }
      L.composite = false;
      if (a.has("linear-velocity")) { const std::vector<double> p = numbers(a.str("linear-velocity")); if (p.size() != 3) return fail("link %s: bad linear-velocity", L.id.c_str()); for (int i = 0; i < 3; i++) L.v[i] = p[i]; }
      if (a.has("angular-velocity")) { const std::vector<double> p = numbers(a.str("angular-velocity")); if (p.size() != 3) return fail("link %s: bad angular-velocity", L.id.c_str()); for (int i = 0; i < 3; i++) L.w[i] = p[i]; }
      if (a.has("rpy")) { const std::vector<double> r = numbers(a.str("rpy")); if (r.size() != 3) return fail("link %s: bad rpy", L.id.c_str()); rpy_to_R(r[0], r[1], r[2], L.R); }
      if (a.has("quat")) { const std::vector<double> q = numbers(a.str("quat")); if (q.size() != 4) return fail("link %s: bad quat", L.id.c_str());
        const double nr = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]); if (!(nr > 0.0)) return fail("link %s: zero quat", L.id.c_str());
        const double qq[4] = { q[1] / nr, q[2] / nr, q[3] / nr, q[0] / nr }; quat_to_R(qq, L.R); }     // the attribute is w x y z (XMLTree.cpp:407-419)
      if (a.has("mass")) L.mass = std::atof(a.str("mass").c_str());
      if (a.has("inertia")) { const std::vector<double> J = numbers(a.str("inertia")); if (J.size() != 9) return fail("link %s: inertia needs 9 numbers", L.id.c_str());
        if (J[1] != 0 || J[2] != 0 || J[3] != 0 || J[5] != 0 || J[6] != 0 || J[7] != 0) return fail("link %s: only diagonal link-frame inertias are supported", L.id.c_str());
        L.J[0] = J[0]; L.J[1] = J[4]; L.J[2] = J[8]; }
      for (xmlNode* k = c->children; k; k = k->next) {
        if (k->type != XML_ELEMENT_NODE) continue;
        const Attrs ka = attrs_of(k);
        if (strcmp((const char*)k->name, "InertiaFromPrimitive") == 0) {
          const std::string pid = ka.str("primitive-id");
          if (!prims.count(pid)) return fail("link %s: InertiaFromPrimitive names the unknown primitive '%s'", L.id.c_str(), pid.c_str());
          if (prims[pid].posed) return fail("link %s: a posed primitive as inertia source is not supported", L.id.c_str());
          L.mass = prims[pid].mass; for (int i = 0; i < 3; i++) L.J[i] = prims[pid].J[i];
        } else if (strcmp((const char*)k->name, "CollisionGeometry") == 0) { L.geom = ka.str("primitive-id"); L.has_cg = true; }
      }
      links.push_back(L);
    } else if (nm == "RevoluteJoint" || nm == "PrismaticJoint") {
      XJoint J; J.id = a.str("id"); J.type = (nm == "RevoluteJoint") ? MH_JOINT_REVOLUTE : MH_JOINT_PRISMATIC;
      J.in = a.str("inboard-link-id"); J.out = a.str("outboard-link-id");
      const std::vector<double> loc = numbers(a.str("location")), ax = numbers(a.str("axis"));
      if (loc.size() != 3) return fail("joint %s: needs a global location", J.id.c_str());
      if (ax.size() != 3) return fail("joint %s: needs a global axis", J.id.c_str());
      for (int i = 0; i < 3; i++) { J.loc[i] = loc[i] + shift[i]; J.axis[i] = ax[i]; }
      J.lo = -1.7976931348623157e308; J.hi = 1.7976931348623157e308; J.q = 0.0; J.qd = 0.0; J.resti = 0.0;       // Joint.cpp:33-54
      auto one = [&](const char* key, double& dst) -> int { if (!a.has(key)) return 0; const std::vector<double> v = numbers(a.str(key)); if (v.size() != 1) return 1; dst = v[0]; return 0; };
      if (one("lower-limits", J.lo) || one("upper-limits", J.hi) || one("q", J.q) || one("qd", J.qd)) return fail("joint %s: a 1-DOF joint takes one number per attribute", J.id.c_str());
      if (a.has("q-tare")) { double t = 0.0; if (one("q-tare", t) || t != 0.0) return fail("joint %s: q-tare is not supported", J.id.c_str()); }
      if (a.has("restitution-coeff")) J.resti = std::atof(a.str("restitution-coeff").c_str());
      if ((a.has("coulomb-friction-coeff") && std::atof(a.str("coulomb-friction-coeff").c_str()) != 0.0) ||
          (a.has("viscous-friction-coeff") && std::atof(a.str("viscous-friction-coeff").c_str()) != 0.0)) return fail("joint %s: joint friction is not supported", J.id.c_str());
      joints.push_back(J);
    } else if (nm == "FixedJoint") {                                       // zero DOF: the outboard link rides on the inboard one (welded below)
      XFixed F; F.id = a.str("id"); F.in = a.str("inboard-link-id"); F.out = a.str("outboard-link-id");
      fixeds.push_back(F);
    } else if (nm.size() > 5 && nm.compare(nm.size() - 5, 5, "Joint") == 0) return fail("%s %s: only revolute, prismatic and fixed joints are supported", nm.c_str(), a.str("id").c_str());
  }
  // ---- FixedJoints: the batch knows 1-DOF joints only, and in reduced coordinates a zero-DOF joint makes ONE rigid body of its two links (as mh_io_load_urdf does for a URDF
  // "fixed" joint): masses add up, the COM and the tensor about it follow the parallel-axis theorem, the welded link's collision geometry and the joints hanging from it
  // ride on the carrier.  Everything in the file is global at q = 0, so the merge is done in global coordinates.
  std::map<std::string, std::string> welded_to;                             // welded link -> the link that ends up carrying it
  if (!fixeds.empty()) {
    std::map<std::string, std::string> onto;
    std::map<std::string, int> li; for (size_t i = 0; i < links.size(); i++) li[links[i].id] = (int)i;
    for (const XFixed& F : fixeds) {
      if (!li.count(F.in) || !li.count(F.out)) return fail("FixedJoint %s: unknown link", F.id.c_str());
      if (onto.count(F.out)) return fail("link %s is the outboard link of two joints (closed chains are not supported)", F.out.c_str());
      for (const XJoint& J : joints) if (J.out == F.out) return fail("link %s is the outboard link of two joints (closed chains are not supported)", F.out.c_str());
      onto[F.out] = F.in;
    }
    for (const auto& kv : onto) {
      std::string t = kv.second; size_t guard = 0;
      while (onto.count(t)) { t = onto[t]; if (++guard > links.size()) return fail("FixedJoint %s: a loop of fixed joints", kv.first.c_str()); }
      welded_to[kv.first] = t;
    }
    auto global_tensor = [](const XLink& L, double* Jw) {                   // R diag(J) R' (or R Jf R' for a composite)
      double D[9] = { L.J[0], 0, 0, 0, L.J[1], 0, 0, 0, L.J[2] }, T[9];
      mat3mul(L.R, L.composite ? L.Jf : D, T);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jw[3*i+j] = (T[3*i] * L.R[3*j] + T[3*i+1] * L.R[3*j+1]) + T[3*i+2] * L.R[3*j+2];
    };
    for (XLink& T : links) {
      if (welded_to.count(T.id)) continue;
      std::vector<const XLink*> parts;
      for (const XLink& W : links) if (welded_to.count(W.id) && welded_to[W.id] == T.id) parts.push_back(&W);
      if (parts.empty()) continue;
      double M = T.mass, c[3] = { T.mass * T.x[0], T.mass * T.x[1], T.mass * T.x[2] };
      for (const XLink* W : parts) { M += W->mass; for (int k = 0; k < 3; k++) c[k] += W->mass * W->x[k]; }
      if (M > 0.0) for (int k = 0; k < 3; k++) c[k] /= M; else for (int k = 0; k < 3; k++) c[k] = T.x[k];
      double Jw[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
      auto add = [&](const XLink& L) { double Jl[9]; global_tensor(L, Jl); const double d[3] = { L.x[0] - c[0], L.x[1] - c[1], L.x[2] - c[2] }; const double dd = (d[0]*d[0] + d[1]*d[1]) + d[2]*d[2];
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) Jw[3*r+q] += Jl[3*r+q] + L.mass * (((r == q) ? dd : 0.0) - d[r] * d[q]); };
      add(T); for (const XLink* W : parts) add(*W);
      for (const XLink* W : parts) if (W->has_cg) { XWeldGeom g; g.id = W->id; g.prim = W->geom; for (int k = 0; k < 3; k++) g.x[k] = W->x[k]; for (int k = 0; k < 9; k++) g.R[k] = W->R[k]; T.wg.push_back(g); }
      double A[9]; mat3Tmul(T.R, Jw, A); mat3mul(A, T.R, T.Jf);
      T.Jf[1] = T.Jf[3]; T.Jf[2] = T.Jf[6]; T.Jf[5] = T.Jf[7];              // exactly symmetric
      T.composite = true; T.mass = M; for (int k = 0; k < 3; k++) T.cx[k] = c[k];
    }
    for (XJoint& J : joints) { if (welded_to.count(J.in)) J.in = welded_to[J.in]; }
    std::vector<XLink> kept; for (const XLink& L : links) if (!welded_to.count(L.id)) kept.push_back(L);
    links.swap(kept);
  }
  if (!from_urdf && joints.empty() && !floating) return fail("%s: an articulated body without joints", path);
  const int nv = floating ? 6 : 0;                                           // virtual joints in front of the file's
  if ((int)joints.size() + nv > MH_ARTIC_MAX_JOINTS) return fail("%zu joints%s > %d", joints.size(), floating ? " + 6 of the floating base" : "", MH_ARTIC_MAX_JOINTS);
  std::map<std::string, int> link_of; for (size_t i = 0; i < links.size(); i++) link_of[links[i].id] = (int)i;
  for (const auto& kv : welded_to) link_of[kv.first] = link_of[kv.second];
  if (from_urdf) for (size_t i = 0; i < rob.link_names.size(); i++) link_of[rob.link_names[i]] = (int)i;   // (ids only: ContactParameters / DisabledPair may name a link)
  std::map<std::string, int> carried;                                      // outboard link -> joint
  for (size_t j = 0; j < joints.size(); j++) {
    if (!link_of.count(joints[j].in) || !link_of.count(joints[j].out)) return fail("joint %s: unknown link", joints[j].id.c_str());
    if (carried.count(joints[j].out)) return fail("link %s is the outboard link of two joints (closed chains are not supported)", joints[j].out.c_str());
    carried[joints[j].out] = (int)j;
  }
  std::string base; int nbase = 0;
  for (const XLink& L : links) if (!carried.count(L.id)) { base = L.id; nbase++; }
  if (from_urdf) { base = rob.base; nbase = 1; }
  if (nbase != 1) return fail("%s: %d links are carried by no joint (exactly one base link expected)", path, nbase);
  std::vector<int> order; std::map<std::string, int> pos_of;               // parents first, file order within a level
  std::vector<char> used(joints.size(), 0);
  while (order.size() < joints.size()) {
    bool progress = false;
    for (size_t j = 0; j < joints.size(); j++) {
      if (used[j]) continue;
      if (joints[j].in != base && !pos_of.count(joints[j].in)) continue;
      pos_of[joints[j].out] = (int)order.size(); order.push_back((int)j); used[j] = 1; progress = true;
    }
    if (!progress) return fail("%s: joints that do not hang from the base link", path);
  }
  // ---- the simulator: gravity, other bodies, pairs, contact parameters ----
  double grav[3] = { 0.0, 0.0, 0.0 };
  std::map<std::string, std::vector<double> > gravs;
  { std::vector<xmlNode*> v; collect(root, "GravityForce", v);
    for (xmlNode* n : v) { const Attrs a = attrs_of(n); std::vector<double> g = numbers(a.str("accel")); if (g.size() != 3) return fail("GravityForce %s: bad accel", a.str("id").c_str()); gravs[a.str("id")] = g; } }
  xmlNode* sim = first(root, "TimeSteppingSimulator");
  if (!sim) return fail("%s: no <TimeSteppingSimulator>", path);
  std::vector<std::string> dyn; std::vector<std::pair<std::string, std::string> > disabled; std::vector<CP> cps;
  for (xmlNode* c = sim->children; c; c = c->next) {
    if (c->type != XML_ELEMENT_NODE) continue;
    const Attrs a = attrs_of(c); const std::string nm = (const char*)c->name;
    if (nm == "RecurrentForce") { const std::string id = a.str("recurrent-force-id"); if (!gravs.count(id)) return fail("RecurrentForce %s is not a GravityForce", id.c_str()); for (int i = 0; i < 3; i++) grav[i] += gravs[id][i]; }
    else if (nm == "DynamicBody") dyn.push_back(a.str("dynamic-body-id"));
    else if (nm == "DisabledPair") disabled.push_back(std::make_pair(a.str("object1-id"), a.str("object2-id")));
    else if (nm == "ContactParameters") { CP p; p.a = a.str("object1-id"); p.b = a.str("object2-id");
      if (a.has("epsilon")) p.eps = std::atof(a.str("epsilon").c_str());
      if (a.has("mu-coulomb")) p.mu = std::atof(a.str("mu-coulomb").c_str());
      if (a.has("mu-viscous")) p.muv = std::atof(a.str("mu-viscous").c_str());
      if (a.has("compliance")) p.comp = std::atof(a.str("compliance").c_str());
      if (a.has("friction-cone-edges")) p.nk = std::atoi(a.str("friction-cone-edges").c_str());
      cps.push_back(p); }
  }
  const std::string abid = aa.str("id");
  // other bodies of the simulator: at most one, disabled, with a Plane
  std::string plane_body; Prim plane_prim; double plane_x[3] = { 0, 0, 0 }; double plane_Rb[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  std::vector<xmlNode*> rbs; collect(root, "RigidBody", rbs);
  for (const std::string& id : dyn) {
    if (id == abid) continue;
    xmlNode* bn = nullptr;
    for (xmlNode* n : rbs) if (n->parent != ab && attrs_of(n).str("id") == id) bn = n;
    if (!bn) return fail("DynamicBody %s: only one articulated body and one static plane body are supported", id.c_str());
    const Attrs a = attrs_of(bn);
    if (!(a.has("enabled") && !boolean(a.str("enabled")))) return fail("RigidBody %s: a second moving body is not supported next to an articulated one", id.c_str());
    std::string geom; for (xmlNode* k = bn->children; k; k = k->next) if (k->type == XML_ELEMENT_NODE && strcmp((const char*)k->name, "CollisionGeometry") == 0) geom = attrs_of(k).str("primitive-id");
    if (geom.empty()) continue;                                              // no geometry: cannot collide
    if (!prims.count(geom) || prims[geom].type != 100) return fail("RigidBody %s: the static body next to an articulated one must carry a Plane", id.c_str());
    if (!plane_body.empty()) return fail("%s: two static planes", path);
    plane_body = id; plane_prim = prims[geom];
    if (a.has("position")) { const std::vector<double> p = numbers(a.str("position")); if (p.size() != 3) return fail("RigidBody %s: bad position", id.c_str()); for (int i = 0; i < 3; i++) plane_x[i] = p[i]; }
    if (a.has("rpy")) { const std::vector<double> r = numbers(a.str("rpy")); if (r.size() != 3) return fail("RigidBody %s: bad rpy", id.c_str()); rpy_to_R(r[0], r[1], r[2], plane_Rb); }
    if (a.has("quat")) return fail("RigidBody %s: quat on the plane body is not supported (use rpy)", id.c_str());
  }
  if (std::find(dyn.begin(), dyn.end(), abid) == dyn.end()) return fail("%s: the articulated body is not a DynamicBody of the simulator", path);
  // ---- the model ----
  std::memset(out, 0, sizeof(*out));
  mh_artic_model& m = out->model;
  m.nj = (int)order.size() + nv;
  for (int k = 0; k < 3; k++) m.gravity[k] = grav[k];
  { std::string alg = aa.has("fdyn-algorithm") ? aa.str("fdyn-algorithm") : "crb"; for (char& ch : alg) ch = (char)tolower(ch);
    alg.erase(0, alg.find_first_not_of(" \t\n\r")); alg.erase(alg.find_last_not_of(" \t\n\r") + 1);
    if (alg == "fsab") m.algorithm = MH_ARTIC_FSAB; else if (alg == "crb") m.algorithm = MH_ARTIC_CRB; else return fail("fdyn-algorithm '%s': crb or fsab", alg.c_str()); }
  XLink urdf_base; urdf_base.id = base; urdf_base.mass = 0.0; urdf_base.has_cg = false; urdf_base.composite = false; urdf_base.cx[0] = urdf_base.cx[1] = urdf_base.cx[2] = 0.0;      // a URDF robot's base link frame is the model frame
  for (int k = 0; k < 3; k++) { urdf_base.x[k] = 0.0; urdf_base.J[k] = 0.0; }
  for (int k = 0; k < 9; k++) urdf_base.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
  const XLink& B0 = from_urdf ? urdf_base : links[link_of[base]];
  if (from_urdf) {
    const int alg = m.algorithm; double g3[3]; for (int k = 0; k < 3; k++) g3[k] = m.gravity[k];
    *out = rob.art;                                                          // kinematics, inertias, limits, ids; q = qd = 0 (a URDF file holds no state)
    m.algorithm = alg; for (int k = 0; k < 3; k++) m.gravity[k] = g3[k];
    for (int i = 0; i < m.nj; i++) { if (q0) q0[i] = 0.0; if (qd0) qd0[i] = 0.0; }
  }
  if (floating) {
    m.floating_base = 1;
    if (!(B0.mass > 0.0) || !((B0.composite ? B0.Jf[0] : B0.J[0]) > 0.0) || !((B0.composite ? B0.Jf[4] : B0.J[1]) > 0.0) || !((B0.composite ? B0.Jf[8] : B0.J[2]) > 0.0)) return fail("link %s: a floating base link needs mass and inertia", B0.id.c_str());
    static const char* const vname[6] = { "tx", "ty", "tz", "rx", "ry", "rz" };
    for (int v = 0; v < 6; v++) {
      m.parent[v] = v - 1;
      m.jtype[v] = (v < 3) ? MH_JOINT_PRISMATIC : MH_JOINT_REVOLUTE;
      for (int k = 0; k < 9; k++) m.Rrel[v][k] = (v == 3) ? B0.R[k] : ((k % 4 == 0) ? 1.0 : 0.0);
      for (int k = 0; k < 3; k++) { m.trel[v][k] = (v == 0) ? B0.cx[k] : 0.0; m.axis[v][k] = (k == v % 3) ? 1.0 : 0.0; m.com[v][k] = 0.0; }
      for (int k = 0; k < 9; k++) m.inertia[v][k] = 0.0;
      m.mass[v] = 0.0;
      m.lolimit[v] = -1.7976931348623157e308; m.hilimit[v] = 1.7976931348623157e308; m.limit_restitution[v] = 0.0;
      if (q0) q0[v] = 0.0;
      if (v == 5) snprintf(out->link_id[v], MH_IO_ID_LEN, "%s", B0.id.c_str()); else snprintf(out->link_id[v], MH_IO_ID_LEN, "%.40s.virtual-%s", B0.id.c_str(), vname[v]);
      snprintf(out->joint_id[v], MH_IO_ID_LEN, "%.40s.base-%s", abid.c_str(), vname[v]);
    }
    m.mass[5] = B0.mass; m.inertia[5][0] = B0.J[0]; m.inertia[5][4] = B0.J[1]; m.inertia[5][8] = B0.J[2];
    if (B0.composite) for (int k = 0; k < 9; k++) m.inertia[5][k] = B0.Jf[k];
    if (qd0) { double wl[3]; mat3Tvec(B0.R, B0.w, wl); for (int k = 0; k < 3; k++) { qd0[k] = B0.v[k]; qd0[3 + k] = wl[k]; } }
  }
  for (int i0 = 0; i0 < (from_urdf ? 0 : m.nj - nv); i0++) {
    const int i = i0 + nv;
    const XJoint& J = joints[order[i0]];
    const XLink& L = links[link_of[J.out]];
    const bool from_base = (J.in == base);
    // the model's link frame i: origin at the joint location, axes of the link; the base link's frame is the model frame
    const double* Rp = from_base ? B0.R : links[link_of[J.in]].R;
    double xp[3];                                                            // origin of the parent frame: its own joint's location (base: the base link's position)
    if (from_base) for (int k = 0; k < 3; k++) xp[k] = floating ? B0.cx[k] : B0.x[k]; else for (int k = 0; k < 3; k++) xp[k] = joints[order[pos_of[J.in]]].loc[k];
    m.parent[i] = from_base ? nv - 1 : pos_of[J.in] + nv;                     // (a floating base link is link 5)
    m.jtype[i] = J.type;
    mat3Tmul(Rp, L.R, m.Rrel[i]);
    const double d[3] = { J.loc[0] - xp[0], J.loc[1] - xp[1], J.loc[2] - xp[2] };
    mat3Tvec(Rp, d, m.trel[i]);
    double al[3]; mat3Tvec(L.R, J.axis, al);
    const double nrm = std::sqrt(al[0]*al[0] + al[1]*al[1] + al[2]*al[2]);
    if (!(nrm > 0.0)) return fail("joint %s: zero axis", J.id.c_str());
    for (int k = 0; k < 3; k++) m.axis[i][k] = al[k] / nrm;
    const double dc[3] = { L.cx[0] - J.loc[0], L.cx[1] - J.loc[1], L.cx[2] - J.loc[2] };   // the RigidBody position is its COM (a composite's: the common one)
    mat3Tvec(L.R, dc, m.com[i]);
    if (!(L.mass > 0.0)) return fail("link %s: no mass (InertiaFromPrimitive or mass / inertia)", L.id.c_str());
    m.mass[i] = L.mass;
    for (int k = 0; k < 9; k++) m.inertia[i][k] = 0.0;
    m.inertia[i][0] = L.J[0]; m.inertia[i][4] = L.J[1]; m.inertia[i][8] = L.J[2];
    if (L.composite) for (int k = 0; k < 9; k++) m.inertia[i][k] = L.Jf[k];
    m.lolimit[i] = J.lo; m.hilimit[i] = J.hi; m.limit_restitution[i] = J.resti;
    if (q0) q0[i] = J.q;
    if (qd0) qd0[i] = J.qd;
    snprintf(out->link_id[i], MH_IO_ID_LEN, "%s", L.id.c_str());
    snprintf(out->joint_id[i], MH_IO_ID_LEN, "%s", J.id.c_str());
  }
  // the base link's frame must be the model frame for the kinematics above: move everything into it
  {
    bool ident = true; for (int k = 0; k < 9; k++) if (B0.R[k] != ((k % 4 == 0) ? 1.0 : 0.0)) ident = false;
    if (!floating && (!ident || B0.x[0] != 0.0 || B0.x[1] != 0.0 || B0.x[2] != 0.0)) {        // (a floating body's model frame is the global frame)
      // joints hanging from the base: Rrel / trel were taken relative to the base pose; the model frame is then the base frame,
      // so gravity and the plane have to be expressed in it as well
      double g2[3]; mat3Tvec(B0.R, grav, g2); for (int k = 0; k < 3; k++) m.gravity[k] = g2[k];
    }
  }
  // ---- collision geometry ----
  auto pair_disabled = [&](const std::string& x, const std::string& y) {
    for (const auto& d : disabled) if ((d.first == x && d.second == y) || (d.first == y && d.second == x)) return true;
    return false;
  };
  // collision geometries of the body: model link (-1 = base), id for <DisabledPair>, a sphere's centre in the model link's frame
  std::vector<UrdfGeom> geoms;
  if (from_urdf) geoms = rob.geoms;
  else {
    if (B0.has_cg && !floating) { UrdfGeom g; g.link = -1; g.id = base; g.sphere = false; g.radius = 0.0; g.center[0] = g.center[1] = g.center[2] = 0.0; geoms.push_back(g); }
    // geometry that a FixedJoint welded onto model link `link` (frame: origin `org`, axes RT): the welded link's pose at q = 0 seen from there
    auto add_welded = [&](const XLink& T, int link, const double* org) {
      for (const XWeldGeom& w : T.wg) {
        UrdfGeom g; g.link = link; g.id = w.id; g.sphere = prims.count(w.prim) && prims[w.prim].type == MH_GEOM_SPHERE; g.radius = g.sphere ? prims[w.prim].dim[0] : 0.0;
        const double d[3] = { w.x[0] - org[0], w.x[1] - org[1], w.x[2] - org[2] };
        double a[3], Rr[9], b[3] = { 0.0, 0.0, 0.0 };
        mat3Tvec(T.R, d, a); mat3Tmul(T.R, w.R, Rr); if (g.sphere) mat3vec(Rr, prims[w.prim].o, b);
        for (int k = 0; k < 3; k++) g.center[k] = a[k] + b[k];
        geoms.push_back(g);
      }
    };
    if (B0.has_cg && floating) {                                             // the floating base link moves: link 5, its frame's origin is its COM
      UrdfGeom g; g.link = 5; g.id = base; g.sphere = prims.count(B0.geom) && prims[B0.geom].type == MH_GEOM_SPHERE; g.radius = g.sphere ? prims[B0.geom].dim[0] : 0.0;
      for (int k = 0; k < 3; k++) g.center[k] = g.sphere ? prims[B0.geom].o[k] : 0.0;
      if (B0.composite) { const double d[3] = { B0.x[0] - B0.cx[0], B0.x[1] - B0.cx[1], B0.x[2] - B0.cx[2] }; double a[3]; mat3Tvec(B0.R, d, a); for (int k = 0; k < 3; k++) g.center[k] += a[k]; }
      geoms.push_back(g);
    }
    if (floating) add_welded(B0, 5, B0.cx);
    else for (const XWeldGeom& w : B0.wg) { UrdfGeom g; g.link = -1; g.id = w.id; g.sphere = false; g.radius = 0.0; g.center[0] = g.center[1] = g.center[2] = 0.0; geoms.push_back(g); }   // rides on the fixed base: static
    for (int i = nv; i < m.nj; i++) {
      const XLink& L = links[link_of[joints[order[i - nv]].out]];
      add_welded(L, i, joints[order[i - nv]].loc);
      if (!L.has_cg) continue;
      UrdfGeom g; g.link = i; g.id = L.id; g.sphere = prims.count(L.geom) && prims[L.geom].type == MH_GEOM_SPHERE; g.radius = g.sphere ? prims[L.geom].dim[0] : 0.0;
      // centre in the model link frame (origin at the joint): COM offset + the primitive's own offset in the link's axes
      for (int k = 0; k < 3; k++) g.center[k] = m.com[i][k] + (g.sphere ? prims[L.geom].o[k] : 0.0);
      if (L.composite) {                                                     // (the COM has moved away from the link's own position, where its geometry is)
        const double* jl = joints[order[i - nv]].loc; const double d[3] = { L.x[0] - jl[0], L.x[1] - jl[1], L.x[2] - jl[2] }; double a[3]; mat3Tvec(L.R, d, a);
        for (int k = 0; k < 3; k++) g.center[k] = a[k] + (g.sphere ? prims[L.geom].o[k] : 0.0);
      }
      geoms.push_back(g);
    }
  }
  // every pair of the body's own geometries must be disabled (the whole body, or link by link) -- geometries riding on one link cannot meet
  if (!pair_disabled(abid, abid))
    for (size_t x = 0; x < geoms.size(); x++) for (size_t y = x + 1; y < geoms.size(); y++)
      if (geoms[x].link != geoms[y].link && !pair_disabled(geoms[x].id, geoms[y].id))
        return fail("links %s and %s can collide: link-link contact is not supported (add a <DisabledPair>)", geoms[x].id.c_str(), geoms[y].id.c_str());
  if (!plane_body.empty()) {
    for (const UrdfGeom& g : geoms) {
      if (g.link < 0) continue;                                              // rides on the fixed base: static against the static plane
      if (pair_disabled(g.id, plane_body) || pair_disabled(abid, plane_body)) continue;
      if (!g.sphere) return fail("link %s: only Sphere collision geometry can meet the plane (others: disable the pair)", g.id.c_str());
      if (m.nspheres >= MH_ARTIC_MAX_SPHERES) return fail("more than %d link spheres", MH_ARTIC_MAX_SPHERES);
      const int s = m.nspheres++;
      m.sphere_link[s] = g.link; m.sphere_radius[s] = g.radius;
      for (int k = 0; k < 3; k++) m.sphere_center[s][k] = g.center[k];
    }
    // the plane: PlanePrimitive's +Y is the normal; body pose times primitive pose, expressed in the model (base) frame
    double Rw[9]; mat3mul(plane_Rb, plane_prim.R, Rw);
    double ow[3]; mat3vec(plane_Rb, plane_prim.o, ow); for (int k = 0; k < 3; k++) ow[k] += plane_x[k];
    if (floating) { for (int k = 0; k < 9; k++) m.plane_R[k] = Rw[k]; for (int k = 0; k < 3; k++) m.plane_o[k] = ow[k]; }
    else {
      mat3Tmul(B0.R, Rw, m.plane_R);
      const double dd[3] = { ow[0] - B0.x[0], ow[1] - B0.x[1], ow[2] - B0.x[2] };
      mat3Tvec(B0.R, dd, m.plane_o);
    }
    m.cp_nk = 4;
    for (const CP& p : cps) {
      const std::string other = (p.a == plane_body) ? p.b : ((p.b == plane_body) ? p.a : std::string());
      if (other.empty()) continue;
      if (other == abid || link_of.count(other)) { m.cp_epsilon = p.eps; m.cp_mu_coulomb = p.mu; m.cp_mu_viscous = p.muv; m.cp_compliance = p.comp; m.cp_nk = p.nk; break; }
    }
    m.min_step_size = std::sqrt(2.220446049250313e-16); m.contact_dist_thresh = 1e-6;
    { const Attrs sa = attrs_of(sim); if (sa.has("min-step-size")) m.min_step_size = std::atof(sa.str("min-step-size").c_str()); }
  }
  // constraint stabilisation (ConstraintSimulator.cpp:574-590): joint-limit rows, and contact rows for the link spheres that can meet the plane
  { const Attrs sa = attrs_of(sim);
    m.cstab_eps = std::sqrt(2.220446049250313e-16);
    if (sa.has("unilateral-stabilization-tol")) m.cstab_eps = std::atof(sa.str("unilateral-stabilization-tol").c_str());
    const unsigned long mi = sa.has("constraint-stabilization-max-iterations") ? std::strtoul(sa.str("constraint-stabilization-max-iterations").c_str(), nullptr, 10) : (unsigned long)MH_CSTAB_DEFAULT_MAX_ITERATIONS;
    m.cstab_max_iterations = (int)((mi > 0x7fffffffUL) ? 0x7fffffffUL : mi); }
  if (step_size) { *step_size = 0.0; if (xmlNode* drv = first(root, "DRIVER")) { const Attrs a = attrs_of(drv); if (a.has("step-size")) *step_size = std::atof(a.str("step-size").c_str()); } }
  return 0;
}

int mh_io_format_row(double t, const double* state, int nb, char* buf, int cap)
{
  std::ostringstream o;                                   // ostream default formatting, as regress.cpp:82-93
  o << t;
  for (int b = 0; b < nb; b++) for (int k = 0; k < 7; k++) o << " " << state[MH_BODY_STATE * b + k];
  const std::string s = o.str();
  if (buf && cap > 0) { snprintf(buf, (size_t)cap, "%s", s.c_str()); }
  return (int)s.size();
}

int mh_io_compare_trajs(const char* file1, const char* file2, double tol, double* max_diff, double* timing)
{
  std::ifstream a(file1), b(file2);
  if (a.fail() || b.fail()) { fail("compare-trajs: unable to open one or both files"); return -1; }
  std::vector<std::string> la, lb; std::string s;
  while (std::getline(a, s)) la.push_back(s);
  while (std::getline(b, s)) lb.push_back(s);
  while (!la.empty() && la.back().find_first_not_of(" \t\r") == std::string::npos) la.pop_back();
  while (!lb.empty() && lb.back().find_first_not_of(" \t\r") == std::string::npos) lb.pop_back();
  if (la.size() != lb.size() || la.empty()) { fail("compare-trajs: unequal numbers of lines (%zu vs %zu)", la.size(), lb.size()); return -1; }
  auto parse = [](const std::string& str) {               // blanks / commas; inf / -inf (compare-trajs.cpp:29-84)
    std::vector<double> v; std::string t = str; for (char& c : t) if (c == ',') c = ' ';
    std::istringstream in(t); std::string w;
    while (in >> w) {
      if (strcasecmp(w.c_str(), "inf") == 0) v.push_back(INFINITY);
      else if (strcasecmp(w.c_str(), "-inf") == 0) v.push_back(-INFINITY);
      else v.push_back(std::atof(w.c_str()));
    }
    return v; };
  double md = 0.0;
  for (size_t i = 0; i + 1 < la.size(); i++) {
    const std::vector<double> va = parse(la[i]), vb = parse(lb[i]);
    if (va.size() != vb.size()) { fail("compare-trajs: row %zu has %zu vs %zu values", i, va.size(), vb.size()); return -1; }
    for (size_t k = 0; k < va.size(); k++) md = std::max(md, std::fabs(va[k] - vb[k]));
  }
  const std::vector<double> ta = parse(la.back()), tb = parse(lb.back());
  if (ta.size() != 1 || tb.size() != 1) { fail("compare-trajs: last line must hold the timing"); return -1; }
  if (timing) { timing[0] = ta[0]; timing[1] = tb[0]; }
  if (max_diff) *max_diff = md;
  return (md > tol) ? 1 : 0;
}

} // extern "C"
