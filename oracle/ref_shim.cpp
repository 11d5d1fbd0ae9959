// ref_shim.cpp -- thin extern "C" wrapper over the REFERENCE's own code, compiled from the
// sources where they lie under /root/reference (never copied).  TEST INFRASTRUCTURE ONLY.
//
// Built by oracle/Makefile into oracle/_ref/libgvtref.so together with
//   src/gvt/render/actor/Ray.cpp, data/primitives/{Material,Mesh,BBox}.cpp, data/scene/Light.cpp
// (these compile stand-alone against the vendored glm; the Embree / TBB / MPI / IceT dependent
// files do not and are treated as unbuildable -- see DESIGN.md).
//
// Every function just marshals PODs into the reference's types and calls the reference.
#include <gvt/core/math/RandEngine.h>
#include <gvt/render/actor/Ray.h>
#include <gvt/render/actor/RayPacket.h>
#include <gvt/render/data/primitives/BBox.h>
#include <gvt/render/data/primitives/Material.h>
#include <gvt/render/data/primitives/Mesh.h>
#include <gvt/render/data/primitives/Shade.h>
#include <gvt/render/data/scene/Light.h>

#include <cstdint>
#include <cstring>
#include <memory>

using namespace gvt::render;
using gvt::render::actor::Ray;

extern "C" {

int ref_sizeof_ray() { return (int)sizeof(Ray); }
int ref_sizeof_material() { return (int)sizeof(data::primitives::Material); }
int ref_sizeof_box3d() { return (int)sizeof(data::primitives::Box3D); }
float ref_ray_epsilon() { return Ray::RAY_EPSILON; }

// light_type: 0 point, 1 area, 2 ambient.  ray80 = 80-byte Ray image, mat92 = 92-byte Material image.
int ref_shade(const void *mat92, const void *ray80, const float *N, int light_type, const float *lpos, const float *lcolor,
              const float *lnormal, float lwidth, float lheight, const float *lightPosSample, float *color_out) {
  data::primitives::Material mat;
  std::memcpy(&mat, mat92, sizeof mat);
  Ray ray((const unsigned char *)ray80);
  glm::vec3 n(N[0], N[1], N[2]), c(0.f);
  glm::vec3 pos(lpos[0], lpos[1], lpos[2]), col(lcolor[0], lcolor[1], lcolor[2]);
  std::unique_ptr<data::scene::Light> L;
  if (light_type == 0) L.reset(new data::scene::PointLight(pos, col));
  else if (light_type == 1)
    L.reset(new data::scene::AreaLight(pos, col, glm::vec3(lnormal[0], lnormal[1], lnormal[2]), lheight, lwidth));
  else L.reset(new data::scene::AmbientLight(col));
  bool ok = data::primitives::Shade(&mat, ray, n, L.get(), glm::vec3(lightPosSample[0], lightPosSample[1], lightPosSample[2]), c);
  color_out[0] = c.x; color_out[1] = c.y; color_out[2] = c.z;
  return ok ? 1 : 0;
}

void ref_area_light_position(const float *lpos, const float *lcolor, const float *lnormal, float lwidth, float lheight,
                             uint32_t *seed, float *out) {
  data::scene::AreaLight L(glm::vec3(lpos[0], lpos[1], lpos[2]), glm::vec3(lcolor[0], lcolor[1], lcolor[2]),
                           glm::vec3(lnormal[0], lnormal[1], lnormal[2]), lheight, lwidth);
  glm::vec3 p = L.GetPosition(seed);
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

// Mesh::generateNormals through Mesh::addVertex / faces (0-based faces pushed directly, as ObjReader does)
void ref_generate_normals(const float *verts, int nV, const int32_t *tris, int nT, float *normals_out) {
  data::primitives::Mesh mesh(new data::primitives::Material());
  for (int i = 0; i < nV; i++) mesh.addVertex(glm::vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]));
  for (int i = 0; i < nT; i++) mesh.faces.push_back(data::primitives::Mesh::Face(tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]));
  mesh.generateNormals();
  for (int i = 0; i < nV; i++) {
    normals_out[3 * i] = mesh.normals[i].x; normals_out[3 * i + 1] = mesh.normals[i].y; normals_out[3 * i + 2] = mesh.normals[i].z;
  }
}

// Mesh::addFace (1-based, drops faces with coincident vertices): returns the surviving 0-based faces
int ref_add_faces(const float *verts, int nV, const int32_t *tris1, int nT, int32_t *tris_out) {
  data::primitives::Mesh mesh(new data::primitives::Material());
  for (int i = 0; i < nV; i++) mesh.addVertex(glm::vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]));
  for (int i = 0; i < nT; i++) mesh.addFace(tris1[3 * i], tris1[3 * i + 1], tris1[3 * i + 2]);
  int k = 0;
  for (auto &f : mesh.faces) {
    tris_out[3 * k] = std::get<0>(f); tris_out[3 * k + 1] = std::get<1>(f); tris_out[3 * k + 2] = std::get<2>(f);
    k++;
  }
  return k;
}

// RayPacketIntersection<1>::intersect(bb, hit, update=true) for one ray and one box.
// Returns hit (1/-1 -> 1/0) and the packet's t after the call.
int ref_raypacket_intersect(const void *ray80, const float *lo, const float *hi, float *t_inout) {
  actor::RayVector rays;
  rays.push_back(Ray((const unsigned char *)ray80));
  actor::RayPacketIntersection<1> rp(rays.begin(), rays.end());
  rp.t[0] = *t_inout;
  data::primitives::Box3D bb(glm::vec3(lo[0], lo[1], lo[2]), glm::vec3(hi[0], hi[1], hi[2]));
  int hit[1];
  rp.intersect(bb, hit, true);
  *t_inout = rp.t[0];
  return hit[0] == 1;
}

float ref_rng(uint32_t *seed) {
  gvt::core::math::RandEngine e;
  return e.rng(*seed);
}
float ref_fastrand_lcg(uint32_t *seed, float mn, float mx) {
  gvt::core::math::RandEngine e;
  return e.fastrand(seed, mn, mx);
}

void ref_default_material(void *mat92) {
  data::primitives::Material m;
  std::memcpy(mat92, &m, sizeof m);
}

// Ray(origin, direction, w, type, depth) constructor image (Ray.h:106-116)
void ref_ray_ctor(const float *o, const float *d, float w, int type, void *ray80_out) {
  std::memset(ray80_out, 0, 80);
  Ray r(glm::vec3(o[0], o[1], o[2]), glm::vec3(d[0], d[1], d[2]), w, (Ray::RayType)type, 1);
  r.mice.depth = 0;
  r.mice.color = glm::vec3(0.f);
  std::memcpy(ray80_out, r.data, sizeof(Ray) < 80 ? sizeof(Ray) : 80);
}

float ref_box_surface_area(const float *lo, const float *hi) {
  data::primitives::Box3D b(glm::vec3(lo[0], lo[1], lo[2]), glm::vec3(hi[0], hi[1], hi[2]));
  return b.surfaceArea();
}
int ref_box_wide_dir(const float *lo, const float *hi) {
  data::primitives::Box3D b(glm::vec3(lo[0], lo[1], lo[2]), glm::vec3(hi[0], hi[1], hi[2]));
  return b.wideRangingBoxDir();
}
}
