// HipMeshAdapter.cpp -- see HipMeshAdapter.h.  Marshals GraviT's types into the C ABI; no arithmetic here.
#include "HipMeshAdapter.h"

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

using namespace gvt::render::adapter::hip::data;
using gvt::render::actor::Ray;
using gvt::render::data::primitives::Material;
using gvt::render::data::primitives::Mesh;

static_assert(sizeof(Ray) == sizeof(gvt_hip_ray), "gvt_hip_ray must be the 80-byte gvt::render::actor::Ray image");
static_assert(sizeof(Material) == sizeof(gvt_hip_material), "gvt_hip_material must be the 92-byte Material POD");

static void fail(const char *what) {
  // the reference's adapters exit(1) on engine errors (EmbreeMeshAdapter.cpp:90-123); an exception lets the caller decide
  throw std::runtime_error(std::string("HipMeshAdapter: ") + what + ": " + gvt_hip_last_error());
}

HipMeshAdapter::HipMeshAdapter(std::shared_ptr<gvt::render::data::primitives::Data> m, int normal_mode)
    : Adapter(m), mesh_(nullptr), normal_mode_(normal_mode), trace_calls_(0) {
  std::shared_ptr<Mesh> mesh = std::dynamic_pointer_cast<Mesh>(m);
  if (!mesh) throw std::runtime_error("HipMeshAdapter: mesh pointer in the database is null"); // GVT_ASSERT, EmbreeMeshAdapter.cpp:128
  if (gvt_hip_abi_version() != GVT_HIP_ABI_VERSION) // the library on the loader's path is another revision than the header this file was compiled against
    throw std::runtime_error("HipMeshAdapter: libgvt_hip.so is ABI revision " + std::to_string(gvt_hip_abi_version()) + ", built against " + std::to_string(GVT_HIP_ABI_VERSION));
  mesh->generateNormals();                                                                       // :129 (mutates the shared Mesh)

  const size_t nV = mesh->vertices.size(), nT = mesh->faces.size();
  std::vector<int32_t> tris(nT * 3);
  for (size_t i = 0; i < nT; i++) { // std::tuple stores its elements in reverse: always go through std::get<> (:152-155)
    const Mesh::Face &f = mesh->faces[i];
    tris[3 * i] = std::get<0>(f); tris[3 * i + 1] = std::get<1>(f); tris[3 * i + 2] = std::get<2>(f);
  }
  // faces_to_materials (vector<Material*>) -> table + per-face index
  std::vector<gvt_hip_material> table;
  std::vector<int32_t> face_mat;
  if (mesh->faces_to_materials.size() == nT && nT) {
    std::map<const Material *, int32_t> seen;
    face_mat.resize(nT);
    for (size_t i = 0; i < nT; i++) {
      const Material *p = mesh->faces_to_materials[i];
      if (!p) { face_mat[i] = -1; continue; }
      auto it = seen.find(p);
      if (it == seen.end()) {
        gvt_hip_material pod;
        std::memcpy(&pod, p, sizeof pod);
        it = seen.insert(std::make_pair(p, (int32_t)table.size())).first;
        table.push_back(pod);
      }
      face_mat[i] = it->second;
    }
  }
  gvt_hip_material mesh_mat;
  const bool have_mat = mesh->getMaterial() != nullptr;
  if (have_mat) std::memcpy(&mesh_mat, mesh->getMaterial(), sizeof mesh_mat);
  const float *vcol = (mesh->vertex_colors.size() == nV && nV) ? &mesh->vertex_colors[0][0] : nullptr;
  const float *vnrm = (mesh->normals.size() == nV && nV) ? &mesh->normals[0][0] : nullptr;

  mesh_ = gvt_hip_mesh_create(nV ? &mesh->vertices[0][0] : nullptr, nV, tris.data(), nT, vnrm, vcol, table.empty() ? nullptr : table.data(),
                              table.size(), face_mat.empty() ? nullptr : face_mat.data(), have_mat ? &mesh_mat : nullptr);
  if (!mesh_) fail("gvt_hip_mesh_create");
}

HipMeshAdapter::~HipMeshAdapter() { gvt_hip_mesh_destroy(mesh_); }

void HipMeshAdapter::trace(gvt::render::actor::RayVector &rayList, gvt::render::actor::RayVector &moved_rays, glm::mat4 *m, glm::mat4 *minv,
                           glm::mat3 *normi, std::vector<std::shared_ptr<gvt::render::data::scene::Light> > &lights, size_t begin,
                           size_t end) {
  using namespace gvt::render::data::scene;
  if (end == 0) end = rayList.size(); // EmbreeMeshAdapter.cpp:642
  std::vector<gvt_hip_light> pods(lights.size());
  for (size_t i = 0; i < lights.size(); i++) {
    gvt_hip_light &L = pods[i];
    std::memset(&L, 0, sizeof L);
    Light *l = lights[i].get();
    std::memcpy(L.position, &l->position[0], 12);
    if (AreaLight *a = dynamic_cast<AreaLight *>(l)) {
      L.type = GVT_HIP_LIGHT_AREA;
      std::memcpy(L.color, &a->color[0], 12);
      std::memcpy(L.normal, &a->LightNormal[0], 12);
      L.width = a->LightWidth; L.height = a->LightHeight;
    } else if (PointLight *p = dynamic_cast<PointLight *>(l)) {
      L.type = GVT_HIP_LIGHT_POINT;
      std::memcpy(L.color, &p->color[0], 12);
    } else if (AmbientLight *am = dynamic_cast<AmbientLight *>(l)) {
      L.type = GVT_HIP_LIGHT_AMBIENT;
      std::memcpy(L.color, &am->color[0], 12);
    } else {
      L.type = GVT_HIP_LIGHT_POINT; // base Light::contribution returns Color() == black (Light.cpp:48)
    }
  }
  const size_t n = end > begin ? end - begin : 0;
  const size_t cap = n * (1 + pods.size()) + 16;
  size_t n_out = 0;
  // The moved rays are written straight behind moved_rays' current contents, under the adapter's lock like the append of
  // EmbreeMeshAdapter.cpp:619-621 (Ray() constructs nothing, so growing the vector touches no memory; the schedulers reserve
  // 10x the input, ImageTracer.h:240): no intermediate buffer, no second copy of 80 bytes per moved ray.
  // One call at a time per adapter, like the reference (its trace() writes the members begin / end / global_scene,
  // EmbreeMeshAdapter.cpp:633-645, and both schedulers call an adapter from their single scheduler thread): the lock is held across
  // the device call because the library writes into moved_rays' own storage -- a second caller would otherwise see the vector
  // grown by the worst case.  Concurrency inside the call is the library's (gvt_hip_trace_ex pipelines the list over several
  // host threads and streams).  cap is the worst case n * (1 + lights), which is also what lets the library take that path.
  std::unique_lock<std::mutex> moved(_outqueue);
  const size_t old = moved_rays.size();
  moved_rays.resize(old + cap);
  // (Both schedulers hand the adapter a moved_rays vector reserved afresh for every call -- ImageTracer.h:240, DomainTracer.h:308 -- whose
  // pages have never been touched: the device-to-host copies fault them in, ~0.4 us per 4 KiB page, 15 ms for the 166 MB a 1080p call
  // returns.  Populating them from a helper thread while the upload runs (MADV_POPULATE_WRITE) was measured and made the call SLOWER,
  // 19.8 -> 23 ms: the runtime's registration of the upload's host pages waits behind the populate's hold on the address space.)
  int rc = gvt_hip_trace_ex(mesh_, reinterpret_cast<gvt_hip_ray *>(rayList.data()), rayList.size(), begin, end,
                            reinterpret_cast<gvt_hip_ray *>(moved_rays.data() + old), cap, &n_out, &(*m)[0][0], &(*minv)[0][0], &(*normi)[0][0],
                            pods.empty() ? nullptr : pods.data(), pods.size(), normal_mode_, trace_calls_++,
                            write_back_ ? 0u : GVT_HIP_TRACE_NO_WRITEBACK);
  moved_rays.resize(old + (rc == GVT_HIP_OK ? n_out : 0));
  moved.unlock();
  if (rc != GVT_HIP_OK) fail("gvt_hip_trace");
}
