// HipMeshAdapter.h -- gvt::render::adapter::hip::data::HipMeshAdapter
//
// The GraviT-side binding of libgvt_hip.so: a gvt::render::Adapter (src/gvt/render/Adapter.h:44-88) with the
// same shape as gvt::render::adapter::embree::data::EmbreeMeshAdapter (adapter/embree/EmbreeMeshAdapter.h),
// written against GraviT's OWN headers.  Everything below the virtual call is the C ABI of
// include/gvt_hip.h; there is no CPU path.
//
// Build inside a GraviT tree: add this directory to src/gvt/render/adapter/hip/, compile with
// -DGVT_RENDER_ADAPTER_HIP, link libgvt_hip.so.
#ifndef GVT_RENDER_ADAPTER_HIP_DATA_HIP_MESH_ADAPTER_H
#define GVT_RENDER_ADAPTER_HIP_DATA_HIP_MESH_ADAPTER_H

#include <gvt/render/Adapter.h>

#include "gvt_hip.h"

namespace gvt {
namespace render {
namespace adapter {
namespace hip {
namespace data {

class HipMeshAdapter : public gvt::render::Adapter {
public:
  /**
   * Construct the adapter from a Mesh data node (EmbreeMeshAdapter.cpp:125-162): generates the mesh's vertex
   * normals like the reference constructor does, copies the geometry to the device and builds the LBVH there.
   * normal_mode: GVT_HIP_NORMALS_FLAT == the current EmbreeMeshAdapter.cpp (FLAT_SHADING),
   *              GVT_HIP_NORMALS_SMOOTH == EmbreeStream / OptiX adapters and the CTest goldens.
   */
  HipMeshAdapter(std::shared_ptr<gvt::render::data::primitives::Data> mesh, int normal_mode = GVT_HIP_NORMALS_FLAT);
  virtual ~HipMeshAdapter();

  /** Adapter::trace (Adapter.h:82-84): same arguments, same output contract as EmbreeMeshAdapter::trace. */
  virtual void trace(gvt::render::actor::RayVector &rayList, gvt::render::actor::RayVector &moved_rays, glm::mat4 *m,
                     glm::mat4 *minv, glm::mat3 *normi, std::vector<std::shared_ptr<gvt::render::data::scene::Light> > &lights,
                     size_t begin = 0, size_t end = 0);

  gvt_hip_mesh *handle() const { return mesh_; }

  /** false: trace() leaves rayList as it is (GVT_HIP_TRACE_NO_WRITEBACK).  Tracer<ImageScheduler> and Tracer<DomainScheduler> clear
   *  the traced queue right after the call (ImageTracer.h:248, DomainTracer.h:316), so under them the in-place update is never read
   *  and a third of the call's PCIe traffic can go.  Default true: the literal contract of EmbreeMeshAdapter::trace. */
  void setWriteBack(bool on) { write_back_ = on; }

private:
  gvt_hip_mesh *mesh_;
  int normal_mode_;
  unsigned trace_calls_;
  bool write_back_ = true;
};

} // namespace data
} // namespace hip
} // namespace adapter
} // namespace render
} // namespace gvt

#endif
