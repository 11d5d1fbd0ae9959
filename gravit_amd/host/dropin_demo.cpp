// dropin_demo.cpp -- drives HipMeshAdapter through a gvt::render::Adapter* with GraviT's OWN types
// (Mesh, Ray, RayVector, PointLight, glm matrices), the way Tracer<ImageScheduler> does
// (algorithm/ImageTracer.h:184-250).  Built only where the GraviT tree is available (oracle/Makefile target
// `dropin`, output oracle/_ref/dropin_demo) and run on the GPU box by tests/test_gpu_dropin.py.
//
//   dropin_demo <mesh.obj> <rays.bin> <normal_mode> <out.bin>
// rays.bin: n 80-byte gvt::render::actor::Ray images (the incoming queue).  Instance transform = identity, one
// PointLight at (0,0.1,0.5) (SimpleFileLoadApp.cpp:223-225).
// writes: uint64 n_moved, then n_moved 80-byte rays (moved_rays), then the n rays of rayList after the call.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <fstream>
#include <sstream>

#include <glm/gtc/matrix_transform.hpp>

#include "HipMeshAdapter.h"

using namespace gvt::render;
using gvt::render::actor::Ray;
using gvt::render::actor::RayVector;

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s mesh.obj rays.bin normal_mode out.bin\n", argv[0]); return 2; }
  const int mode = std::atoi(argv[3]);
  auto mesh = std::make_shared<data::primitives::Mesh>(new data::primitives::Material());
  std::ifstream in(argv[1]);
  std::string line;
  while (std::getline(in, line)) { // ObjReader.cpp:122-135 pushes 0-based faces without the degenerate filter
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "v") { float x, y, z; ss >> x >> y >> z; mesh->addVertex(glm::vec3(x, y, z)); }
    else if (tag == "f") { int a, b, c; ss >> a >> b >> c; mesh->faces.push_back(data::primitives::Mesh::Face(a - 1, b - 1, c - 1)); }
  }
  std::shared_ptr<Adapter> adapter = std::make_shared<adapter::hip::data::HipMeshAdapter>(mesh, mode);

  RayVector rays;
  {
    std::ifstream rf(argv[2], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)rf.tellg();
    rf.seekg(0);
    rays.resize(bytes / sizeof(Ray));
    rf.read((char *)rays.data(), rays.size() * sizeof(Ray));
  }
  glm::mat4 m(1.f), minv = glm::inverse(m);
  glm::mat3 normi = glm::transpose(glm::inverse(glm::mat3(m)));
  std::vector<std::shared_ptr<data::scene::Light> > lights;
  lights.push_back(std::make_shared<data::scene::PointLight>(glm::vec3(0.0, 0.1, 0.5), glm::vec3(1.0, 1.0, 1.0)));

  // optional 5th argument: timed repetitions of the same call on copies of the list (the drop-in path's rate: host RayVector in,
  // host RayVector out, through the virtual interface)
  const int reps = argc > 5 ? std::atoi(argv[5]) : 0;
  double best_ms = -1.0, best_reused_ms = -1.0;
  for (int r = 0; r < reps; r++) { // (a) the schedulers' own pattern: moved_rays reserved afresh for every call (ImageTracer.h:240) -- its pages untouched
    RayVector in2 = rays, mv2;
    mv2.reserve(in2.size() * 10);
    const auto t0 = std::chrono::steady_clock::now();
    adapter->trace(in2, mv2, &m, &minv, &normi, lights);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (best_ms < 0 || ms < best_ms) best_ms = ms;
  }
  if (reps) {
    RayVector mv3; // (b) a moved_rays vector whose capacity is kept between the calls (pages already touched): the call's transfers and kernels alone
    mv3.reserve(rays.size() * 10);
    for (int r = 0; r < reps + 1; r++) {
      RayVector in2 = rays;
      mv3.clear();
      const auto t0 = std::chrono::steady_clock::now();
      adapter->trace(in2, mv3, &m, &minv, &normi, lights);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (r > 0 && (best_reused_ms < 0 || ms < best_reused_ms)) best_reused_ms = ms;
    }
  }
  RayVector moved;
  moved.reserve(rays.size() * 10); // ImageTracer.h:240
  adapter->trace(rays, moved, &m, &minv, &normi, lights);
  if (reps) std::printf("dropin_demo: trace_ms %.3f (best of %d) for %zu rays in, %zu moved; reused_ms %.3f with moved_rays' capacity kept between calls\n", best_ms, reps,
                        rays.size(), moved.size(), best_reused_ms);

  std::ofstream out(argv[4], std::ios::binary);
  const unsigned long long n = moved.size();
  out.write((const char *)&n, 8);
  out.write((const char *)moved.data(), n * sizeof(Ray));
  out.write((const char *)rays.data(), rays.size() * sizeof(Ray));
  std::printf("dropin_demo: %zu rays in, %llu moved\n", rays.size(), n);
  return 0;
}
