// api.hip -- C ABI glue of include/gvt_hip.h: context, meshes, device ray queues, Adapter::trace.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "gvt_internal.h"

size_t trav_spill_ints_per_thread();
int trav_block_threads();
void abi_pool_destroy(void *);

static thread_local std::string g_err;
static Ctx g_default_ctx;
static thread_local Ctx *tl_ctx = nullptr;
#define g_ctx (gctx())
Ctx &gctx() { return tl_ctx ? *tl_ctx : g_default_ctx; }

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
extern "C" const char *gvt_hip_last_error(void) { return g_err.c_str(); }

static int ctx_setup(Ctx &C, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return GVT_HIP_ERR_NODEVICE; }
  if (device < 0 || device >= count) { set_error("device %d out of range (%d visible)", device, count); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
    return GVT_HIP_ERR_NODEVICE;
  }
  C.device = device;
  C.n_cu = prop.multiProcessorCount;
  HIPCHK(hipStreamCreateWithFlags(&C.own_stream, hipStreamNonBlocking));
  C.stream = C.own_stream;
  // traversal kernels: 24 KiB LDS per 256-thread block -> 6 blocks/CU; grid = resident blocks, waves pull work
  C.trav_blocks = std::max(C.n_cu, C.n_cu * 32 / (trav_block_threads() / 64)); // at most 8 waves per SIMD, whatever the block size
  const size_t spill_ints = (size_t)C.trav_blocks * trav_block_threads() * trav_spill_ints_per_thread();
  HIPCHK(hipMalloc((void **)&C.d_spill, spill_ints * sizeof(int)));
  HIPCHK(hipMalloc((void **)&C.d_counters, 64 * sizeof(unsigned)));
  HIPCHK(hipMemset(C.d_counters, 0, 64 * sizeof(unsigned)));
  HIPCHK(hipHostMalloc((void **)&C.h_pinned, 64 * sizeof(unsigned), hipHostMallocDefault));
  std::memset(C.h_pinned, 0, 64 * sizeof(unsigned));
  C.ready = true;
  return 0;
}

// The first launch from a translation unit loads its code object, rocPRIM sets itself up at its first call: the FIRST acceleration-structure build of a
// process used to take 13-18 ms where the second takes 4 (10 M triangles).  gvt_hip_init pays that once, off every build's clock: a 2,048-triangle mesh is
// built (every kernel of lbvh.hip + the radix sort and scan) and a handful of rays traced through it (trace.hip, api.hip), then everything is released and the
// statistics start from zero.  GVT_HIP_NO_WARMUP=1 skips it.
static void drain_events();
static void warm_up() {
  if (getenv("GVT_HIP_NO_WARMUP")) return;
  const size_t nT = 2048;
  std::vector<float> v(nT * 9);
  std::vector<int32_t> t(nT * 3);
  uint32_t r = 12345u;
  for (size_t i = 0; i < nT; i++) {
    float c[3];
    for (int k = 0; k < 3; k++) { r = r * 1664525u + 1013904223u; c[k] = (float)(r >> 8) * (1.0f / 16777216.0f); }
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) { r = r * 1664525u + 1013904223u; v[9 * i + 3 * j + k] = c[k] + ((float)(r >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.02f; }
    t[3 * i] = (int32_t)(3 * i); t[3 * i + 1] = (int32_t)(3 * i + 1); t[3 * i + 2] = (int32_t)(3 * i + 2);
  }
  gvt_hip_mesh *m = gvt_hip_mesh_create(v.data(), nT * 3, t.data(), nT, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
  if (m) {
    float org[64 * 3], dir[64 * 3];
    gvt_hip_hit hits[64];
    int32_t occ[64];
    for (int i = 0; i < 64; i++) { org[3 * i] = (float)(i % 8) / 8.f; org[3 * i + 1] = (float)(i / 8) / 8.f; org[3 * i + 2] = 2.f; dir[3 * i] = 0.f; dir[3 * i + 1] = 0.f; dir[3 * i + 2] = -1.f; }
    gvt_hip_intersect(m, org, dir, 64, 0.f, hits);
    gvt_hip_occluded(m, org, dir, 64, 0.f, occ);
    gvt_hip_mesh_destroy(m);
  }
  Ctx &C = g_default_ctx;
  hipStreamSynchronize(C.stream);
  drain_events();
  C.stats = gvt_hip_stats{};
  set_error("%s", ""); // (a failed warm-up is not an error of the caller's: the first real call reports its own)
}

extern "C" int gvt_hip_init(int device) {
  Ctx &C = g_default_ctx;
  if (C.ready && C.device == device) { if (!tl_ctx) HIPCHK(hipSetDevice(device)); return 0; }
  if (C.ready) { set_error("gvt_hip_init: already initialised on device %d", C.device); return GVT_HIP_ERR_INVALID; }
  const int rc = ctx_setup(C, device);
  if (!rc && !tl_ctx) warm_up();
  return rc;
}

// ---- additional contexts: one per thread that wants its own stream / scratch / statistics (e.g. several ranks of a scheduler in
//      one process).  HIP's current device is per thread, so make_current also selects the context's device. ----
extern "C" gvt_hip_ctx *gvt_hip_ctx_create(int device) {
  Ctx *C = new Ctx();
  if (ctx_setup(*C, device) != 0) { delete C; return nullptr; }
  return (gvt_hip_ctx *)C;
}
extern "C" int gvt_hip_ctx_make_current(gvt_hip_ctx *c) {
  tl_ctx = (Ctx *)c;
  Ctx &C = gctx();
  if (C.ready) HIPCHK(hipSetDevice(C.device));
  return 0;
}
extern "C" void gvt_hip_ctx_destroy(gvt_hip_ctx *c) {
  Ctx *C = (Ctx *)c;
  if (!C) return;
  if (tl_ctx == C) tl_ctx = nullptr;
  if (C->ready) {
    hipSetDevice(C->device);
    hipStreamSynchronize(C->stream);
    for (auto &p : C->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : C->event_pool) hipEventDestroy(e);
    if (C->abi_qin) gvt_hip_queue_destroy(C->abi_qin);
    if (C->abi_qout) gvt_hip_queue_destroy(C->abi_qout);
    abi_pool_destroy(C->abi_pool); C->abi_pool = nullptr; // (joins the lane threads)
    for (Ctx *L : C->abi_lanes) gvt_hip_ctx_destroy((gvt_hip_ctx *)L);
    C->abi_lanes.clear();
    hipSetDevice(C->device);
    for (int k = 0; k < 24; k++) if (C->scratch[k]) hipFree(C->scratch[k]);
    hipFree(C->d_spill); hipFree(C->d_counters); hipHostFree(C->h_pinned);
    hipStreamDestroy(C->own_stream);
  }
  delete C;
}

int ensure_init() {
  if (gctx().ready) return 0;
  return gvt_hip_init(0);
}

extern "C" int gvt_hip_set_stream(void *s) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  g_ctx.stream = s ? (hipStream_t)s : g_ctx.own_stream;
  return 0;
}
extern "C" int gvt_hip_synchronize(void) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

void *scratch_get(int slot, size_t bytes) {
  Ctx &C = g_ctx;
  if (bytes <= C.scratch_bytes[slot] && C.scratch[slot]) return C.scratch[slot];
  hipStreamSynchronize(C.stream);
  if (C.scratch[slot]) hipFree(C.scratch[slot]);
  size_t want = bytes + bytes / 4 + 4096;
  C.scratch[slot] = nullptr;
  C.scratch_bytes[slot] = 0;
  if (hipMalloc(&C.scratch[slot], want) != hipSuccess) { set_error("scratch hipMalloc(%zu) failed", want); return nullptr; }
  C.scratch_bytes[slot] = want;
  return C.scratch[slot];
}

void scratch_release(int slot) {
  Ctx &C = g_ctx;
  if (!C.scratch[slot]) return;
  hipStreamSynchronize(C.stream);
  hipFree(C.scratch[slot]);
  C.scratch[slot] = nullptr;
  C.scratch_bytes[slot] = 0;
}

// ---- profiling: HIP events on the launch stream ----
ProfScope::ProfScope(int c) : cls(c) {
  Ctx &C = g_ctx;
  if (!C.profile || (C.profile == 2 && c != KC_CLOSEST && c != KC_ANY && c != KC_LONG) || (C.profile == 3 && c != KC_CLOSEST) || (C.profile == 4 && c != KC_ANY)) return;
  auto take = [&]() {
    hipEvent_t e;
    if (!C.event_pool.empty()) { e = C.event_pool.back(); C.event_pool.pop_back(); }
    else hipEventCreate(&e);
    return e;
  };
  a = take(); b = take();
  hipEventRecord(a, C.stream);
}
ProfScope::~ProfScope() {
  Ctx &C = g_ctx;
  if (!a) return;
  hipEventRecord(b, C.stream);
  C.pending.push_back({ a, b, cls });
}
static void drain_events() {
  Ctx &C = g_ctx;
  for (auto &p : C.pending) {
    hipEventSynchronize(p.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, p.a, p.b);
    double *slot[KC_COUNT] = { &C.stats.ms_closest, &C.stats.ms_any, &C.stats.ms_shade, &C.stats.ms_convert, &C.stats.ms_shuffle,
                               &C.stats.ms_camera, &C.stats.ms_build, &C.stats.ms_sort, &C.stats.ms_long };
    *slot[p.cls] += ms;
    C.event_pool.push_back(p.a);
    C.event_pool.push_back(p.b);
  }
  C.pending.clear();
}
extern "C" int gvt_hip_profile(int enable) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  drain_events();
  g_ctx.profile = enable < 0 ? 0 : enable;
  return 0;
}
extern "C" int gvt_hip_stats_read(gvt_hip_stats *out) {
  if (!out) { set_error("stats_read: null"); return GVT_HIP_ERR_INVALID; }
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  drain_events();
  *out = g_ctx.stats;
  return 0;
}
extern "C" int gvt_hip_stats_reset(void) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  drain_events();
  g_ctx.stats = gvt_hip_stats{};
  return 0;
}

// ---- tuning knobs: one table.  `shipped`: the knob is part of the product's surface -- a switch of behaviour (the reference's hop-by-hop
// shuffle rule, the sink, the list order), a budget (memory, chunk sizes, round sizes) or a test hook.  The others have ONE measured-best
// value (DESIGN.md / EXPERIMENTS.md); they can be moved only in the experiments build of the library (libgvt_hip_exp.so), where the knob
// sweeps run; the shipped library answers GVT_HIP_ERR_INVALID when one of them is switched away from its default.
namespace {
struct KnobDef { const char *name; int Knobs::*field; int lo, hi; bool shipped; };
const KnobDef g_knobs[] = {
  // shipped (27)
  { "skip_known", &Knobs::skip_known, 0, 1, true },           { "frame_timing", &Knobs::frame_timing, 0, 1, true },
  { "term_sink", &Knobs::term_sink, 0, 1, true },
  { "sort_rays", &Knobs::sort_rays, 0, 1, true },             { "leaf_max", &Knobs::leaf_max, 1, 4, true },
  { "long_steps", &Knobs::long_steps, 0, 1 << 20, true },     { "small_rays", &Knobs::small_rays, 0, 1 << 30, true },
  { "finish_rays", &Knobs::finish_rays, 0, 1 << 30, true },   { "round_room_mb", &Knobs::round_room_mb, 0, 1 << 30, true },
  { "abi_lanes", &Knobs::abi_lanes_n, 0, 8, true },           { "abi_chunk", &Knobs::abi_chunk, 16384, 1 << 30, true },
  { "inject_fail_tick", &Knobs::inject_fail_tick, -1, 1 << 30, true },
  { "long_min_rays", &Knobs::long_min_rays, 0, 1 << 30, true }, { "long_auto", &Knobs::long_auto, 0, 1, true },
  { "finish_auto", &Knobs::finish_auto, 0, 1, true },
  { "payload_overlap_kb", &Knobs::payload_overlap_kb, 0, 1 << 30, true },
  { "packet", &Knobs::packet, 0, 2, true },                   { "packet_sah_max", &Knobs::packet_sah_max, 0, 1 << 30, true },
  { "packet_min_rays", &Knobs::packet_min_rays, 0, 1 << 30, true },
  { "shadow_order", &Knobs::shadow_order, 0, 1, true },       { "shadow_order_min_rays", &Knobs::shadow_order_min_rays, 0, 1 << 30, true },

  { "inline_kb", &Knobs::inline_kb, 0, 1024, true },          { "comm_cus", &Knobs::comm_cus, 0, 128, true },
  { "comm_stream", &Knobs::comm_stream, 0, 1, true },         { "spec_ticks", &Knobs::spec_ticks, 0, 1, true },
  { "finish_clusters", &Knobs::finish_clusters, 0, 1, true }, { "hop_local", &Knobs::hop_local, 0, 3, true },
  // experiments build only: the alternative was measured and lost, or the value is a tuned constant
  { "trav_kernel", &Knobs::trav_kernel, 0, 1, false },        { "wide4", &Knobs::wide4, 0, 1, false },
  { "coop_fetch", &Knobs::coop_fetch, 0, 1, false },          { "fused", &Knobs::fused, 0, 1, false },
  { "quad", &Knobs::quad, 0, 1, false },
  { "quad_inner_min", &Knobs::quad_inner_min, 1, 16, false }, { "quad_refill_min", &Knobs::quad_refill_min, 1, 16, false },
  { "blocks_per_cu_quad", &Knobs::blocks_per_cu_quad, 1, 8, false },
  { "blocks_per_cu", &Knobs::blocks_per_cu, 1, 8, false },    { "blocks_per_cu_closest", &Knobs::blocks_per_cu_closest, 0, 8, false },
  { "refill_min", &Knobs::refill_min, 1, 64, false },         { "inner_min", &Knobs::inner_min, 1, 64, false },
  { "share", &Knobs::share, 0, 3, false },                    { "share_min_rays", &Knobs::share_min_rays, 0, 1 << 30, false },
  { "sort_gather", &Knobs::sort_gather, 0, 1, false },        { "sort_bits", &Knobs::sort_bits, 8, 32, false },
  { "long_steps_drain", &Knobs::long_steps_drain, 0, 1 << 20, false }, { "long_save", &Knobs::long_save, 0, 1, false },
  { "lean_frame", &Knobs::lean_frame, 0, 1, false },          { "report_poll", &Knobs::report_poll, 0, 1, false },
  { "first_round_async", &Knobs::first_round_async, 0, 1, false }, { "wave_single", &Knobs::wave_single, 0, 1, false },
  { "shadow_direct", &Knobs::shadow_direct, 0, 1, false },    { "top_ordered", &Knobs::top_ordered, 0, 1, false },
  { "top_lds", &Knobs::top_lds, 0, 1, false },                { "camera_tile", &Knobs::camera_tile, 0, 8, false },
  { "abi_pipe_min", &Knobs::abi_pipe_min, 0, 1 << 30, false },
  { "shadow_cls_lo", &Knobs::shadow_cls_lo, 0, 255, false },  { "shadow_cls_shift", &Knobs::shadow_cls_shift, 0, 6, false },
  { "fused1", &Knobs::fused1, 0, 1, false },                  { "fused1_min_rays", &Knobs::fused1_min_rays, 0, 1 << 30, false },
};
} // namespace

extern "C" int gvt_hip_set_option(const char *name, int value) {
  if (!name) { set_error("set_option: null name"); return GVT_HIP_ERR_INVALID; }
  if (!std::strcmp(name, "defaults")) { static_cast<Knobs &>(g_ctx) = Knobs{}; return 0; }
  for (const KnobDef &k : g_knobs) {
    if (std::strcmp(name, k.name)) continue;
    if (value < k.lo || value > k.hi) { set_error("set_option: %s must be %d..%d", name, k.lo, k.hi); return GVT_HIP_ERR_INVALID; }
    if (!std::strcmp(name, "camera_tile") && value != 0 && value != 8) { set_error("camera_tile: 0 or 8"); return GVT_HIP_ERR_INVALID; }
#ifndef GVT_EXPERIMENTS
    if (!k.shipped && value != Knobs{}.*(k.field)) {
      set_error("set_option: %s=%d needs the experiments build of the library (libgvt_hip_exp.so); the shipped library carries its measured-best value %d", name, value,
                Knobs{}.*(k.field));
      return GVT_HIP_ERR_INVALID;
    }
#endif
    static_cast<Knobs &>(g_ctx).*(k.field) = value;
    return 0;
  }
  set_error("set_option: unknown option '%s'", name);
  return GVT_HIP_ERR_INVALID;
}

// ---- meshes ----
static void default_material(gvt_hip_material *m) { // Material.h:62-77
  std::memset(m, 0, sizeof *m);
  m->type = 0;
  m->kd[0] = m->kd[1] = m->kd[2] = .5f;
  m->ks[0] = m->ks[1] = m->ks[2] = .5f;
  m->alpha = 1.f;
  m->eta[0] = .19f; m->eta[1] = 1.45f; m->eta[2] = 1.50f;
  m->k[0] = 3.06f; m->k[1] = 2.40f; m->k[2] = 1.88f;
  m->roughness = 0.05f;
}

// Mesh::generateNormals (Mesh.cpp:116-154).  Host side, one-off per mesh like the reference; the sum is
// order dependent (unweighted accumulation in face order), so it is kept sequential for bit parity.
static void generate_normals(const float *verts, size_t nV, const int32_t *tris, size_t nT, std::vector<float> &normals) {
  normals.assign(nV * 3, 0.0f);
  for (size_t i = 0; i < nT; i++) {
    int I = tris[3 * i], J = tris[3 * i + 1], K = tris[3 * i + 2];
    V3 a = ld3(verts + 3 * I), b = ld3(verts + 3 * J), c = ld3(verts + 3 * K);
    V3 u = sub3(b, a), v = sub3(c, a), n;
    n.x = u.y * v.z - u.z * v.y;
    n.y = u.z * v.x - u.x * v.z;
    n.z = u.x * v.y - u.y * v.x;
    n = norm3(n);
    int idx[3] = { I, J, K };
    for (int k = 0; k < 3; k++) {
      float *d = &normals[3 * idx[k]];
      d[0] += n.x; d[1] += n.y; d[2] += n.z;
    }
  }
  for (size_t i = 0; i < nV; i++) {
    V3 n = norm3(ld3(&normals[3 * i]));
    normals[3 * i] = n.x; normals[3 * i + 1] = n.y; normals[3 * i + 2] = n.z;
  }
}

template <typename T> static bool upload(T **dst, const T *src, size_t n) {
  *dst = nullptr;
  if (hipMalloc((void **)dst, sizeof(T) * (n ? n : 1)) != hipSuccess) return false;
  if (n && hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) return false;
  return true;
}

extern "C" gvt_hip_mesh *gvt_hip_mesh_create(const float *verts, size_t nV, const int32_t *tris, size_t nT, const float *vnormals,
                                             const float *vcolors, const gvt_hip_material *materials, size_t nMat,
                                             const int32_t *face_mat, const gvt_hip_material *mesh_mat) {
  if (ensure_init()) return nullptr;
  if ((nV && !verts) || (nT && !tris)) { set_error("mesh_create: null vertex/triangle array"); return nullptr; } // GVT_ASSERT :128
  if (nT >= (1u << 28)) { set_error("mesh_create: %zu triangles exceed the 2^28 leaf-slot encoding", nT); return nullptr; }
  for (size_t i = 0; i < nT * 3; i++)
    if (tris[i] < 0 || (size_t)tris[i] >= nV) { set_error("mesh_create: triangle %zu references vertex %d (nV=%zu)", i / 3, tris[i], nV); return nullptr; }
  if (face_mat)
    for (size_t i = 0; i < nT; i++)
      if (face_mat[i] >= 0 && (size_t)face_mat[i] >= nMat) { set_error("mesh_create: face_mat[%zu]=%d out of range", i, face_mat[i]); return nullptr; }
  gvt_hip_mesh *M = new gvt_hip_mesh();
  M->nV = nV; M->nT = nT; M->nMat = materials ? nMat : 0;
  if (mesh_mat) M->mesh_mat = *mesh_mat; else default_material(&M->mesh_mat);
  std::vector<float> gen;
  const float *nrm = vnormals;
  if (!nrm) { generate_normals(verts, nV, tris, nT, gen); nrm = gen.data(); } // EmbreeMeshAdapter.cpp:129
  bool ok = upload(&M->d_verts, verts, nV * 3) && upload(&M->d_tris, tris, nT * 3) && upload(&M->d_normals, nrm, nV * 3);
  if (ok && vcolors) ok = upload(&M->d_vcolors, vcolors, nV * 3);
  if (ok && M->nMat) ok = upload(&M->d_materials, materials, M->nMat);
  if (ok && face_mat && M->nMat) ok = upload(&M->d_face_mat, face_mat, nT);
  if (!ok) { set_error("mesh_create: device upload failed (%s)", hipGetErrorString(hipGetLastError())); gvt_hip_mesh_destroy(M); return nullptr; }
  if (build_lbvh(M) != 0) { gvt_hip_mesh_destroy(M); return nullptr; }
  return M;
}

extern "C" void gvt_hip_mesh_destroy(gvt_hip_mesh *M) {
  if (!M) return;
  if (g_ctx.ready) hipStreamSynchronize(g_ctx.stream);
  hipFree(M->d_verts); hipFree(M->d_tris); hipFree(M->d_normals); hipFree(M->d_vcolors); hipFree(M->d_materials);
  hipFree(M->d_face_mat); hipFree(M->d_nodes); hipFree(M->d_tri); hipFree(M->d_slot_of); hipFree(M->d_nodes4); hipFree(M->d_nodes4c); hipFree(M->d_nodes4q); hipFree(M->d_triq);
  delete M;
}

extern "C" int gvt_hip_mesh_get_info(const gvt_hip_mesh *M, gvt_hip_mesh_info *o) {
  if (!M || !o) { set_error("mesh_get_info: null"); return GVT_HIP_ERR_INVALID; }
  std::memset(o, 0, sizeof *o);
  o->n_tris = M->nT; o->n_verts = M->nV; o->n_nodes = M->nNodes; o->n_leaves = M->nLeaves;
  for (int k = 0; k < 3; k++) { o->bbox_lo[k] = M->lo[k]; o->bbox_hi[k] = M->hi[k]; }
  o->build_ms = M->build_ms; o->max_leaf = M->leaf_max; o->packet = M->packet_ok ? 1u : 0u; o->sah_inner = M->sah_inner;
  o->bytes_nodes = M->nNodes * sizeof(BvhNode) + M->nNodes4 * 64; o->bytes_tris = M->nT * 64;
  return 0;
}
extern "C" int gvt_hip_mesh_get_normals(const gvt_hip_mesh *M, float *out) {
  if (!M || !out) { set_error("mesh_get_normals: null"); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipMemcpy(out, M->d_normals, sizeof(float) * 3 * M->nV, hipMemcpyDeviceToHost));
  return 0;
}

// ---- device ray queues ----
int queue_reserve(gvt_hip_queue *q, size_t cap) {
  if (cap <= q->cap) return 0;
  Ctx &C = g_ctx;
  size_t ncap = cap + cap / 8 + 1024;
  if (ncap >= 0xffffffffull) { set_error("queue_reserve: %zu rays exceed the 32-bit slot counter", ncap); return GVT_HIP_ERR_INVALID; }
  float4 *np = nullptr;
  HIPCHK(hipMalloc((void **)&np, GVT_QUEUE_BYTES_PER_RAY * ncap));
  if (q->size) {
    for (int k = 0; k < 4; k++)
      HIPCHK(hipMemcpyAsync(np + (size_t)k * ncap, q->d_planes + (size_t)k * q->cap, sizeof(float4) * q->size, hipMemcpyDeviceToDevice, C.stream));
    HIPCHK(hipMemcpyAsync(np + 4 * ncap, q->d_planes + 4 * q->cap, sizeof(uint32_t) * q->size, hipMemcpyDeviceToDevice, C.stream));
    HIPCHK(hipMemcpyAsync((uint32_t *)(np + 4 * ncap) + ncap, (const uint32_t *)(q->d_planes + 4 * q->cap) + q->cap, 3 * sizeof(uint32_t) * q->size, hipMemcpyDeviceToDevice, C.stream));
  }
  HIPCHK(hipStreamSynchronize(C.stream));
  if (q->d_planes) HIPCHK(hipFree(q->d_planes));
  q->d_planes = np;
  q->cap = ncap;
  return 0;
}

extern "C" gvt_hip_queue *gvt_hip_queue_create(size_t capacity) {
  if (ensure_init()) return nullptr;
  gvt_hip_queue *q = new gvt_hip_queue();
  if (hipMalloc((void **)&q->d_count, 64) != hipSuccess || hipMemset(q->d_count, 0, 64) != hipSuccess) {
    set_error("queue_create: hipMalloc failed"); delete q; return nullptr;
  }
  if (capacity && queue_reserve(q, capacity) != 0) { gvt_hip_queue_destroy(q); return nullptr; }
  return q;
}
extern "C" void gvt_hip_queue_destroy(gvt_hip_queue *q) {
  if (!q) return;
  if (g_ctx.ready) hipStreamSynchronize(g_ctx.stream);
  hipFree(q->d_planes); hipFree(q->d_count);
  delete q;
}
extern "C" int gvt_hip_queue_reserve(gvt_hip_queue *q, size_t cap) {
  if (!q) { set_error("queue_reserve: null"); return GVT_HIP_ERR_INVALID; }
  return queue_reserve(q, cap);
}
extern "C" int gvt_hip_queue_clear(gvt_hip_queue *q) {
  if (!q) { set_error("queue_clear: null"); return GVT_HIP_ERR_INVALID; }
  q->size = 0;
  HIPCHK(hipMemsetAsync(q->d_count, 0, sizeof(unsigned), g_ctx.stream));
  return 0;
}
extern "C" int gvt_hip_queue_size(gvt_hip_queue *q, size_t *n) {
  if (!q || !n) { set_error("queue_size: null"); return GVT_HIP_ERR_INVALID; }
  *n = q->size;
  return 0;
}
extern "C" int gvt_hip_queue_sizes(gvt_hip_queue *const *queues, size_t n, uint64_t *out) {
  if ((n && !queues) || !out) { set_error("queue_sizes: null"); return GVT_HIP_ERR_INVALID; }
  for (size_t i = 0; i < n; i++) out[i] = queues[i] ? queues[i]->size : 0;
  return 0;
}

extern "C" int gvt_hip_abi_version(void) { return GVT_HIP_ABI_VERSION; }

extern "C" int gvt_hip_queue_append(gvt_hip_queue *q, const gvt_hip_ray *rays, size_t n, int src_on_device) {
  if (src_on_device != 0 && src_on_device != 1) { // a caller of revision 5 passing its flag word: an error, never a host pointer read as device memory
    set_error("queue_append: the 4th argument is the boolean src_on_device (0 / 1), got %d; flag words go to gvt_hip_queue_append_flags", src_on_device);
    return GVT_HIP_ERR_INVALID;
  }
  return gvt_hip_queue_append_flags(q, rays, n, src_on_device ? (GVT_HIP_APPEND_DEVICE | GVT_HIP_APPEND_KEEP_STATE) : 0);
}
extern "C" int gvt_hip_queue_append_flags(gvt_hip_queue *q, const gvt_hip_ray *rays, size_t n, int flags) {
  if (!q || (n && !rays)) { set_error("queue_append: null"); return GVT_HIP_ERR_INVALID; }
  if (flags & ~(GVT_HIP_APPEND_DEVICE | GVT_HIP_APPEND_KEEP_STATE)) { set_error("queue_append: unknown flag bits %d", flags); return GVT_HIP_ERR_INVALID; }
  const int src_on_device = flags & GVT_HIP_APPEND_DEVICE;
  if (!n) return 0;
  Ctx &C = g_ctx;
  int rc = queue_reserve(q, q->size + n);
  if (rc) return rc;
  const gvt_hip_ray *d_src = rays;
  if (!src_on_device) {
    void *stage = scratch_get(0, sizeof(gvt_hip_ray) * n);
    if (!stage) return GVT_HIP_ERR_DEVICE;
    HIPCHK(hipMemcpyAsync(stage, rays, sizeof(gvt_hip_ray) * n, hipMemcpyHostToDevice, C.stream));
    d_src = (const gvt_hip_ray *)stage;
  }
  rc = convert_aos_to_planes(d_src, n, make_planes(q->d_planes, q->cap), q->size, (flags & GVT_HIP_APPEND_KEEP_STATE) != 0);
  if (rc) return rc;
  q->size += n;
  if ((rc = set_device_u32(q->d_count, (unsigned)q->size))) return rc;
  if (!src_on_device) HIPCHK(hipStreamSynchronize(C.stream)); // the staging copy reads the caller's host buffer
  return 0;
}

extern "C" int gvt_hip_queue_export(gvt_hip_queue *q, gvt_hip_ray *dst, size_t cap, size_t *n, int dst_on_device) {
  if (!q || !n) { set_error("queue_export: null"); return GVT_HIP_ERR_INVALID; }
  *n = q->size;
  if (q->size > cap) { set_error("queue_export: %zu rays, capacity %zu", q->size, cap); return GVT_HIP_ERR_CAPACITY; }
  if (!q->size) return 0;
  if (!dst) { set_error("queue_export: null destination"); return GVT_HIP_ERR_INVALID; }
  Ctx &C = g_ctx;
  gvt_hip_ray *d_dst = dst;
  if (!dst_on_device) {
    d_dst = (gvt_hip_ray *)scratch_get(0, sizeof(gvt_hip_ray) * q->size);
    if (!d_dst) return GVT_HIP_ERR_DEVICE;
  }
  int rc = convert_planes_to_aos(make_planes(q->d_planes, q->cap), 0, q->size, d_dst);
  if (rc) return rc;
  if (!dst_on_device) HIPCHK(hipMemcpyAsync(dst, d_dst, sizeof(gvt_hip_ray) * q->size, hipMemcpyDeviceToHost, C.stream));
  HIPCHK(hipStreamSynchronize(C.stream));
  return 0;
}

// ---- Adapter::trace ----
static int fill_params(TraceParams &P, const float m[16], const float minv[16], const float normi[9], size_t n_lights, int normal_mode,
                       uint32_t seed, const gvt_hip_light *lights) {
  if (!m || !minv || !normi) { set_error("trace: null matrix"); return GVT_HIP_ERR_INVALID; }
  if (n_lights && !lights) { set_error("trace: null lights"); return GVT_HIP_ERR_INVALID; }
  if (n_lights > 64) { set_error("trace: %zu lights (max 64)", n_lights); return GVT_HIP_ERR_INVALID; }
  if (normal_mode != GVT_HIP_NORMALS_FLAT && normal_mode != GVT_HIP_NORMALS_SMOOTH) { set_error("trace: bad normal_mode %d", normal_mode); return GVT_HIP_ERR_INVALID; }
  std::memcpy(P.m.m, m, 64); std::memcpy(P.minv.m, minv, 64); std::memcpy(P.normi.n, normi, 36);
  P.normal_mode = normal_mode; P.seed = seed; P.n_lights = (int)n_lights;
  P.sink = TermSink{};
  P.update_in_place = 1;
  P.carried_rng = 0;
  return 0;
}

extern "C" int gvt_hip_trace_queue(gvt_hip_mesh *M, gvt_hip_queue *q_in, gvt_hip_queue *q_out, const float m[16], const float minv[16],
                                   const float normi[9], const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed) {
  return gvt_hip_trace_queue_sink(M, q_in, q_out, m, minv, normi, lights, n_lights, normal_mode, seed, nullptr, -1, nullptr);
}

extern "C" int gvt_hip_trace_queue_sink(gvt_hip_mesh *M, gvt_hip_queue *q_in, gvt_hip_queue *q_out, const float m[16], const float minv[16],
                                        const float normi[9], const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed,
                                        gvt_hip_top *top, int from_inst, gvt_hip_fb *fb) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !q_in || !q_out || q_in == q_out) { set_error("trace_queue: null or aliased queue"); return GVT_HIP_ERR_INVALID; }
  TraceParams P;
  int rc = fill_params(P, m, minv, normi, n_lights, normal_mode, seed, lights);
  if (rc) return rc;
  if (top && fb && g_ctx.term_sink) {
    P.sink.top = top->dev(); P.sink.from = from_inst;
    P.sink.fb = fb->d_rgba; P.sink.n_pix = (unsigned)(fb->w * fb->h);
  }
  const size_t n = q_in->size;
  rc = queue_reserve(q_out, q_out->size + n * (1 + n_lights));
  if (rc) return rc;
  P.update_in_place = 0; // q_in is cleared below: nobody reads the updated rays
  P.carried_rng = 1;     // device-resident rays own their RNG stream
  rc = trace_core(M, make_planes(q_in->d_planes, q_in->cap), n, 0, q_out, P, lights);
  if (rc) return rc;
  return gvt_hip_queue_clear(q_in); // the caller's queue[instTarget].clear(), ImageTracer.h:248
}

extern "C" int gvt_hip_trace(gvt_hip_mesh *M, gvt_hip_ray *rays, size_t n_rays, size_t begin, size_t end, gvt_hip_ray *rays_out, size_t cap,
                             size_t *n_out, const float m[16], const float minv[16], const float normi[9], const gvt_hip_light *lights,
                             size_t n_lights, int normal_mode, uint32_t seed) {
  return gvt_hip_trace_ex(M, rays, n_rays, begin, end, rays_out, cap, n_out, m, minv, normi, lights, n_lights, normal_mode, seed, 0u);
}

// Adapter::trace on a HOST RayVector, pipelined like the reference's own GPU adapter (adapter/optix/OptixMeshAdapter.cpp:622-684: the
// list cut into packets on two streams so that upload, trace and download overlap).  The list is cut into chunks; `abi_lanes` host
// threads, each with a context of its own (stream, scratch, counters, staging queues), take chunks in turn: upload of the 80-byte rays
// + conversion to planes, the closest / shade / any chain, download of the updated rayList slice and of the chunk's moved rays to
// their place behind the moved rays of the chunks before it (a running offset; the order of moved_rays is unspecified in the
// reference too, EmbreeMeshAdapter.cpp:619-621).  While one lane waits for its chain or its download, another uploads: the two
// directions of the link and the kernels overlap.  Results do not depend on the cut: a ray's RNG stream is keyed on its index in
// rayList (trace_core's index_base).  Used when rays_out has room for the worst case n * (1 + n_lights), so that a capacity error
// cannot arise after the rayList has begun to change; smaller buffers take the one-shot path below.
// The lanes are PERSISTENT threads of the calling context (created at the first pipelined call, parked on a condition variable between
// calls, joined when the context is destroyed; the default context's are never joined -- the process ends with them parked).
struct AbiPool {
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> threads;
  std::function<void(int)> job; // the current call's work, by lane number
  uint64_t generation = 0;      // bumped per call
  int wanted = 0, running = 0;  // lanes that take part in the current call / have not finished it yet
  bool stop = false;
  void worker(int li) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return stop || (generation != seen && li < wanted); });
        if (stop) return;
        seen = generation;
        j = job;
      }
      j(li);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }
  // runs job(0..n-1) on n lanes and waits for them; false: the threads could not be created (the caller falls back to the one-shot path)
  bool run(int n, std::function<void(int)> f) {
    try {
      while ((int)threads.size() < n) { const int li = (int)threads.size(); threads.emplace_back([this, li] { worker(li); }); }
    } catch (...) { return false; }
    std::unique_lock<std::mutex> lk(mu);
    job = std::move(f); wanted = n; running = n; generation++;
    cv_go.notify_all();
    cv_done.wait(lk, [&] { return running == 0; });
    job = nullptr;
    return true;
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_go.notify_all();
    for (auto &t : threads) if (t.joinable()) t.join();
    threads.clear();
  }
};
void abi_pool_destroy(void *p) { AbiPool *P = (AbiPool *)p; if (P) { P->shutdown(); delete P; } }

static int trace_pipelined(Ctx &C, gvt_hip_mesh *M, gvt_hip_ray *rays, size_t begin, size_t n, gvt_hip_ray *rays_out, size_t cap, size_t *n_out, const TraceParams &P,
                           const gvt_hip_light *lights, size_t n_lights, bool write_back, bool *fell_back) {
  const int L = C.abi_lanes_n < 1 ? 1 : (C.abi_lanes_n > 8 ? 8 : C.abi_lanes_n);
  while ((int)C.abi_lanes.size() < L) {
    Ctx *LC = (Ctx *)gvt_hip_ctx_create(C.device);
    if (!LC) return GVT_HIP_ERR_DEVICE;
    C.abi_lanes.push_back(LC);
  }
  if (!C.abi_pool) C.abi_pool = new (std::nothrow) AbiPool();
  if (!C.abi_pool) { *fell_back = true; return 0; }
  const size_t chunk = (size_t)(C.abi_chunk < 16384 ? 16384 : C.abi_chunk);
  const size_t n_chunks = (n + chunk - 1) / chunk;
  std::atomic<size_t> next{ 0 }, out_pos{ 0 }, traced{ 0 };
  std::atomic<int> err{ 0 };
  std::mutex mu; // (one upload / one download at a time -- a mutex per direction, 128 K-ray chunks -- was measured: 6.1 ms against 5.0-5.6 with the lanes'
  std::string err_msg; // copies left to share the link: each chunk's conversion kernel and synchronisation then sit on the link's critical path)
  Ctx *caller_ctx = &C;
  static const bool trace_chunks = getenv("GVT_HIP_ABI_TRACE") != nullptr;
  const auto t_call = std::chrono::steady_clock::now();
  auto work = [&](int li) {
    Ctx *LC = C.abi_lanes[(size_t)li];
    gvt_hip_ctx_make_current((gvt_hip_ctx *)LC);
    static_cast<Knobs &>(*LC) = static_cast<const Knobs &>(*caller_ctx); // the caller's tuning knobs
    LC->profile = 0;
    const gvt_hip_stats before = LC->stats;
    int rc = 0;
    if (!LC->abi_qin) { LC->abi_qin = gvt_hip_queue_create(0); LC->abi_qout = gvt_hip_queue_create(0); }
    gvt_hip_queue *qin = LC->abi_qin, *qout = LC->abi_qout;
    if (!qin || !qout) rc = GVT_HIP_ERR_DEVICE;
    while (!rc && !err.load()) {
      const size_t k = next.fetch_add(1);
      if (k >= n_chunks) break;
      const size_t off = k * chunk, cn = std::min(chunk, n - off);
      size_t got = 0;
      const auto tc0 = std::chrono::steady_clock::now();
      if ((rc = gvt_hip_queue_clear(qin)) || (rc = gvt_hip_queue_clear(qout))) break;
      if ((rc = gvt_hip_queue_append_flags(qin, rays + begin + off, cn, GVT_HIP_APPEND_KEEP_STATE))) break; // (Adapter::trace copies a forwarded ray whole, padding included, and interprets none of it)
      const auto tc1 = std::chrono::steady_clock::now();
      if ((rc = queue_reserve(qout, cn * (1 + n_lights)))) break;
      if ((rc = trace_core(M, make_planes(qin->d_planes, qin->cap), cn, begin + off, qout, P, lights))) break;
      traced.fetch_add(1);
      const size_t moved = qout->size, pos = out_pos.fetch_add(moved);
      if (pos + moved > cap) { // only rays that bounce (depth > 1) can emit more than the n * (1 + lights) the caller was asked to provide
        set_error("trace: more than %zu outgoing rays (rays that bounce emit shadow rays in every pass); rayList and rays_out are partly written", cap);
        rc = GVT_HIP_ERR_CAPACITY;
        break;
      }
      const auto tc2 = std::chrono::steady_clock::now();
      if (write_back && (rc = gvt_hip_queue_export(qin, rays + begin + off, cn, &got, 0))) break; // rayList is updated in place (r.mice.t, bounce state)
      const auto tc3 = std::chrono::steady_clock::now();
      if (moved && (rc = gvt_hip_queue_export(qout, rays_out + pos, moved, &got, 0))) break;
      if (trace_chunks) { // GVT_HIP_ABI_TRACE=1: where a chunk's time goes (ms since the call began)
        const auto ms = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - t_call).count(); };
        std::fprintf(stderr, "abi chunk %2zu lane %d: upload %.3f-%.3f trace -%.3f rayList back -%.3f moved (%zu) back -%.3f\n", k, li, ms(tc0), ms(tc1), ms(tc2), ms(tc3), moved,
                     ms(std::chrono::steady_clock::now()));
      }
    }
    if (rc) {
      std::lock_guard<std::mutex> lk(mu);
      if (!err.load()) { err.store(rc); err_msg = gvt_hip_last_error(); }
    }
    { // the lane's counters belong to the caller's context
      std::lock_guard<std::mutex> lk(mu);
      gvt_hip_stats &S = caller_ctx->stats;
      const gvt_hip_stats &A = LC->stats;
      S.rays_closest += A.rays_closest - before.rays_closest; S.rays_any += A.rays_any - before.rays_any; S.rays_shaded += A.rays_shaded - before.rays_shaded;
      S.rays_forwarded += A.rays_forwarded - before.rays_forwarded; S.launches_closest += A.launches_closest - before.launches_closest;
      S.launches_any += A.launches_any - before.launches_any; S.trace_calls += A.trace_calls - before.trace_calls;
    }
    gvt_hip_ctx_make_current(nullptr);
  };
  const int n_thr = (int)std::min<size_t>((size_t)L, n_chunks);
  if (!((AbiPool *)C.abi_pool)->run(n_thr, work)) { *fell_back = true; return 0; } // no threads to be had: nothing has been touched yet
  const uint64_t done = traced.load();
  if (done > 1) C.stats.trace_calls -= done - 1; // ONE Adapter::trace call, however many chunks were traced
  // A device error in the middle of a call leaves rayList partly updated and rays_out partly filled (only capacity errors are excluded
  // up front, by the worst-case room the pipelined path requires): the error is returned and *n_out stays 0
  if (err.load()) { set_error("%s", err_msg.c_str()); return err.load(); }
  *n_out = out_pos.load();
  return 0;
}

extern "C" int gvt_hip_trace_ex(gvt_hip_mesh *M, gvt_hip_ray *rays, size_t n_rays, size_t begin, size_t end, gvt_hip_ray *rays_out, size_t cap,
                                size_t *n_out, const float m[16], const float minv[16], const float normi[9], const gvt_hip_light *lights,
                                size_t n_lights, int normal_mode, uint32_t seed, uint32_t flags) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (flags & ~GVT_HIP_TRACE_NO_WRITEBACK) { set_error("trace: unknown flags 0x%x", flags); return GVT_HIP_ERR_INVALID; }
  const bool write_back = !(flags & GVT_HIP_TRACE_NO_WRITEBACK);
  if (!M || !n_out || (n_rays && !rays)) { set_error("trace: null argument"); return GVT_HIP_ERR_INVALID; }
  if (end == 0) end = n_rays; // EmbreeMeshAdapter.cpp:642
  if (begin > end || end > n_rays) { set_error("trace: bad range [%zu,%zu) of %zu", begin, end, n_rays); return GVT_HIP_ERR_INVALID; }
  TraceParams P;
  int rc = fill_params(P, m, minv, normi, n_lights, normal_mode, seed, lights);
  if (rc) return rc;
  Ctx &C = g_ctx;
  const size_t n = end - begin;
  *n_out = 0;
  if (!n) return 0;
  if (C.abi_lanes_n > 0 && n >= (size_t)C.abi_pipe_min && rays_out && cap >= n * (1 + n_lights)) {
    P.update_in_place = write_back ? 1 : 0;
    bool fell_back = false;
    rc = trace_pipelined(C, M, rays, begin, n, rays_out, cap, n_out, P, lights, n_lights, write_back, &fell_back);
    if (!fell_back) return rc;
  }
  if (!C.abi_qin) { C.abi_qin = gvt_hip_queue_create(0); C.abi_qout = gvt_hip_queue_create(0); }
  gvt_hip_queue *qin = C.abi_qin, *qout = C.abi_qout;
  if (!qin || !qout) return GVT_HIP_ERR_DEVICE;
  if ((rc = gvt_hip_queue_clear(qin)) || (rc = gvt_hip_queue_clear(qout))) return rc;
  if ((rc = gvt_hip_queue_append_flags(qin, rays + begin, n, GVT_HIP_APPEND_KEEP_STATE))) return rc; // bytes 64..79 pass through (a forwarded ray is a copy of all 80 bytes); this path never reads them
  if ((rc = queue_reserve(qout, n * (1 + n_lights)))) return rc;
  P.update_in_place = write_back ? 1 : 0;
  if ((rc = trace_core(M, make_planes(qin->d_planes, qin->cap), n, begin, qout, P, lights))) return rc;
  *n_out = qout->size;
  if (qout->size > cap) { set_error("trace: %zu outgoing rays, capacity %zu", qout->size, cap); return GVT_HIP_ERR_CAPACITY; }
  // rayList is updated in place (r.mice.t, bounce state)
  size_t got = 0;
  if (write_back && (rc = gvt_hip_queue_export(qin, rays + begin, n, &got, 0))) return rc;
  if (qout->size) {
    if (!rays_out) { set_error("trace: null rays_out"); return GVT_HIP_ERR_INVALID; }
    if ((rc = gvt_hip_queue_export(qout, rays_out, cap, &got, 0))) return rc;
  }
  return 0;
}

// ---- rtcIntersect / rtcOccluded equivalents on object-space rays ----
static int stage_od(const float *org, const float *dir, size_t n, RayPlanes &planes) {
  Ctx &C = g_ctx;
  float *d_org = (float *)scratch_get(2, sizeof(float) * 6 * n);
  float4 *d_pl = (float4 *)scratch_get(3, sizeof(float4) * 2 * n);
  if (!d_org || !d_pl) return GVT_HIP_ERR_DEVICE;
  float *d_dir = d_org + 3 * n;
  HIPCHK(hipMemcpyAsync(d_org, org, sizeof(float) * 3 * n, hipMemcpyHostToDevice, C.stream));
  HIPCHK(hipMemcpyAsync(d_dir, dir, sizeof(float) * 3 * n, hipMemcpyHostToDevice, C.stream));
  planes.p0 = d_pl; planes.p1 = d_pl + n; planes.p2 = nullptr; planes.p3 = nullptr; planes.p4 = nullptr; planes.p5 = nullptr;
  return convert_od_to_planes(d_org, d_dir, n, planes);
}

extern "C" int gvt_hip_intersect(gvt_hip_mesh *M, const float *org, const float *dir, size_t n, float tnear, gvt_hip_hit *hits) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || (n && (!org || !dir || !hits))) { set_error("intersect: null argument"); return GVT_HIP_ERR_INVALID; }
  if (!n) return 0;
  Ctx &C = g_ctx;
  RayPlanes pl;
  int rc = stage_od(org, dir, n, pl);
  if (rc) return rc;
  gvt_hip_hit *d_hits = (gvt_hip_hit *)scratch_get(0, sizeof(gvt_hip_hit) * n);
  if (!d_hits) return GVT_HIP_ERR_DEVICE;
  Mat4 id{};
  if ((rc = launch_closest(M, pl, nullptr, n, false, id, tnear, d_hits))) return rc;
  HIPCHK(hipMemcpyAsync(hits, d_hits, sizeof(gvt_hip_hit) * n, hipMemcpyDeviceToHost, C.stream));
  if ((rc = trav_overflow_fetch_async())) return rc;
  HIPCHK(hipStreamSynchronize(C.stream));
  return trav_overflow_result();
}

extern "C" int gvt_hip_occluded(gvt_hip_mesh *M, const float *org, const float *dir, size_t n, float tnear, int32_t *out) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || (n && (!org || !dir || !out))) { set_error("occluded: null argument"); return GVT_HIP_ERR_INVALID; }
  if (!n) return 0;
  Ctx &C = g_ctx;
  RayPlanes pl;
  int rc = stage_od(org, dir, n, pl);
  if (rc) return rc;
  int *d_flags = (int *)scratch_get(0, sizeof(int) * n);
  if (!d_flags) return GVT_HIP_ERR_DEVICE;
  Mat4 id{};
  if ((rc = launch_any_flags(M, pl, n, false, id, tnear, d_flags))) return rc;
  HIPCHK(hipMemcpyAsync(out, d_flags, sizeof(int) * n, hipMemcpyDeviceToHost, C.stream));
  if ((rc = trav_overflow_fetch_async())) return rc;
  HIPCHK(hipStreamSynchronize(C.stream));
  return trav_overflow_result();
}

// diagnostic: per-ray visit counts of the closest-hit traversal for object-space rays (see k_visit_stats)
extern "C" int gvt_hip_visit_stats(gvt_hip_mesh *M, const float *org, const float *dir, size_t n, float tnear, uint32_t *counts /* n*3 */) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || (n && (!org || !dir || !counts))) { set_error("visit_stats: null argument"); return GVT_HIP_ERR_INVALID; }
  if (!n) return 0;
  Ctx &C = g_ctx;
  RayPlanes pl;
  int rc = stage_od(org, dir, n, pl);
  if (rc) return rc;
  unsigned *d_out = (unsigned *)scratch_get(0, 3 * n * sizeof(unsigned));
  if (!d_out) return GVT_HIP_ERR_DEVICE;
  if ((rc = launch_visit_stats(M, pl, n, tnear, d_out))) return rc;
  HIPCHK(hipMemcpyAsync(counts, d_out, 3 * n * sizeof(unsigned), hipMemcpyDeviceToHost, C.stream));
  HIPCHK(hipStreamSynchronize(C.stream));
  return 0;
}

extern "C" int gvt_hip_wide_visit_stats(gvt_hip_mesh *M, const float *org, const float *dir, size_t n, float tnear, int width, uint32_t *counts /* n */,
                                        uint64_t *n_wide_nodes) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || (n && (!org || !dir || !counts)) || width < 2 || width > 8) { set_error("wide_visit_stats: null argument or width outside 2..8"); return GVT_HIP_ERR_INVALID; }
  if (!n || !M->nNodes) { for (size_t i = 0; i < n; i++) counts[i] = 0; if (n_wide_nodes) *n_wide_nodes = 0; return 0; }
  Ctx &C = g_ctx;
  unsigned char *d_marks = nullptr;
  HIPCHK(hipMalloc((void **)&d_marks, M->nNodes));
  size_t nw = 0;
  int rc = wide_root_marks(M, width, d_marks, &nw);
  RayPlanes pl;
  if (!rc) rc = stage_od(org, dir, n, pl);
  unsigned *d_out = rc ? nullptr : (unsigned *)scratch_get(0, n * sizeof(unsigned));
  if (!rc && !d_out) rc = GVT_HIP_ERR_DEVICE;
  if (!rc) rc = launch_wide_visit_stats(M, pl, n, tnear, d_marks, d_out);
  if (!rc && hipMemcpyAsync(counts, d_out, n * sizeof(unsigned), hipMemcpyDeviceToHost, C.stream) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (hipStreamSynchronize(C.stream) != hipSuccess && !rc) rc = GVT_HIP_ERR_DEVICE;
  hipFree(d_marks);
  if (n_wide_nodes) *n_wide_nodes = nw;
  return rc;
}

// diagnostics behind tools/wide_dp.py: the binary LBVH as the builder left it (64-byte nodes, gvt_device.h BvhNode), and the visit count of
// an ARBITRARY collapse of it -- marks[k] = 1 where binary node k is the root of a wide node (the caller's choice, e.g. a cost-optimal
// collapse computed on the host); counts[j] = marked nodes ray j's closest-hit traversal visits.
extern "C" int gvt_hip_mesh_download_nodes(gvt_hip_mesh *M, void *out, size_t n_nodes) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !out || n_nodes != M->nNodes) { set_error("mesh_download_nodes: null argument or n_nodes != %zu", M ? M->nNodes : (size_t)0); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  if (n_nodes) HIPCHK(hipMemcpy(out, M->d_nodes, n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
  return 0;
}
// (measurement) the traversal layout itself -- the compressed 4-wide nodes (gvt_device.h: 64 bytes each) and the triangle slots in leaf order (64 bytes
// each) -- for bench.py's SIMD CPU baseline (oracle/simd_baseline.c), which walks the very tree the kernels walk, on the host's cores
extern "C" int gvt_hip_mesh_download_wide(gvt_hip_mesh *M, void *nodes4, size_t n_nodes4, void *slots, size_t n_slots) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !nodes4 || !slots || n_nodes4 != M->nNodes4 || n_slots != M->nT) {
    set_error("mesh_download_wide: null argument, or (%zu nodes, %zu slots) asked of (%zu, %zu)", n_nodes4, n_slots, M ? M->nNodes4 : (size_t)0, M ? M->nT : (size_t)0);
    return GVT_HIP_ERR_INVALID;
  }
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  if (n_nodes4) HIPCHK(hipMemcpy(nodes4, M->d_nodes4, n_nodes4 * 64, hipMemcpyDeviceToHost));
  if (n_slots) HIPCHK(hipMemcpy(slots, M->d_tri, n_slots * 64, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int gvt_hip_mesh_download_clusters(gvt_hip_mesh *M, void *nodes4c, size_t n_nodes4, int32_t *root_entry) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !nodes4c || !root_entry || n_nodes4 != M->nNodes4) {
    set_error("mesh_download_clusters: null argument, or %zu nodes asked of %zu", n_nodes4, M ? M->nNodes4 : (size_t)0);
    return GVT_HIP_ERR_INVALID;
  }
  if (int rc = build_nodes4c(M)) return rc;
  *root_entry = M->d_nodes4c ? M->root_entry4c : -1;
  if (M->d_nodes4c) HIPCHK(hipMemcpy(nodes4c, M->d_nodes4c, n_nodes4 * 64, hipMemcpyDeviceToHost));
  return 0;
}
// (diagnostic) replaces the binary nodes by a tree of the caller's over the SAME leaves (same node count, root = node 0): only the visit-count
// diagnostics traverse the binary tree; the 4-wide layout the product kernels use is NOT rebuilt from it
extern "C" int gvt_hip_mesh_upload_nodes(gvt_hip_mesh *M, const void *in, size_t n_nodes) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !in || n_nodes != M->nNodes) { set_error("mesh_upload_nodes: null argument or n_nodes != %zu", M ? M->nNodes : (size_t)0); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  if (n_nodes) HIPCHK(hipMemcpy(M->d_nodes, in, n_nodes * sizeof(BvhNode), hipMemcpyHostToDevice));
  return 0;
}
extern "C" int gvt_hip_marked_visit_stats(gvt_hip_mesh *M, const float *org, const float *dir, size_t n, float tnear, const unsigned char *marks, uint32_t *counts) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!M || !marks || (n && (!org || !dir || !counts))) { set_error("marked_visit_stats: null argument"); return GVT_HIP_ERR_INVALID; }
  if (!n || !M->nNodes) { for (size_t i = 0; i < n; i++) counts[i] = 0; return 0; }
  Ctx &C = g_ctx;
  unsigned char *d_marks = nullptr;
  HIPCHK(hipMalloc((void **)&d_marks, M->nNodes));
  int rc = hipMemcpy(d_marks, marks, M->nNodes, hipMemcpyHostToDevice) == hipSuccess ? 0 : GVT_HIP_ERR_DEVICE;
  RayPlanes pl;
  if (!rc) rc = stage_od(org, dir, n, pl);
  unsigned *d_out = rc ? nullptr : (unsigned *)scratch_get(0, n * sizeof(unsigned));
  if (!rc && !d_out) rc = GVT_HIP_ERR_DEVICE;
  if (!rc) rc = launch_wide_visit_stats(M, pl, n, tnear, d_marks, d_out);
  if (!rc && hipMemcpyAsync(counts, d_out, n * sizeof(unsigned), hipMemcpyDeviceToHost, C.stream) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (hipStreamSynchronize(C.stream) != hipSuccess && !rc) rc = GVT_HIP_ERR_DEVICE;
  hipFree(d_marks);
  if (rc == GVT_HIP_ERR_DEVICE && !*gvt_hip_last_error()) set_error("marked_visit_stats: HIP error");
  return rc;
}

int debug_stamps(unsigned long long *out, int reset);
// diagnostic: the launching context's counter words (work counters, parked-ray count [3], overflow flags [8]), after a synchronisation
extern "C" int gvt_hip_counters_peek(uint32_t out[32]) {
  Ctx &C = gctx();
  if (!out || !C.d_counters) { set_error("counters_peek: null"); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipStreamSynchronize(C.stream));
  HIPCHK(hipMemcpy(out, C.d_counters, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int gvt_hip_is_experiments_build(void) {
#ifdef GVT_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
extern "C" int gvt_hip_debug_stamps(unsigned long long *out, int reset) { return debug_stamps(out, reset); }
