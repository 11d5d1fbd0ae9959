// trace.hip -- the adapter hot path on gfx950: Adapter::trace
// (src/gvt/render/Adapter.h:82-84, adapter/embree/EmbreeMeshAdapter.cpp:436-660).
//
// The kernels live in include files by family (all inside this translation unit's anonymous namespace):
//   trace_lane.inc     k_trace: ONE LANE PER RAY, persistent waves with lane refill.  k_trace<false,..> = rtcIntersect (EmbreeMeshAdapter.cpp:474):
//                      ray -> object space, nearest hit in (1e-6, inf); k_trace<true,..> = rtcOccluded (:364-385): un-occluded shadow rays are
//                      appended to moved_rays -- or, with a sink, end right there by the terminal rule of the shuffleRays that would follow
//   trace_wave.inc     A WAVE PER RAY: k_long_closest (the few rays k_trace parks after `long_steps` node steps), k_long_seed / k_wave_any (small rounds)
//   shade.inc          k_shade: the per-lane block :483-609 -- miss forward / shadow drop / normal / material / Shade / shadow-ray generation
//                      (:320-358) / Russian-roulette bounce (:584-602)
//   finish_kernel.inc  k_finish: a small scheduler round in one launch, every ray followed to its end on this rank
//   convert.inc        80-byte Ray AoS <-> planes, ray sort keys, bookkeeping kernels;  diag_kernels.inc: the visit-count diagnostic
// and this file keeps the host side:
//   trace_core         one Adapter::trace call: closest -> shade -> any per pass, one read-back per pass
//   wave_trace_chain   the same chain for a whole scheduler ROUND (all non-empty local queues at once, MULTI kernel variants), ray counts in
//                      device memory, no host round trip;  finish_round: k_finish
//
// k_trace in one paragraph: a wave pulls index ranges from a device counter (an exit every wave reaches), refills lanes whose ray has finished
// while the others go on, alternates a tight loop over the compressed 4-wide nodes with a leaf phase, keeps its traversal stacks in LDS
// (stack[level][lane]: lane-contiguous, bank conflict free) and compacts survivors per wave (__ballot + mbcnt, one atomic per >= 64 rays).
// Compiled only with -DGVT_EXPERIMENTS (libgvt_hip_exp.so, the library the knob sweeps and probes run against): experiments/ -- the
// first-version kernels (trav_kernel=0), the binary-node and quad-cooperative-fetch arms of k_trace (wide4=0, coop_fetch=1), k_fused,
// k_packet, k_traceq (four lanes per ray).  EXPERIMENTS.md lists what was measured on the way.
#include "gvt_internal.h"

#ifndef TRAV_BLOCK
#define TRAV_BLOCK 256   // threads per traversal block (64: every wave is its own block and gives its registers / LDS back when IT ends)
#endif
#define TRAV_WAVES_PER_BLOCK (TRAV_BLOCK / 64) // (the blocks-per-CU knobs count 256-thread blocks = waves per SIMD)
#ifndef TRAV_STACK
#define TRAV_STACK 24   // LDS entries per lane; deeper levels spill to a per-thread global area
#endif
#ifndef KT_BLOCKS_CLOSEST
#define KT_BLOCKS_CLOSEST 5 // resident 256-thread blocks per CU the compiler must leave room for (register budget 512 / waves per SIMD)
#endif
#ifndef RETIRE_BATCH
#define RETIRE_BATCH 1
#endif
#ifndef KT_BLOCKS_ANY
#define KT_BLOCKS_ANY 4
#endif
#define TRAV_SPILL 128  // 24 + 128 = 152 pending entries: > 63 + 24 levels of a 63-bit Karras tree with index tie-breaks, and > 3 x 44, the
                        // worst case of its 4-wide collapse; k_trace reports an error beyond that instead of losing entries

namespace {

struct Trav {
  const BvhNode *__restrict__ nodes;
  const float4 *__restrict__ tris;
  const uint4 *__restrict__ nodes4;  // compressed 4-wide collapse (64 B per node) or null
};

#include "diag_kernels.inc" // visit counts (+ the first-version kernels in the experiments build)

#include "trace_lane.inc" // the lane-per-ray traversal

#ifdef GVT_EXPERIMENTS
#include "experiments/quad_kernel.inc" // k_traceq: four lanes per ray (measured slower: VALU bound, EXPERIMENTS.md)
#endif

#include "trace_wave.inc" // the wave-per-ray traversal

#include "packet_kernel.inc" // k_packet: a wave walks the BVH for a packet of 64 coherent rays (meshes the builder found packet-friendly)

#include "shade.inc" // Shade(), lights, the bounce direction, k_shade

#include "finish_kernel.inc" // k_finish: a small round in one launch, every ray followed to its end on this rank

#ifdef GVT_EXPERIMENTS
#include "experiments/fused_lean.inc" // k_frame1: closest hit + Lambert / point-light shade + shadow ray + deposit of a one-instance depth-1 frame in ONE launch (measured slower, EXPERIMENTS.md; knob `fused1`)
#endif

#ifdef GVT_EXPERIMENTS
#include "experiments/fused_kernel.inc" // k_fused: the one-kernel closest + shade + shadow variant (measured slower, EXPERIMENTS.md; knob `fused`)
#endif

#include "convert.inc" // layout conversions at the ABI boundary, ray sort keys, the small bookkeeping kernels of a trace call

inline unsigned blocks_for(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }

int trav_grid(size_t n) {
  Ctx &C = gctx();
  size_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
  return (int)(need < (size_t)C.trav_blocks ? (need ? need : 1) : (size_t)C.trav_blocks);
}

template <bool ANY, bool XFORM, int MODE, typename... Args> void launch_trace(bool have_nodes4, int grid, hipStream_t st, RayPlanes q, const unsigned *idx, unsigned n, Mat4 minv, Args... args) {
  // k_trace never reads the matrix's bottom row (an instance matrix is affine); it carries the direction transform's `translation x 0` terms there instead
  // (xfm_vector: m[12..14] * 0.0f, kept for parity with glm's mat4 * vec4(d, 0)): kernel arguments live in SCALAR registers, where the products computed on
  // the device sat in three vector registers for the whole launch -- spilled to scratch in the shipped closest-hit kernel
  minv.m[3] = minv.m[12] * 0.0f; minv.m[7] = minv.m[13] * 0.0f; minv.m[11] = minv.m[14] * 0.0f;
#ifdef GVT_EXPERIMENTS
  if (!(gctx().wide4 && have_nodes4)) { // the uncompressed binary nodes
    if (gctx().coop_fetch) k_trace<ANY, XFORM, MODE, true, false><<<grid, TRAV_BLOCK, 0, st>>>(q, idx, n, minv, args...);
    else k_trace<ANY, XFORM, MODE, false, false><<<grid, TRAV_BLOCK, 0, st>>>(q, idx, n, minv, args...);
    return;
  }
#endif
  k_trace<ANY, XFORM, MODE, false, true><<<grid, TRAV_BLOCK, 0, st>>>(q, idx, n, minv, args...); // a mesh without 4-wide nodes has no triangles: every ray retires as a miss
}

// scratch of a closest-hit launch's parked rays: n records, then LONG_SAVE stack entries for each of the first LONG_STK_CAP of them
static size_t long_scratch_bytes(size_t n) { return sizeof(LongRec) * n + sizeof(int) * LONG_SAVE * (size_t)LONG_STK_CAP; }
static void long_limits(LongQ &LQ, size_t n) {
  Ctx &C = gctx();
  LQ.steps = C.long_steps_override > 0 ? C.long_steps_override : C.long_steps;
  LQ.steps_drain = C.long_steps_drain > 0 && C.long_steps_drain < LQ.steps ? C.long_steps_drain : LQ.steps;
  LQ.stk = C.long_save ? (int *)(LQ.recs + n) : nullptr;
  LQ.stk_cap = LONG_STK_CAP;
}

// persistent-wave kernel: fewer, longer-lived waves so that every lane is refilled several times
int trav_grid2(size_t n, bool closest = false) {
  Ctx &C = gctx();
  size_t want = (size_t)C.n_cu * (((size_t)((closest && C.blocks_per_cu_closest) ? C.blocks_per_cu_closest : C.blocks_per_cu) * 4) / TRAV_WAVES_PER_BLOCK);
  if (!want) want = C.n_cu;
  size_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
  if (want > (size_t)C.trav_blocks) want = (size_t)C.trav_blocks;
  return (int)(need < want ? (need ? need : 1) : want);
}

#ifdef GVT_EXPERIMENTS
// k_traceq: 64 rays in flight per 256-thread block
int quad_grid(size_t n) {
  Ctx &C = gctx();
  size_t want = (size_t)C.n_cu * (size_t)C.blocks_per_cu_quad;
  size_t need = (n + 63) / 64;
  if (want > (size_t)C.trav_blocks) want = (size_t)C.trav_blocks;
  return (int)(need < want ? (need ? need : 1) : want);
}
// (32-bit byte offsets into the node and leaf-block arrays: meshes beyond 2^26 triangles keep the one-lane-per-ray kernels)
inline bool quad_usable(const gvt_hip_mesh *M) { return gctx().quad && gctx().trav_kernel == 1 && M->d_nodes4q && M->d_triq && M->nT < ((size_t)1 << 26) && M->nNodes4 < ((size_t)1 << 26); }
#endif

} // namespace

int debug_stamps(unsigned long long *out, int reset) {
#if GVT_STAMP
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 24);
  if (reset) { unsigned long long z[24] = { 0 }; hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof z); }
  return 0;
#else
  for (int i = 0; i < 24; i++) out[i] = 0;
  return -1;
#endif
}
// queued behind the traversal launches on the stream; read after the caller's next synchronisation through trav_overflow_result()
int trav_overflow_fetch_async() {
  Ctx &C = gctx();
  HIPCHK(hipMemcpyAsync(C.h_pinned + 8, C.d_counters + TRAV_OVF_WORD, sizeof(unsigned), hipMemcpyDeviceToHost, C.stream));
  return 0;
}
int trav_overflow_result() {
  Ctx &C = gctx();
  if (!C.h_pinned[8]) return 0;
  C.h_pinned[8] = 0;
  hipMemsetAsync(C.d_counters + TRAV_OVF_WORD, 0, sizeof(unsigned), C.stream);
  set_error("BVH traversal stack overflow (more than %d pending entries for one ray): results of this call are incomplete", TRAV_STACK + TRAV_SPILL);
  return GVT_HIP_ERR_DEVICE;
}
size_t trav_spill_ints_per_thread() { return TRAV_SPILL; }
int trav_block_threads() { return TRAV_BLOCK; }

int convert_aos_to_planes(const gvt_hip_ray *d_src, size_t n, RayPlanes dst, size_t dst_off, bool keep_state) {
  if (!n) return 0;
  ProfScope ps(KC_CONVERT);
  k_aos_to_planes<<<blocks_for(n), 256, 0, gctx().stream>>>((const float4 *)d_src, (unsigned)n, dst, dst_off, keep_state ? 1 : 0);
  HIPCHK(hipGetLastError());
  return 0;
}
int convert_planes_to_aos(RayPlanes src, size_t src_off, size_t n, gvt_hip_ray *d_dst) {
  if (!n) return 0;
  ProfScope ps(KC_CONVERT);
  k_planes_to_aos<<<blocks_for(n), 256, 0, gctx().stream>>>(src, src_off, (unsigned)n, (float4 *)d_dst);
  HIPCHK(hipGetLastError());
  return 0;
}
int convert_od_to_planes(const float *d_org, const float *d_dir, size_t n, RayPlanes dst) {
  if (!n) return 0;
  ProfScope ps(KC_CONVERT);
  k_od_to_planes<<<blocks_for(n), 256, 0, gctx().stream>>>(d_org, d_dir, (unsigned)n, dst);
  HIPCHK(hipGetLastError());
  return 0;
}

int launch_closest(gvt_hip_mesh *M, RayPlanes q, const unsigned *idx, size_t n, bool xform, const Mat4 &minv, float tnear,
                   gvt_hip_hit *d_hits, bool counter_is_zero) {
  if (!n) return 0;
  Ctx &C = gctx();
  if (gctx().wide4 && !M->d_nodes4) { int rc4 = build_nodes4(M); if (rc4) return rc4; }
  const bool have_nodes4 = M->d_nodes4 != nullptr;
  Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
  unsigned *counter = C.d_counters + 0;
  if (!counter_is_zero) HIPCHK(hipMemsetAsync(C.d_counters, 0, 5 * sizeof(unsigned), C.stream));
  LongQ LQ{};
  if (C.trav_kernel == 1 && C.wide4 && have_nodes4 && C.long_steps > 0 && n >= (size_t)C.long_min_rays) { // long rays are parked and traversed a wave per ray
    LQ.recs = (LongRec *)scratch_get(15, long_scratch_bytes(n));
    if (!LQ.recs) return GVT_HIP_ERR_DEVICE;
    LQ.count = C.d_counters + 3;
    long_limits(LQ, n);
  }
  {
    ProfScope ps(KC_CLOSEST);
    RayPlanes none{};
#ifdef GVT_EXPERIMENTS
    if (quad_usable(M)) {
      TravQ TQ{ M->d_nodes4q, M->d_triq };
      if (xform) k_traceq<false, true, 0><<<quad_grid(n), 256, 0, C.stream>>>(q, idx, (unsigned)n, minv, TQ, tnear, d_hits, nullptr, none, nullptr, counter, C.d_spill, C.quad_refill_min, C.quad_inner_min, nullptr, TermSink{}, LQ);
      else k_traceq<false, false, 0><<<quad_grid(n), 256, 0, C.stream>>>(q, idx, (unsigned)n, minv, TQ, tnear, d_hits, nullptr, none, nullptr, counter, C.d_spill, C.quad_refill_min, C.quad_inner_min, nullptr, TermSink{}, LQ);
    } else if (C.trav_kernel != 1) {
      if (xform) k_closest<true><<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, idx, (unsigned)n, minv, T, tnear, d_hits, counter, C.d_spill);
      else k_closest<false><<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, idx, (unsigned)n, minv, T, tnear, d_hits, counter, C.d_spill);
    } else
#endif
    {
      if (xform) launch_trace<false, true, 0>(have_nodes4, trav_grid2(n, true), C.stream, q, idx, (unsigned)n, minv, T, tnear, d_hits, nullptr, none, nullptr, counter, C.d_spill, C.refill_min, C.inner_min, nullptr, C.share, (unsigned)C.share_min_rays, TermSink{}, LQ);
      else launch_trace<false, false, 0>(have_nodes4, trav_grid2(n, true), C.stream, q, idx, (unsigned)n, minv, T, tnear, d_hits, nullptr, none, nullptr, counter, C.d_spill, C.refill_min, C.inner_min, nullptr, C.share, (unsigned)C.share_min_rays, TermSink{}, LQ);
    }
  }
  if (LQ.steps) {
    {
      ProfScope ps(KC_LONG);
      {
        const int grid = C.n_cu * 3; // 48 KiB of LDS per block
        if (xform) k_long_closest<true><<<grid, 256, 0, C.stream>>>(q, LQ.recs, LQ.count, minv, T, tnear, d_hits, C.d_counters + 4, WaveSet{}, LQ.stk);
        else k_long_closest<false><<<grid, 256, 0, C.stream>>>(q, LQ.recs, LQ.count, minv, T, tnear, d_hits, C.d_counters + 4, WaveSet{}, LQ.stk);
      }
    }
  }
  HIPCHK(hipGetLastError());
  C.stats.rays_closest += n;
  C.stats.launches_closest++;
  return 0;
}

int launch_visit_stats(gvt_hip_mesh *M, RayPlanes q, size_t n, float tnear, unsigned *d_out) {
  if (!n) return 0;
  Ctx &C = gctx();
  if (gctx().wide4 && !M->d_nodes4) { int rc4 = build_nodes4(M); if (rc4) return rc4; }
  Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
  unsigned *counter = C.d_counters + 0;
  HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned), C.stream));
  k_visit_stats<<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, (unsigned)n, T, tnear, d_out, counter, C.d_spill);
  HIPCHK(hipGetLastError());
  return 0;
}

int launch_wide_visit_stats(gvt_hip_mesh *M, RayPlanes q, size_t n, float tnear, const unsigned char *d_marks, unsigned *d_out) {
  if (!n) return 0;
  Ctx &C = gctx();
  Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
  unsigned *counter = C.d_counters + 0;
  HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned), C.stream));
  k_wide_visit_stats<<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, (unsigned)n, T, tnear, d_marks, d_out, counter, C.d_spill);
  HIPCHK(hipGetLastError());
  return 0;
}

int launch_any_flags(gvt_hip_mesh *M, RayPlanes q, size_t n, bool xform, const Mat4 &minv, float tnear, int *d_flags) {
  if (!n) return 0;
  Ctx &C = gctx();
  if (gctx().wide4 && !M->d_nodes4) { int rc4 = build_nodes4(M); if (rc4) return rc4; }
  const bool have_nodes4 = M->d_nodes4 != nullptr;
  Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
  unsigned *counter = C.d_counters + 0;
  HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned), C.stream));
  RayPlanes none{};
  {
    ProfScope ps(KC_ANY);
#ifdef GVT_EXPERIMENTS
    if (quad_usable(M)) {
      TravQ TQ{ M->d_nodes4q, M->d_triq };
      if (xform) k_traceq<true, true, 0><<<quad_grid(n), 256, 0, C.stream>>>(q, nullptr, (unsigned)n, minv, TQ, tnear, nullptr, d_flags, none, nullptr, counter, C.d_spill, C.quad_refill_min, C.quad_inner_min, nullptr, TermSink{}, LongQ{});
      else k_traceq<true, false, 0><<<quad_grid(n), 256, 0, C.stream>>>(q, nullptr, (unsigned)n, minv, TQ, tnear, nullptr, d_flags, none, nullptr, counter, C.d_spill, C.quad_refill_min, C.quad_inner_min, nullptr, TermSink{}, LongQ{});
    } else if (C.trav_kernel != 1) {
      if (xform) k_any<true, 0><<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, (unsigned)n, minv, T, tnear, d_flags, none, nullptr, counter, C.d_spill);
      else k_any<false, 0><<<trav_grid(n), TRAV_BLOCK, 0, C.stream>>>(q, (unsigned)n, minv, T, tnear, d_flags, none, nullptr, counter, C.d_spill);
    } else
#endif
    {
      if (xform) launch_trace<true, true, 0>(have_nodes4, trav_grid2(n), C.stream, q, nullptr, (unsigned)n, minv, T, tnear, nullptr, d_flags, none, nullptr, counter, C.d_spill, C.refill_min, C.inner_min, nullptr, C.share, (unsigned)C.share_min_rays, TermSink{}, LongQ{});
      else launch_trace<true, false, 0>(have_nodes4, trav_grid2(n), C.stream, q, nullptr, (unsigned)n, minv, T, tnear, nullptr, d_flags, none, nullptr, counter, C.d_spill, C.refill_min, C.inner_min, nullptr, C.share, (unsigned)C.share_min_rays, TermSink{}, LongQ{});
    }
  }
  HIPCHK(hipGetLastError());
  C.stats.rays_any += n;
  C.stats.launches_any++;
  return 0;
}

// Adapter::trace on device planes.  `in` holds n rays at [0,n); `out` must have been reserved for
// out->size + n*(1+n_lights) -- what the first pass can emit; further passes (bounces) grow it themselves.  On return out->size is exact (one small read-back).
int trace_core(gvt_hip_mesh *M, RayPlanes in, size_t n, uint64_t index_base, gvt_hip_queue *out, const TraceParams &P,
               const gvt_hip_light *lights_host) {
  Ctx &C = gctx();
  C.stats.trace_calls++;
  if (!n) return 0;
  hipStream_t st = C.stream;
  const int nL = P.n_lights;
  // scratch: hits, rng, shadow queue, two index lists, lights
  gvt_hip_hit *d_hits = (gvt_hip_hit *)scratch_get(0, sizeof(gvt_hip_hit) * n);
  const size_t shadow_cap = n * (size_t)(nL > 0 ? nL : 1);
  float4 *d_shadow = (float4 *)scratch_get(2, sizeof(float4) * 4 * shadow_cap);
  unsigned *d_idx_a = (unsigned *)scratch_get(3, sizeof(unsigned) * n);
  unsigned *d_idx_b = (unsigned *)scratch_get(4, sizeof(unsigned) * n);
  gvt_hip_light *d_lights = (gvt_hip_light *)scratch_get(5, sizeof(gvt_hip_light) * (nL > 0 ? nL : 1));
  if (!d_hits || !d_shadow || !d_idx_a || !d_idx_b || !d_lights) return GVT_HIP_ERR_DEVICE;
  { // the light list rarely changes between calls: upload only when it (or its scratch buffer) did
    std::vector<unsigned char> &cached = C.lights_cached;
    const void *&cached_dst = C.lights_cached_dst;
    const size_t bytes = sizeof(gvt_hip_light) * (size_t)nL;
    if (nL && (cached_dst != d_lights || cached.size() != bytes || std::memcmp(cached.data(), lights_host, bytes) != 0)) {
      cached.assign((const unsigned char *)lights_host, (const unsigned char *)lights_host + bytes);
      cached_dst = d_lights;
      HIPCHK(hipMemcpyAsync(d_lights, cached.data(), bytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st)); // pageable source: finish the copy before `cached` can change
    }
  }
  RayPlanes shadow = make_planes(d_shadow, shadow_cap);
  shadow.p4 = nullptr; shadow.p5 = nullptr; // shadow rays carry no stream and no list
  RayPlanes outp = make_planes(out->d_planes, out->cap);
  unsigned *c_shadow = C.d_counters + 1, *c_next = C.d_counters + 2;
  k_trace_begin<<<1, 64, 0, st>>>(out->d_count, (unsigned)out->size, C.d_counters); // c_shadow, c_next, work counter

  MeshView mv;
  mv.slots = M->d_tri; mv.slot_of = M->d_slot_of; mv.verts = M->d_verts; mv.tris = M->d_tris; mv.normals = M->d_normals; mv.vcolors = M->d_vcolors;
  mv.materials = M->d_materials; mv.n_mat = (unsigned)M->nMat; mv.face_mat = M->d_face_mat;
  mv.mat = M->mesh_mat;

  size_t n_active = n;
  const size_t out_size0 = out->size;
  const unsigned *idx = nullptr;
  unsigned *next = d_idx_a;
  int pass = 0;
#ifdef GVT_EXPERIMENTS
  const bool persistent = C.trav_kernel == 1;
#else
  const bool persistent = true;
#endif
  while (n_active) { // while (validRayLeft) :465
    int rc;
    if (C.sort_rays && n_active >= 8192) { // ray sorting: traverse in Morton order of the object-space origin
      unsigned *k_in = (unsigned *)scratch_get(8, sizeof(unsigned) * n), *k_out = (unsigned *)scratch_get(9, sizeof(unsigned) * n);
      unsigned *v_in = (unsigned *)scratch_get(10, sizeof(unsigned) * n), *v_out = (unsigned *)scratch_get(11, sizeof(unsigned) * n);
      if (!k_in || !k_out || !v_in || !v_out) return GVT_HIP_ERR_DEVICE;
      const float ex = M->hi[0] - M->lo[0], ey = M->hi[1] - M->lo[1], ez = M->hi[2] - M->lo[2];
      const float3 blo = make_float3(M->lo[0], M->lo[1], M->lo[2]);
      const float3 inv = make_float3(ex > 0 ? 1.f / ex : 0.f, ey > 0 ? 1.f / ey : 0.f, ez > 0 ? 1.f / ez : 0.f);
      {
        ProfScope ps(KC_SORT);
        k_ray_keys<<<blocks_for(n_active), 256, 0, st>>>(in, idx, (unsigned)n_active, P.minv, blo, inv, k_in, v_in, C.sort_bits);
        if ((rc = sort_pairs_u32(k_in, k_out, v_in, v_out, n_active, C.sort_bits))) return rc;
      }
      HIPCHK(hipGetLastError());
      idx = v_out;
    }
    if (idx && C.sort_rays && C.sort_gather && n_active >= 8192) {
      float4 *od = (float4 *)scratch_get(13, sizeof(float4) * 2 * n);
      if (!od) return GVT_HIP_ERR_DEVICE;
      {
        ProfScope ps(KC_SORT);
        k_gather_od<<<blocks_for(n_active), 256, 0, st>>>(in, idx, (unsigned)n_active, P.minv, od, od + n);
      }
      HIPCHK(hipGetLastError());
      RayPlanes sorted{ od, od + n, nullptr, nullptr, nullptr };
      rc = launch_closest(M, sorted, nullptr, n_active, false, P.minv, GVT_RAY_EPSILON, d_hits, pass == 0);
    } else
      rc = launch_closest(M, in, idx, n_active, true, P.minv, GVT_RAY_EPSILON, d_hits, pass == 0);
    if (rc) return rc;
    if (pass > 0) HIPCHK(hipMemsetAsync(c_shadow, 0, 2 * sizeof(unsigned), st)); // c_shadow, c_next adjacent (pass 0: k_trace_begin)
    ShadeArgs A{};
    A.in = in; A.idx = idx; A.n = (unsigned)n_active; A.index_base = index_base; A.hits = d_hits;
    A.first_pass = (pass == 0); A.carried_rng = P.carried_rng; A.out = outp; A.out_count = out->d_count; A.shadow = shadow; A.shadow_count = c_shadow;
    A.next_idx = next; A.next_count = c_next; A.lights = d_lights; A.normi = P.normi; A.normal_mode = P.normal_mode;
    A.n_lights = nL; A.seed = P.seed; A.zero_word = persistent ? C.d_counters + 0 : nullptr;
    A.sink = P.sink; A.update_in_place = P.update_in_place;
    A.n_dev = nullptr; A.W = WaveSet{}; A.out_from = nullptr; A.shadow_inst = nullptr; A.shadow_stride = 0;
    {
      ProfScope ps(KC_SHADE);
      k_shade<false><<<blocks_for(n_active, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, mv);
    }
    HIPCHK(hipGetLastError());
    C.stats.rays_shaded += n_active;
    // traceShadowRays :364-385.  With the persistent kernel the shadow-ray count stays on the device (the kernel reads
    // it, the grid does not depend on it); the host learns the counts of the pass in ONE read-back afterwards.
    unsigned n_shadow = 0, n_next = 0;
    const size_t shadow_ub = n_active * (size_t)nL;
    if (persistent) {
      if (shadow_ub) {
        Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
        unsigned *counter = C.d_counters + 0; // zeroed by k_shade
        {
          ProfScope ps(KC_ANY);
#ifdef GVT_EXPERIMENTS
          if (quad_usable(M))
            k_traceq<true, true, 1><<<quad_grid(shadow_ub), 256, 0, st>>>(shadow, nullptr, 0u, P.minv, TravQ{ M->d_nodes4q, M->d_triq }, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count, counter,
                                                                          C.d_spill, C.quad_refill_min, C.quad_inner_min, c_shadow, P.sink, LongQ{});
          else
#endif
          launch_trace<true, true, 1>(M->d_nodes4 != nullptr, trav_grid2(shadow_ub), st, shadow, nullptr, 0u, P.minv, T, GVT_RAY_EPSILON, nullptr, nullptr, outp,
                                                                               out->d_count, counter, C.d_spill, C.refill_min, C.inner_min, c_shadow, C.share, (unsigned)C.share_min_rays, P.sink, LongQ{});
        }
        HIPCHK(hipGetLastError());
        C.stats.launches_any++;
      }
      HIPCHK(hipMemcpyAsync(C.h_pinned, c_shadow, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(C.h_pinned + 2, out->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      if ((rc = trav_overflow_fetch_async())) return rc;
      HIPCHK(hipStreamSynchronize(st));
      if ((rc = trav_overflow_result())) return rc;
      n_shadow = C.h_pinned[0]; n_next = C.h_pinned[1];
      C.stats.rays_any += n_shadow;
    }
#ifdef GVT_EXPERIMENTS
    else { // trav_kernel=0: the first-version kernels size their grid from the host-known count
      HIPCHK(hipMemcpyAsync(C.h_pinned, c_shadow, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      n_shadow = C.h_pinned[0]; n_next = C.h_pinned[1];
      if (n_shadow) {
        Trav T{ M->d_nodes, M->d_tri, M->d_nodes4 };
        unsigned *counter = C.d_counters + 0;
        HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
        {
          ProfScope ps(KC_ANY);
          k_any<true, 1><<<trav_grid(n_shadow), TRAV_BLOCK, 0, st>>>(shadow, n_shadow, P.minv, T, GVT_RAY_EPSILON, nullptr, outp, out->d_count, counter,
                                                                      C.d_spill);
        }
        HIPCHK(hipGetLastError());
        C.stats.rays_any += n_shadow;
        C.stats.launches_any++;
      }
      HIPCHK(hipMemcpyAsync(C.h_pinned + 2, out->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
#endif
    n_active = n_next;
    idx = next;
    next = (next == d_idx_a) ? d_idx_b : d_idx_a;
    pass++;
    if (n_active) { // a further pass (bounces): room for everything IT can emit -- a forward or nL shadow rays per ray -- behind what is there
      out->size = C.h_pinned[2];
      if ((rc = queue_reserve(out, out->size + n_active * (size_t)(1 + nL)))) return rc;
      outp = make_planes(out->d_planes, out->cap);
    }
  }
  C.stats.rays_forwarded += C.h_pinned[2] - out_size0; // read back with the last pass
  out->size = C.h_pinned[2];
  return 0;
}

extern "C" int gvt_hip_math_probe(int kind, const float *in, size_t n, float *out) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (kind < 0 || kind > 2 || (n && (!in || !out)) || n > 0xffffffffull) { set_error("math_probe: bad argument"); return GVT_HIP_ERR_INVALID; }
  if (!n) return 0;
  Ctx &C = gctx();
  float *d = (float *)scratch_get(0, sizeof(float) * 2 * n);
  if (!d) return GVT_HIP_ERR_DEVICE;
  HIPCHK(hipMemcpyAsync(d, in, sizeof(float) * n, hipMemcpyHostToDevice, C.stream));
  k_math_probe<<<blocks_for(n), 256, 0, C.stream>>>(kind, d, (unsigned)n, d + n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d + n, sizeof(float) * n, hipMemcpyDeviceToHost, C.stream));
  HIPCHK(hipStreamSynchronize(C.stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Merged launch chain of one scheduler round (gvt_internal.h "wave"): closest hit -> shade -> any hit over the rays of all local
// queues at once, `passes` times (bounces), WITHOUT any host round trip: later passes read their ray count from device memory.
// Device counters of the context (d_counters): [0] work counter, [1] shadow rays of the pass, [2] / [5] bounce list counts
// (alternating), [3] parked long rays, [4] their work counter, [8] stack overflow flag, [16..19] two 64-bit ray totals
// (closest, any) accumulated over the frame and read back with the queue sizes.
// ------------------------------------------------------------------------------------------------
namespace {
// segs (pass 0 of a merged chain whose queues were filled by the camera filter without a read-back; the host uploaded the segments with
// the BOUNDS it knows): each segment's length is its queue's count word, its beginning the sum of the lengths before it, the list's
// length goes to *n_dev0 (a handful of segments: one thread)
__global__ void k_wave_pass_begin(unsigned *c, int pass, unsigned n_host, unsigned *out_count, const unsigned *n_dev0 = nullptr, WaveSeg *segs = nullptr, int n_seg = 0,
                                  unsigned *const *__restrict__ count_ptr = nullptr) {
  if (blockIdx.x || threadIdx.x) return;
  if (segs) {
    unsigned run = 0;
    for (int k = 0; k < n_seg; k++) { const unsigned n = *count_ptr[segs[k].inst]; segs[k].begin = run; segs[k].n = n; run += n; }
    *const_cast<unsigned *>(n_dev0) = run;
  }
  unsigned long long *tot = (unsigned long long *)(c + 16);
  const int cur = (pass & 1) ? 5 : 2, prev = (pass & 1) ? 2 : 5;
  if (pass == 0) { *out_count = 0u; c[2] = 0u; c[5] = 0u; tot[0] += n_dev0 ? *n_dev0 : n_host; }
  else { tot[1] += c[1]; tot[0] += c[prev]; c[cur] = 0u; }
  c[20] += c[3]; // rays parked by the launch before (the frame's total: the tracer adapts its parking threshold to it)
  c[0] = 0u; c[1] = 0u; c[3] = 0u; c[4] = 0u; c[6] = 0u;
  for (int k = 0; k < SHADOW_CLASSES; k++) c[SHADOW_CLS_WORD + k] = 0u; // the shadow list's class counts (shade.inc)
}
// end of a round's chain: the last pass's shadow rays into the frame total; and the traced queues' clear() (count words of the
// queues whose mask byte is set) in the same launch
__global__ void k_wave_end(unsigned *c, unsigned *const *__restrict__ count_ptr, const unsigned char *__restrict__ mask, int n_inst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { unsigned long long *tot = (unsigned long long *)(c + 16); tot[1] += c[1]; }
  if (count_ptr && i < n_inst && mask[i]) *count_ptr[i] = 0u;
}
} // namespace

int wave_trace_chain(const WaveSet &W, size_t n_total, int passes, gvt_hip_queue *out, int *d_out_from, const TraceParams &P,
                     const gvt_hip_light *lights_host, const WaveSingle *single, unsigned *const *d_count_ptr, const unsigned char *d_mask, int n_inst,
                     bool defer_end, const unsigned *n_dev0_multi, bool multi_packets, bool simple_meshes) {
  Ctx &C = gctx();
  if (!n_total) return 0;
  hipStream_t st = C.stream;
  const int nL = P.n_lights;
  const size_t n = n_total;
  gvt_hip_hit *d_hits = (gvt_hip_hit *)scratch_get(0, sizeof(gvt_hip_hit) * n);
  const size_t shadow_cap = n * (size_t)(nL > 0 ? nL : 1);
  // a single-mesh round with one light: the shadow rays listed by how long their primaries' tiles took, the longest first (knob shadow_order; shade.inc) --
  // SHADOW_CLASSES regions of `cls_stride` slots; not for packets (no per-ray step counts) nor small rounds (a wave per ray)
  const bool by_class_wanted = single && single->mesh->d_nodes4 && C.shadow_order && nL == 1 && n >= (size_t)C.shadow_order_min_rays && n > (size_t)C.small_rays &&
                        C.long_steps > 0 && n >= (size_t)C.long_min_rays &&
                        !((C.packet == 2 || (C.packet == 1 && single->mesh->packet_ok && n >= (size_t)C.packet_min_rays)) && single->coherent)
#ifdef GVT_EXPERIMENTS
                        && !quad_usable(single->mesh) // (k_traceq walks the shadow list as one contiguous run counted by c[1])
#endif
      ;
  const size_t cls_stride = (n + 63) & ~(size_t)63;
  // (every region has room for the whole list -- 8 x 64 bytes per ray: 0.5 GB at the benchmark's 1.04 M rays; beyond 8 GiB the list stays in arrival order)
  // The ordering is an optimisation whose results do not depend on it: it must never make a frame fail that fits without it.  A device that cannot
  // give the 8x list falls back to arrival order with the plain n-slot list, for this context from then on (no allocation attempt per frame).
  bool by_class = by_class_wanted && !C.shadow_order_denied && cls_stride * SHADOW_CLASSES * 64 <= ((size_t)8 << 30);
  size_t shadow_slots = by_class ? cls_stride * SHADOW_CLASSES : shadow_cap;
  if (shadow_slots >= 0xffffffffull) { set_error("round: %zu shadow slots exceed the 32-bit slot counters", shadow_slots); return GVT_HIP_ERR_INVALID; }
  float4 *d_shadow = (float4 *)scratch_get(2, sizeof(float4) * 4 * shadow_slots);
  unsigned char *d_steps = (by_class && d_shadow) ? (unsigned char *)scratch_get(19, n) : nullptr;
  if (by_class && (!d_shadow || !d_steps)) {
    (void)hipGetLastError(); // (the failed hipMalloc's sticky-less error)
    C.shadow_order_denied = true;
    by_class = false;
    d_steps = nullptr;
    shadow_slots = shadow_cap;
    d_shadow = (float4 *)scratch_get(2, sizeof(float4) * 4 * shadow_slots);
  }
  unsigned *d_idx_a = (unsigned *)scratch_get(3, sizeof(unsigned) * n);
  unsigned *d_idx_b = (unsigned *)scratch_get(4, sizeof(unsigned) * n);
  gvt_hip_light *d_lights = (gvt_hip_light *)scratch_get(5, sizeof(gvt_hip_light) * (nL > 0 ? nL : 1));
  int *d_shadow_inst = (int *)scratch_get(16, sizeof(int) * shadow_cap);
  LongRec *d_long = (LongRec *)scratch_get(15, long_scratch_bytes(n));
  if (!d_hits || !d_shadow || !d_idx_a || !d_idx_b || !d_lights || !d_shadow_inst || !d_long) return GVT_HIP_ERR_DEVICE;
  {
    std::vector<unsigned char> &cached = C.lights_cached;
    const void *&cached_dst = C.lights_cached_dst;
    const size_t bytes = sizeof(gvt_hip_light) * (size_t)nL;
    if (nL && (cached_dst != d_lights || cached.size() != bytes || std::memcmp(cached.data(), lights_host, bytes) != 0)) {
      cached.assign((const unsigned char *)lights_host, (const unsigned char *)lights_host + bytes);
      cached_dst = d_lights;
      HIPCHK(hipMemcpyAsync(d_lights, cached.data(), bytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st)); // pageable source: finish the copy before `cached` can change (first frame only)
    }
  }
  // hops (MultiSrc, trace_lane.inc): in a merged chain a ray that leaves its instance without a hit goes on into the next instance of this rank inside the launch;
  // d_hop[virtual index] = the instance it is in now (-1: its segment's), for the kernels behind (k_long_closest, k_shade, the next pass's refill)
  int *d_hop = nullptr;
  if (!single && P.hop && P.sink.fb) {
    d_hop = (int *)scratch_get(23, sizeof(int) * n);
    if (!d_hop) return GVT_HIP_ERR_DEVICE;
    HIPCHK(hipMemsetAsync(d_hop, 0xff, sizeof(int) * n, st));
  }
  unsigned long long *hop_tot = (unsigned long long *)(C.d_counters + 16); // (the chain's closest-hit total: k_wave_pass_begin)
  RayPlanes shadow = make_planes(d_shadow, shadow_slots);
  shadow.p4 = nullptr; shadow.p5 = nullptr;
  RayPlanes outp = make_planes(out->d_planes, out->cap);
  RayPlanes none{};
  unsigned *c = C.d_counters;
  Trav T{};
  Mat4 id{};
  const bool use_long = C.long_steps > 0 && n >= (size_t)C.long_min_rays;
  const bool small = n <= (size_t)C.small_rays; // a wave per ray (see k_long_seed)
  const int small_grid = (int)std::min<size_t>((n + 3) / 4, (size_t)C.n_cu * 3);
  // pass 0's ray count where only the device knows it (the chain straight behind the camera filter): the single queue's count word, or
  // the word the caller's segment kernel left the merged list's length in (n_total is then the bound the launches are sized by)
  const unsigned *nd0 = single ? single->n_dev : n_dev0_multi;
  for (int pass = 0; pass < passes; pass++) {
    const unsigned *n_dev = pass ? c + ((pass & 1) ? 2 : 5) : nd0; // count written by the previous pass's k_shade
    const unsigned *idx = pass ? ((pass & 1) ? d_idx_a : d_idx_b) : nullptr;    // its bounce list
    unsigned *next = (pass & 1) ? d_idx_b : d_idx_a;
    unsigned *c_next = c + ((pass & 1) ? 5 : 2);
    if (!(pass == 0 && single && single->pass0_begun))
      k_wave_pass_begin<<<1, 64, 0, st>>>(c, pass, (unsigned)n, out->d_count, nd0, (pass == 0 && n_dev0_multi && !single) ? const_cast<WaveSeg *>(W.segs) : nullptr, W.n_seg, d_count_ptr);
    // ray sorting where it should pay (knob sort_rays, off by default: measured, profiles/r06_sort_cfg5.txt): a BOUNCE list -- secondary rays in their parents'
    // order, i.e. neighbours in the list leave neighbouring surface points in unrelated directions -- is reordered by direction octant, then the Morton cell of the
    // ray's origin (k_ray_keys) before the closest-hit launch; the pass's shadow rays inherit the order (k_shade allocates their slots in list order).  The list's
    // length lives on the device: all n entries of the bound are sorted, the unused ones with the largest key
    if (C.sort_rays && pass > 0 && single && n >= 8192 && single->mesh->d_nodes4) {
      gvt_hip_mesh *Ms = single->mesh;
      unsigned *k_in = (unsigned *)scratch_get(8, sizeof(unsigned) * n), *k_out = (unsigned *)scratch_get(9, sizeof(unsigned) * n);
      unsigned *v_in = (unsigned *)scratch_get(10, sizeof(unsigned) * n), *v_out = (unsigned *)scratch_get(11, sizeof(unsigned) * n);
      if (!k_in || !k_out || !v_in || !v_out) return GVT_HIP_ERR_DEVICE;
      const float ex = Ms->hi[0] - Ms->lo[0], ey = Ms->hi[1] - Ms->lo[1], ez = Ms->hi[2] - Ms->lo[2];
      const float3 blo = make_float3(Ms->lo[0], Ms->lo[1], Ms->lo[2]);
      const float3 inv = make_float3(ex > 0 ? 1.f / ex : 0.f, ey > 0 ? 1.f / ey : 0.f, ez > 0 ? 1.f / ez : 0.f);
      int rc_s;
      {
        ProfScope ps(KC_SORT);
        k_ray_keys<<<blocks_for(n), 256, 0, st>>>(single->planes, idx, (unsigned)n, single->minv, blo, inv, k_in, v_in, C.sort_bits, n_dev);
        if ((rc_s = sort_pairs_u32(k_in, k_out, v_in, v_out, n, C.sort_bits))) return rc_s;
      }
      HIPCHK(hipGetLastError());
      idx = v_out;
    }
    if (single) {
      // one segment (a single non-empty local queue, e.g. the one-domain benchmark): the single-mesh kernels -- no segment lookup and
      // no per-ray table loads at a refill -- with the same device-side counts
      gvt_hip_mesh *M = single->mesh;
      const bool have4 = M->d_nodes4 != nullptr;
      Trav TS{ M->d_nodes, M->d_tri, M->d_nodes4 };
      LongQ LQ{};
      if (use_long && have4) { LQ.recs = d_long; LQ.count = c + 3; long_limits(LQ, n); }
      LQ.steps_out = by_class ? d_steps : nullptr;
      const bool small1 = small && have4;
      // a coherent list (camera rays in 8x8 tiles, straight from the filter) over a mesh the builder found packet-friendly (gvt_hip_mesh::packet_ok;
      // knob packet: 0 never, 1 the mesh's own choice, 2 always): a wave walks the tree ONCE for 64 rays (k_packet) -- the reference chooses its packet
      // width per build as well (EmbreeMeshAdapter.cpp:50-74)
      const bool pkt = (C.packet == 2 || (C.packet == 1 && M->packet_ok && n >= (size_t)C.packet_min_rays)) && single->coherent && pass == 0 && have4 && !small1;
      // ... the CLOSEST hit, that is.  Shadow rays as packets lost on both surfaces they were measured on (bun_zipper any hit 0.162 -> 0.189 ms, the hall
      // 1.23 -> 1.33: an any-hit packet goes on until its LAST ray is occluded or through): under the per-mesh choice they stay a lane per ray; 2 forces both
      const bool pkt_any = pkt && C.packet == 2;
#ifdef GVT_EXPERIMENTS
      // the whole adapter call in ONE launch (experiments/fused_lean.inc; measured 1.19-1.27 ms per benchmark frame against 0.94 for the three launches) where the frame is the simple case: one instance in the scene, camera rays, depth 1, one
      // point / ambient light, a Lambert mesh material, the terminal sink on
      const bool lean1 = C.fused1 && !pkt && single->coherent && pass == 0 && passes == 1 && have4 && !small1 && nL == 1 && lights_host[0].type != GVT_HIP_LIGHT_AREA &&
                         P.sink.fb && P.sink.top.n_inst == 1 && !M->d_vcolors && !M->d_face_mat && M->mesh_mat.type == 0 && n >= (size_t)C.fused1_min_rays;
      if (lean1) {
        ProfScope ps(KC_CLOSEST);
        Frame1Shade K;
        std::memset(&K, 0, sizeof K);
        K.normi = single->normi; K.kd[0] = M->mesh_mat.kd[0]; K.kd[1] = M->mesh_mat.kd[1]; K.kd[2] = M->mesh_mat.kd[2]; K.normal_mode = P.normal_mode;
        K.slot_of = M->d_slot_of; K.tri_idx = M->d_tris; K.normals = M->d_normals; K.p3 = single->planes.p3; K.light = lights_host[0]; K.fb = P.sink.fb; K.n_pix = P.sink.n_pix;
        Frame1Shade *d_K = (Frame1Shade *)scratch_get(22, sizeof K);
        if (!d_K) return GVT_HIP_ERR_DEVICE;
        if (C.frame1_cached_dst != d_K || C.frame1_cached.size() != sizeof K || std::memcmp(C.frame1_cached.data(), &K, sizeof K) != 0) { // (uploaded when it changes: steady-state frames skip it)
          C.frame1_cached.assign((const unsigned char *)&K, (const unsigned char *)&K + sizeof K);
          C.frame1_cached_dst = d_K;
          HIPCHK(hipMemcpyAsync(d_K, C.frame1_cached.data(), sizeof K, hipMemcpyHostToDevice, st));
          HIPCHK(hipStreamSynchronize(st)); // pageable source
        }
        Frame1Args F;
        F.p0 = single->planes.p0; F.p1 = single->planes.p1; F.n = (unsigned)n; F.n_dev = n_dev; F.minv = single->minv; F.nodes4 = M->d_nodes4; F.tris = M->d_tri;
        F.tnear = GVT_RAY_EPSILON; F.counter = c + 0; F.spill_base = C.d_spill; F.refill_min = C.refill_min; F.inner_min = C.inner_min; F.shade = d_K;
        k_frame1<<<trav_grid2(n, true), TRAV_BLOCK, 0, st>>>(F);
        HIPCHK(hipGetLastError());
        C.stats.launches_closest++;
        continue;
      }
#endif
      if (pkt) {
        ProfScope ps(KC_CLOSEST);
        LongQ LP{ d_long, c + 3, nullptr, 0u, 0, 0 };
        k_packet<false><<<blocks_for(n), 256, 0, st>>>(single->planes, (unsigned)n, n_dev, single->minv, TS, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                                     TermSink{}, LP, nullptr, nullptr, nullptr, c + 9);
        k_long_closest<true><<<C.n_cu * 3, 256, 0, st>>>(single->planes, d_long, c + 3, single->minv, TS, GVT_RAY_EPSILON, d_hits, c + 4); // packets that bailed out
      } else
      if (small1) {
        ProfScope ps(KC_CLOSEST);
        k_long_seed<<<blocks_for(n), 256, 0, st>>>(d_long, c + 3, idx, (unsigned)n, n_dev);
        k_long_closest<true><<<small_grid, 256, 0, st>>>(single->planes, d_long, c + 3, single->minv, TS, GVT_RAY_EPSILON, d_hits, c + 4);
      }
#ifdef GVT_EXPERIMENTS
      else if (quad_usable(M)) {
        ProfScope ps(KC_CLOSEST);
        k_traceq<false, true, 0><<<quad_grid(n), 256, 0, st>>>(single->planes, idx, (unsigned)n, single->minv, TravQ{ M->d_nodes4q, M->d_triq }, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                                             c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, n_dev, TermSink{}, LQ);
      }
#endif
      else {
        ProfScope ps(KC_CLOSEST);
        launch_trace<false, true, 0>(have4, trav_grid2(n, true), st, single->planes, idx, (unsigned)n, single->minv, TS, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                     c + 0, C.d_spill, C.refill_min, C.inner_min, n_dev, C.share, (unsigned)C.share_min_rays, TermSink{}, LQ);
      }
      if (LQ.steps && !small1 && !pkt) {
        ProfScope ps(KC_LONG);
        k_long_closest<true><<<C.n_cu * 3, 256, 0, st>>>(single->planes, d_long, c + 3, single->minv, TS, GVT_RAY_EPSILON, d_hits, c + 4, WaveSet{}, LQ.stk);
      }
      ShadeArgs A{};
      A.in = single->planes; A.idx = idx; A.n = (unsigned)n; A.index_base = 0; A.hits = d_hits;
      A.first_pass = (pass == 0); A.carried_rng = 1; A.out = outp; A.out_count = out->d_count; A.shadow = shadow; A.shadow_count = c + 1;
      A.next_idx = next; A.next_count = c_next; A.lights = d_lights; A.normi = single->normi; A.normal_mode = P.normal_mode;
      A.n_lights = nL; A.seed = P.seed; A.zero_word = c + 0;
      A.sink = P.sink; A.sink.from = single->inst; A.update_in_place = 0;
      A.n_dev = n_dev; A.W = WaveSet{}; A.out_from = nullptr; A.shadow_inst = nullptr; A.shadow_stride = 0;
      if (pkt_any) { A.shadow_inst = d_shadow_inst; A.shadow_stride = (unsigned)n; } // shadow rays in the primaries' order: packets again
      A.steps = d_steps; A.cls_cnt = by_class ? c + SHADOW_CLS_WORD : nullptr; A.cls_stride = (unsigned)cls_stride; A.cls_lo = C.shadow_cls_lo; A.cls_shift = C.shadow_cls_shift;
      MeshView mv;
      mv.slots = M->d_tri; mv.slot_of = M->d_slot_of; mv.verts = M->d_verts; mv.tris = M->d_tris; mv.normals = M->d_normals; mv.vcolors = M->d_vcolors;
      mv.materials = M->d_materials; mv.n_mat = (unsigned)M->nMat; mv.face_mat = M->d_face_mat; mv.mat = M->mesh_mat;
      // the simple frame (k_shade's LEAN instantiation): camera rays of the scene's only instance, depth 1, no area light, a LAMBERT mesh material
      const bool lean_shade = single->coherent && pass == 0 && passes == 1 && P.sink.fb && P.sink.top.n_inst == 1 && nL >= 1 && !M->d_vcolors && !M->d_face_mat &&
                              M->mesh_mat.type == 0 && !A.update_in_place && std::all_of(lights_host, lights_host + nL, [](const gvt_hip_light &l) { return l.type != GVT_HIP_LIGHT_AREA; });
      {
        ProfScope ps(KC_SHADE);
        if (lean_shade) k_shade<false, true><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, mv);
        else k_shade<false><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, mv);
      }
      if (nL) {
        ProfScope ps(KC_ANY);
        TermSink sk = P.sink;
        sk.from = single->inst;
        if (pkt_any) {
          unsigned *d_retry = (unsigned *)scratch_get(18, sizeof(unsigned) * shadow_cap);
          if (!d_retry) return GVT_HIP_ERR_DEVICE;
          k_packet<true><<<blocks_for(shadow_cap), 256, 0, st>>>(shadow, (unsigned)shadow_cap, nullptr, single->minv, TS, GVT_RAY_EPSILON, nullptr, d_shadow_inst, outp, out->d_count,
                                                               sk, LongQ{}, d_retry, c + 6, (unsigned long long *)(c + 18), c + 9);
          // rays of packets that bailed out: one lane per ray (an empty list costs a few microseconds)
          launch_trace<true, true, 1>(have4, trav_grid2(4096), st, shadow, d_retry, 0u, single->minv, TS, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                      c + 0, C.d_spill, C.refill_min, C.inner_min, c + 6, C.share, (unsigned)C.share_min_rays, sk, LongQ{});
        } else
#ifdef GVT_EXPERIMENTS
        if (quad_usable(M) && !small1)
          k_traceq<true, true, 1><<<quad_grid(shadow_cap), 256, 0, st>>>(shadow, nullptr, 0u, single->minv, TravQ{ M->d_nodes4q, M->d_triq }, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                       c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, c + 1, sk, LongQ{});
        else
#endif
        if (small1) k_wave_any<false><<<(int)std::min<size_t>((shadow_cap + 3) / 4, (size_t)C.n_cu * 3), 256, 0, st>>>(shadow, c + 1, single->minv, TS, GVT_RAY_EPSILON, outp, out->d_count, c + 0, sk, MultiSrc{});
        else {
          MultiSrc by{};
          if (by_class) { by.cls_cnt = c + SHADOW_CLS_WORD; by.cls_stride = (unsigned)cls_stride; by.cls_total = c + 1; }
          launch_trace<true, true, 1>(have4, trav_grid2(shadow_cap), st, shadow, nullptr, 0u, single->minv, TS, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                      c + 0, C.d_spill, C.refill_min, C.inner_min, by_class ? nullptr : c + 1, C.share, (unsigned)C.share_min_rays, sk, LongQ{}, by);
        }
      }
      HIPCHK(hipGetLastError());
      C.stats.launches_closest++;
      C.stats.launches_any++;
      continue;
    }
#ifdef GVT_EXPERIMENTS
    if (C.fused && P.sink.fb) {
      // one launch: closest hit, shade, the first light's shadow rays, terminal rule (k_fused); lights 1.. through the list
      FusedArgs F;
      F.W = W; F.idx = idx; F.n = (unsigned)n; F.n_dev = n_dev; F.lights = d_lights; F.n_lights = nL; F.normal_mode = P.normal_mode;
      F.first_pass = (pass == 0); F.seed = P.seed; F.out = outp; F.out_count = out->d_count; F.out_from = d_out_from;
      F.shadow = shadow; F.shadow_count = c + 1; F.shadow_inst = d_shadow_inst; F.next_idx = next; F.next_count = c_next; F.sink = P.sink;
      F.counter = c + 0; F.spill_base = C.d_spill; F.refill_min = C.refill_min; F.inner_min = C.inner_min;
      F.tot = (unsigned long long *)(c + 16); F.tnear = GVT_RAY_EPSILON;
      {
        ProfScope ps(KC_CLOSEST);
        k_fused<<<trav_grid2(n), TRAV_BLOCK, 0, st>>>(F);
      }
      if (nL > 1) {
        k_set_u32<<<1, 64, 0, st>>>(c + 0, 0u); // the work counter of the launch that follows
        ProfScope ps(KC_ANY);
        MultiSrc MA{ W, d_shadow_inst, d_out_from, nullptr };
        k_trace<true, true, 1, false, true, true><<<trav_grid2(n * (size_t)(nL - 1)), TRAV_BLOCK, 0, st>>>(shadow, nullptr, 0u, id, T, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                                                       c + 0, C.d_spill, C.refill_min, C.inner_min, c + 1, C.share, (unsigned)C.share_min_rays,
                                                                                                       P.sink, LongQ{}, MA);
        C.stats.launches_any++;
      }
      HIPCHK(hipGetLastError());
      C.stats.launches_closest++;
      continue;
    }
#endif
    LongQ LQ{};
    if (use_long) { LQ.recs = d_long; LQ.count = c + 3; long_limits(LQ, n); }
    MultiSrc MS{ W, nullptr, nullptr, nullptr };
    if (d_hop) { MS.hop_inst = d_hop; MS.hop_read = pass > 0; MS.hop_owner = P.hop_owner; MS.hop_rank = P.hop_rank; MS.hop_tot = hop_tot; MS.hop_early = P.hop == 1 ? 1 : 0; }
    // the round's queues hold camera rays in tile order, over packet-friendly meshes only: the closest hits a packet of 64 rays per wave
    const bool pktm = multi_packets && pass == 0 && !small && idx == nullptr;
    if (pktm) {
      ProfScope ps(KC_CLOSEST);
      LongQ LP{ d_long, c + 3, nullptr, 0u, 0, 0 };
      k_packet_multi<<<blocks_for(n), 256, 0, st>>>(W, (unsigned)n, n_dev, GVT_RAY_EPSILON, d_hits, LP, c + 9);
      k_long_closest<true, true><<<C.n_cu * 3, 256, 0, st>>>(none, d_long, c + 3, id, T, GVT_RAY_EPSILON, d_hits, c + 4, W); // packets that bailed out
    } else
    if (small) {
      ProfScope ps(KC_CLOSEST);
      k_long_seed<<<blocks_for(n), 256, 0, st>>>(d_long, c + 3, idx, (unsigned)n, n_dev);
      k_long_closest<true, true><<<small_grid, 256, 0, st>>>(none, d_long, c + 3, id, T, GVT_RAY_EPSILON, d_hits, c + 4, W, nullptr, pass > 0 ? d_hop : nullptr);
    }
#ifdef GVT_EXPERIMENTS
    else if (C.quad && W.quad_ok) {
      ProfScope ps(KC_CLOSEST);
      k_traceq<false, true, 0, true><<<quad_grid(n), 256, 0, st>>>(none, idx, (unsigned)n, id, TravQ{}, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                                                 c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, n_dev, TermSink{}, LQ, MS);
    }
#endif
    else {
      ProfScope ps(KC_CLOSEST);
      k_trace<false, true, 0, false, true, true><<<trav_grid2(n, true), TRAV_BLOCK, 0, st>>>(none, idx, (unsigned)n, id, T, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                                                                      c + 0, C.d_spill, C.refill_min, C.inner_min, n_dev, C.share, (unsigned)C.share_min_rays,
                                                                                      d_hop ? P.sink : TermSink{}, LQ, MS);
    }
    if (use_long && !small && !pktm) {
      ProfScope ps(KC_LONG);
      k_long_closest<true, true><<<C.n_cu * 3, 256, 0, st>>>(none, d_long, c + 3, id, T, GVT_RAY_EPSILON, d_hits, c + 4, W, LQ.stk, d_hop);
    }
    ShadeArgs A{};
    A.in = none; A.idx = idx; A.n = (unsigned)n; A.index_base = 0; A.hits = d_hits;
    A.first_pass = (pass == 0); A.carried_rng = 1; A.out = outp; A.out_count = out->d_count; A.shadow = shadow; A.shadow_count = c + 1;
    A.next_idx = next; A.next_count = c_next; A.lights = d_lights; A.normi = P.normi; A.normal_mode = P.normal_mode;
    A.n_lights = nL; A.seed = P.seed; A.zero_word = c + 0;
    A.sink = P.sink; A.update_in_place = 0;
    A.n_dev = n_dev; A.W = W; A.out_from = d_out_from; A.shadow_inst = d_shadow_inst; A.hop_inst = d_hop;
    const bool direct = C.shadow_direct && pass == 0 && !small; // later passes and small rounds hold few rays: compacted slots
    // (k_shade marks every slot below the stride -- the bound -- empty or taken; with one light the list is as long as the traced one,
    // and where only the device knows that length the any-hit launch stops there instead of skipping empty slots up to the bound)
    const bool direct_dev = direct && nd0 != nullptr && nL == 1;
    A.shadow_stride = direct ? (unsigned)n : 0u;
    // (k_shade's LEAN instantiation for the merged chain: depth 1, no area light, plain LAMBERT meshes; a missing ray still walks the instances ahead of it)
    const bool lean_multi = simple_meshes && pass == 0 && passes == 1 && nL >= 1 && std::all_of(lights_host, lights_host + nL, [](const gvt_hip_light &l) { return l.type != GVT_HIP_LIGHT_AREA; });
    {
      ProfScope ps(KC_SHADE);
      if (lean_multi) k_shade<true, true><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, MeshView{});
      else k_shade<true><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, MeshView{});
    }
    if (nL) {
      ProfScope ps(KC_ANY);
      MultiSrc MA{ W, d_shadow_inst, d_out_from, direct ? (unsigned long long *)(c + 18) : nullptr };
      if (d_hop && P.sink.fb && P.hop == 2) { MA.hop_owner = P.hop_owner; MA.hop_rank = P.hop_rank; MA.hop_tot = hop_tot; } // (an un-occluded shadow ray goes on the same way -- not in the early mode: 1.372 -> 1.340 ms on the soup tiles without)
      if (small) k_wave_any<true><<<(int)std::min<size_t>((shadow_cap + 3) / 4, (size_t)C.n_cu * 3), 256, 0, st>>>(shadow, c + 1, id, T, GVT_RAY_EPSILON, outp, out->d_count, c + 0, P.sink, MA);
#ifdef GVT_EXPERIMENTS
      else if (C.quad && W.quad_ok)
        k_traceq<true, true, 1, true><<<quad_grid(shadow_cap), 256, 0, st>>>(shadow, nullptr, direct ? (unsigned)shadow_cap : 0u, id, TravQ{}, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                           c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, direct ? (direct_dev ? nd0 : nullptr) : c + 1, P.sink, LongQ{}, MA);
#endif
      else
      k_trace<true, true, 1, false, true, true><<<trav_grid2(shadow_cap), TRAV_BLOCK, 0, st>>>(shadow, nullptr, direct ? (unsigned)shadow_cap : 0u, id, T, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                                              c + 0, C.d_spill, C.refill_min, C.inner_min, direct ? (direct_dev ? nd0 : nullptr) : c + 1, C.share, (unsigned)C.share_min_rays,
                                                                                              P.sink, LongQ{}, MA);
    }
    HIPCHK(hipGetLastError());
    C.stats.launches_closest++;
    C.stats.launches_any++;
  }
  if (!defer_end) k_wave_end<<<(unsigned)((std::max(n_inst, 1) + 255) / 256), 256, 0, st>>>(c, d_count_ptr, d_mask, n_inst); // + queue[instTarget].clear()
  HIPCHK(hipGetLastError());
  C.stats.trace_calls++;
  return 0;
}

// A small round in ONE launch (finish_kernel.inc): the rays of the segments in W are traced, shaded, their shadow rays traced and
// everything that moves on is followed through this rank's instances; rays for other ranks' instances are appended to their queues
// (which must have room: n_total * (1 + n_lights * depth) each), everything else ends in the framebuffer.  No shuffle follows.
// the context's device copy of the light list is this list (finish_round / wave_trace_chain upload it when it changes): a launch may use it without a copy
bool finish_lights_resident(const gvt_hip_light *lights_host, int nL) {
  Ctx &C = gctx();
  const size_t bytes = sizeof(gvt_hip_light) * (size_t)nL;
  return !nL || (C.lights_cached_dst && C.lights_cached_dst == C.scratch[5] && C.lights_cached.size() == bytes && std::memcmp(C.lights_cached.data(), lights_host, bytes) == 0);
}
// spec_words != nullptr: the launch is enqueued ahead of the host's knowledge (domain.hip): n_total is only the bound its grid is sized by, the count is
// spec_words[2], the launch is void unless spec_words[1]; the light list must be resident already (no upload, no synchronisation in here)
int finish_round(const WaveSet &W, size_t n_total, const TraceParams &P, const gvt_hip_light *lights_host, const void *d_qdesc, const int *d_owner, int rank,
                 unsigned *d_queue_overflow, unsigned *const *d_count_ptr, const unsigned char *d_mask, const unsigned *spec_words) {
  Ctx &C = gctx();
  if (!n_total) return 0;
  hipStream_t st = C.stream;
  const int nL = P.n_lights;
  gvt_hip_light *d_lights = (gvt_hip_light *)scratch_get(5, sizeof(gvt_hip_light) * (nL > 0 ? nL : 1));
  if (!d_lights) return GVT_HIP_ERR_DEVICE;
  {
    std::vector<unsigned char> &cached = C.lights_cached;
    const void *&cached_dst = C.lights_cached_dst;
    const size_t bytes = sizeof(gvt_hip_light) * (size_t)nL;
    if (nL && (cached_dst != d_lights || cached.size() != bytes || std::memcmp(cached.data(), lights_host, bytes) != 0)) {
      if (spec_words) { set_error("finish_round: a speculative launch needs the light list resident"); return GVT_HIP_ERR_INVALID; }
      cached.assign((const unsigned char *)lights_host, (const unsigned char *)lights_host + bytes);
      cached_dst = d_lights;
      HIPCHK(hipMemcpyAsync(d_lights, cached.data(), bytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  FinishArgs A;
  A.n_dev = spec_words ? spec_words + 2 : nullptr; A.valid = spec_words ? spec_words + 1 : nullptr;
  A.W = W; A.n = (unsigned)n_total; A.lights = d_lights; A.n_lights = nL; A.normal_mode = P.normal_mode;
  A.top = P.sink.top; A.fb = P.sink.fb; A.n_pix = P.sink.n_pix;
  A.queues = (const QueueDesc *)d_qdesc; A.owner = d_owner; A.rank = rank;
  A.counter = C.d_counters + 0; A.tot = (unsigned long long *)(C.d_counters + 16);
  A.queue_overflow = d_queue_overflow; A.trav_overflow = C.d_counters + TRAV_OVF_WORD;
  A.seed = P.seed; A.index_base = 0ull; A.skip_known = C.skip_known;
  A.count_ptr = d_count_ptr; A.clear_mask = d_mask; A.n_inst = W.n_inst;
  // (the work counter is 0: the frame's start and every round's report leave it so -- k_zero_totals, k_round_report)
  {
    ProfScope ps(KC_CLOSEST);
    k_finish<<<(int)std::min<size_t>((n_total + 3) / 4, (size_t)C.n_cu * 4), 256, 0, st>>>(A); // 40 KiB of LDS per block: four blocks per CU
  }
  HIPCHK(hipGetLastError());
  C.stats.launches_closest++;
  C.stats.trace_calls++;
  return 0;
}

int set_device_u32(unsigned *p, unsigned v) {
  k_set_u32<<<1, 64, 0, gctx().stream>>>(p, v);
  HIPCHK(hipGetLastError());
  return 0;
}
