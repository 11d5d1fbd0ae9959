// gvt_internal.h -- host-side objects behind the opaque handles of include/gvt_hip.h
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "gvt_device.h"

enum KernelClass { KC_CLOSEST = 0, KC_ANY, KC_SHADE, KC_CONVERT, KC_SHUFFLE, KC_CAMERA, KC_BUILD, KC_SORT, KC_LONG, KC_COUNT };

struct PendingEvent {
  hipEvent_t a, b;
  int cls;
};

// tuning knobs (gvt_hip_set_option); results never depend on them.  "defaults" restores this initial state.
// the shadow list of a single-mesh round in SHADOW_CLASSES regions (class = how long the primaries of a 64-ray tile took), their ray counts in the
// context's counter words SHADOW_CLS_WORD.. (reset with the pass's other counters)
#define SHADOW_CLASSES 8
#define SHADOW_CLS_WORD 24
struct Knobs {
  int trav_kernel = 1;   // 1 = persistent waves with lane refill (k_trace), 0 = one 64-ray batch at a time (k_closest/k_any)
  int blocks_per_cu = 4; // k_trace grid: resident 256-thread blocks per CU
  int blocks_per_cu_closest = 5; // ... for closest-hit launches (0: blocks_per_cu); 5: 0.528 vs 0.554 ms per 1 M rays (4) and 0.552 (6); any-hit: 4 is best (5: 0.50 vs 0.40)
  int refill_min = 16;   // k_trace: idle lanes needed before a refill
  int inner_min = 32;    // k_trace: the inner-node loop is left once fewer lanes than this still descend
  int coop_fetch = 0;    // k_trace: quad-cooperative 64-byte fetches (DPP transpose) instead of 4 loads per lane
  int wide4 = 1;         // k_trace: traverse the compressed 4-wide collapse (64-B nodes, four 8-bit child boxes per fetch)
  int share = 1;         // k_trace drain-phase work sharing, bit 0: any-hit launches (0.66 vs 0.80 ms per 1 M shadow rays), bit 1: closest-hit
                         // launches (no gain: the pending subtrees of a closest-hit ray are mostly pruned by its eventual hit)
  int share_min_rays = 131072; // ... only in launches of at least this many rays (15 K shadow rays: the hand-off costs more than the tail it trims)
  int sort_rays = 0;     // Morton-sort the rays of a list before traversal (pays on incoherent lists; camera rays arrive in 8x8 tiles and the shuffle keeps list order)
  int sort_gather = 0;   // after sorting, traverse a contiguous object-space copy (o,d) of the rays
  int sort_bits = 20;    // radix-sorted key width (8 bits per rocPRIM pass)
  int long_steps = 96;   // closest hit: a ray that exceeds this many 4-wide node steps is parked and finished by a whole wave (0: off)
  int long_steps_drain = 0; // ... after the launch's work counter ran dry (k_trace waves in their drain phase); 0 or >= long_steps: the same limit.
                         // Measured (10 M soup, 1 M rays): 72 shortens the closest launch 0.519 -> 0.504 ms but k_long_closest grows 0.044 -> 0.094
  int long_save = 1;     // parked rays carry their pending stack and go on from it (0: they start again at the root with their best hit as the bound)
  int long_min_rays = 65536; // ... only in launches of at least this many rays (small launches have no tail to speak of)
  int fused = 0;         // scheduler rounds: closest hit + shade + first-light shadow rays in one kernel (k_fused) instead of three launches.
                         // Measured 2.4x SLOWER than the three launches (EXPERIMENTS.md): shading inside the persistent kernel is latency-exposed
  int packet = 1;        // scheduler rounds: camera rays in tile order (and their direct-mapped shadow rays) traversed a packet of 64 per wave (k_packet): 0 never,
                         // 1 on meshes the builder found packet-friendly (gvt_hip_mesh::packet_ok), 2 always.  4-10 % faster on surfaces (bun_zipper, the hall),
                         // 2.5x-20x SLOWER on the random soups (EXPERIMENTS.md): hence the per-mesh choice
  int fused1 = 0;        // experiments build: one-instance depth-1 frames with one point / ambient light and a Lambert mesh material run the whole adapter call in ONE
                         // launch (k_frame1: closest hit -> lean shade -> the lane goes on with its shadow ray -> deposit).  Bit-exact, and SLOWER: 1.19-1.27 ms per
                         // benchmark frame against 0.94 for the three launches (EXPERIMENTS.md round 5): closest-hit and any-hit work co-resident costs more than the tails
  int fused1_min_rays = 65536; // ... in launches of at least this many rays
  int shadow_order = 1;  // single-mesh rounds with one light: the shadow rays are listed by the step count of their primaries' tiles, the longest first (shade.inc);
                         // 0: in arrival order.  The any-hit launch's drain is then left to short rays: 0.318 -> 0.281 ms in tools/order_probe.py
  int shadow_cls_lo = 24, shadow_cls_shift = 3; // ... class of a 64-ray tile = (node steps of its longest primary - lo) >> shift, clamped to 0..7 (tuned constants)
  int shadow_order_min_rays = 262144; // ... in launches of at least this many rays (a small launch has no drain worth ordering)
  int packet_min_rays = 524288; // ... and only in launches of at least this many rays (bound): a small launch is a few thousand packets, each a long serial walk
  int packet_sah_max = 128; // meshes created afterwards: packet-friendly when sum(area(inner node)) / area(root) is at most this (lbvh.hip k_sah_sum)
  int round_room_mb = 16384; // scheduler rounds: memory the worst-case reservation of the destination queues may add (MiB); beyond it the round shuffles with exact growth
  int finish_rays = 32768; // scheduler rounds holding at most this many rays are run by ONE kernel that follows every ray to its end on this rank (k_finish):
                         // no per-hop rounds for the few rays that move between the rank's own domains (0: off)
  int finish_auto = 1;   // one rank, several instances: whether small rounds go through k_finish at all is decided per tracer by timing a few frames each way
                         // (following every ray to its end in one launch wins where rays hop many times -- soup tiles --, per-hop rounds where a hop is a long
                         // traversal of its own -- a row of bunnies); re-probed every 2048 frames.  Results never depend on it.
  int small_rays = 4096; // scheduler rounds holding at most this many rays give every ray a whole wave (k_long_closest / k_wave_any): ~40 us
                         // per traversal launch instead of the ~150 us latency floor of a one-lane-per-ray launch
  int abi_lanes_n = 4;   // gvt_hip_trace on a host RayVector: host threads (each with a context of its own) that pipeline the list's chunks (0: one shot)
  int abi_chunk = 262144; // ... rays per chunk
  int abi_pipe_min = 131072; // ... lists shorter than this take the one-shot path
  int lean_frame = 1;    // one-instance scenes on one rank: framebuffer clear, counter resets and the chain's begin / end folded into the camera filter's
                         // two kernels and the round's report (8 launches per frame instead of 14)
  int skip_known = 0;    // shuffleRays' known-miss shortcut (gvt_device.h): a ray is not traced again in an instance it has already crossed without a hit on the
                         // same straight segment.  OFF by default = the reference's hop-by-hop rule, ray for ray.  On, it saves the hand-back hops between
                         // overlapping boxes (exchanges 6 / 8 / 8 -> 2 / 4 / 6 on 2 / 4 / 8 soup tiles), but it is NOT image-identical in general: a re-trace from
                         // the advanced origin can flip an edge-grazing triangle test that missed from the earlier origin (round 5, found by the strict checker:
                         // 9 of 575,174 shadow rays, 2 of 147,456 pixels of the hall in 8 slabs) -- an approximation a user may opt into, never the default
  int long_auto = 1;     // native tracer: raise the parking threshold from frame to frame while more than 0.3 % of a frame's closest-hit rays get parked (sparser scenes than the benchmark)
  int payload_overlap_kb = 1024; // Domain scheduler: a tick's payload of at least this many KiB (sent + received) moves on the communicator's own stream while the next
                         // chain runs; smaller ones stay on the compute stream (no cross-stream event pairs).  0: every payload on its own stream
  int inline_kb = 16;    // Domain scheduler: a pair's payload of at most this many KiB per tick travels INSIDE the announce (one exchange per tick instead of two); every
                         // rank must use the same value (the announce message has a fixed length).  0: always the two-step exchange
  int comm_cus = 0;      // Domain scheduler: payloads that move on the communicator's own stream (payload_overlap_kb) get this many compute units to themselves:
                         // the communicator's stream is created with a CU mask of that many CUs and the persistent traversal grids are sized for the rest (0: no reservation)
  int hop_local = 1;     // merged chains: a ray without a hit goes on into the next LOCAL instance inside the traversal launch (shuffleRays' rule applied by the lane) instead of waiting
                         // for the next round: 0 never, 2 always, 3 = early: in the closest-hit launch only and only while it still has rays to hand out (a ray that goes on during the drain stretches the launch's tail), 1 = per tracer --
                         // never / early / always timed like finish_auto on one rank; by the meshes' kind (surfaces: always, else never) on several
  int finish_clusters = 1; // k_finish walks the cluster layout of the 4-wide nodes (two levels per memory round trip; built per mesh when a tracer with several instances is made)
  int spec_ticks = 1;    // Domain scheduler, asynchronous ticks: the next tick's local work (a small round through k_finish) and its report are enqueued BEHIND the
                         // current tick's exchange before the host has read that exchange's result; the device itself voids them when the result calls for the
                         // host (a payload beyond the inline area in or out, an error, more rays than finish_rays).  0: every tick waits for the host first
  int comm_stream = 0;   // Domain scheduler: 1 = every exchange of a frame on the communicator's OWN stream, ordered against the compute stream by events (round 3's
                         // arrangement; also GVT_HIP_COMM_STREAM in the environment); 0 = on the compute stream itself, only large payloads beside it (payload_overlap_kb)
  int frame_timing = 0;  // multi-rank frames: fill gvt_hip_frame_stats' ms_chain / ms_announce / ms_payload / ms_composite (five more event calls per exchange)
  int inject_fail_tick = -1; // tests: this rank's local work "fails" at that exchange of a multi-rank frame (the announce carries the error to every rank)
  int report_poll = 1;   // one rank: a round's report is written into pinned host memory by the kernel and polled (no copy, no stream synchronisation)
  int first_round_async = 1; // one-instance scenes on one rank: no read-back after the camera filter (the chain reads its ray count on the device)
  int wave_single = 1;   // scheduler rounds: a round with ONE non-empty local queue uses the single-mesh kernels (no per-ray segment / instance lookups)
  int shadow_direct = 1; // scheduler rounds: shadow rays in direct-mapped slots (the order of the traced list) instead of block-arrival order
  int term_sink = 1;     // gvt_hip_trace_queue_sink: deposit terminal shadow rays from the any-hit kernel (0: always through moved_rays)
  int camera_tile = 8;   // gvt_hip_image_frame: camera rays listed in 8x8-pixel tiles (0: pixel-major like generateRays)
  int top_ordered = 1;   // shuffle: order-preserving, deterministic slots (<= 64 destinations) instead of arrival-order atomics
  int top_lds = 1;       // shuffle kernels: aggregate destination counters in LDS per 1024-thread block
  int quad = 0;          // experiments build: four lanes per ray (k_traceq, experiments/quad_kernel.inc) instead of one (k_trace); meshes created while it
                         // is set carry the quad layouts.  Measured 1.5x slower on the 10 M soup (VALU bound: 16 rays per wave), EXPERIMENTS.md
  int leaf_max = 2;      // triangles per leaf at mesh build (1..4): closest hit on the 10 M soup 0.51 ms (2) vs 0.57 ms (4)
  int quad_inner_min = 8; // k_traceq: the node loop is left once fewer quads than this still descend
  int quad_refill_min = 4; // k_traceq: idle quads needed before a refill
  int blocks_per_cu_quad = 8; // k_traceq grid: resident 256-thread blocks per CU
};

struct Ctx : Knobs {
  bool ready = false;
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int profile = 0; // 0 off, 1 every kernel class, 2 the traversal kernels only (closest, long, any)
  int long_steps_override = 0; // > 0: the parking threshold of the frame in progress (gvt_hip_tracer_frame's long_auto), instead of Knobs::long_steps
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  gvt_hip_stats stats{};
  // traversal launch geometry + per-thread stack spill area
  int n_cu = 256;        // compute units the context's launches are sized for (all of the device's, minus cu_reserved)
  bool shadow_order_denied = false; // the class-ordered shadow list (8 x the plain one) could not be allocated once: this context lists shadow rays in arrival order
  int cu_reserved = 0;   // compute units masked away from the context's stream for a communicator's own stream (knob comm_cus)
  int trav_blocks = 0;
  int *d_spill = nullptr;
  unsigned *d_counters = nullptr; // small array of device counters (work fetch, temps)
  // pinned host scratch for small read-backs
  unsigned *h_pinned = nullptr;
  // grow-only device scratch arenas (never freed inside hot calls)
  void *scratch[24] = { nullptr };
  size_t scratch_bytes[24] = { 0 };
  // light list of the last trace call (uploaded only when it changes)
  std::vector<unsigned char> lights_cached;
  const void *lights_cached_dst = nullptr;
  std::vector<unsigned char> frame1_cached; // k_frame1's shading constants as last uploaded (trace.hip)
  const void *frame1_cached_dst = nullptr;
  // staging queues of gvt_hip_trace (host RayVector in / out)
  gvt_hip_queue *abi_qin = nullptr, *abi_qout = nullptr;
  std::vector<Ctx *> abi_lanes; // contexts of the pipelined host path's lanes (api.hip trace_pipelined), created on first use
  void *abi_pool = nullptr;     // ... and their persistent threads (api.hip AbiPool)
};
// The context API calls run on: the calling thread's current context (gvt_hip_ctx_make_current), else the process default one
// (gvt_hip_init).  A context owns a stream, scratch arenas, counters, statistics and knobs; meshes / queues / framebuffers are plain
// device objects usable from any context of their device, one at a time.
Ctx &gctx();
void set_error(const char *fmt, ...);
int ensure_init();
void *scratch_get(int slot, size_t bytes); // grow-only; contents NOT preserved on growth
void scratch_release(int slot); // gives a slot's memory back (after a synchronisation of the context's stream)

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
      return GVT_HIP_ERR_DEVICE;                                                                  \
    }                                                                                             \
  } while (0)

// profiling bracket: records HIP events on the launch stream around `stmt` when enabled
struct ProfScope {
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  explicit ProfScope(int cls);
  ~ProfScope();
};

struct gvt_hip_mesh {
  size_t nV = 0, nT = 0;
  float *d_verts = nullptr;   // nV*3
  int *d_tris = nullptr;      // nT*3
  float *d_normals = nullptr; // nV*3
  float *d_vcolors = nullptr; // nV*3 or null
  gvt_hip_material *d_materials = nullptr;
  size_t nMat = 0;
  int *d_face_mat = nullptr;
  gvt_hip_material mesh_mat;
  // acceleration structure
  BvhNode *d_nodes = nullptr;
  size_t nNodes = 0;
  float4 *d_tri = nullptr; // 4 float4 per slot, leaf order: (v0 | prim), (e1 | v1.x), (e2 | v1.y), (v1.z, v2)
  unsigned *d_slot_of = nullptr; // primID -> slot (the shading kernel reads the triangle from the slot the traversal touched)
  uint4 *d_nodes4 = nullptr;  // compressed 4-wide collapse: 4 x 16 B (64 B) per node
  size_t nNodes4 = 0;
  std::vector<unsigned> levels4; // nodes per level of the breadth-first 4-wide array (build_nodes4)
  uint4 *d_nodes4c = nullptr; // the same nodes in CLUSTER order for the wave-per-ray traversals (build_nodes4c, lbvh.hip): every even-level node followed by its
                              // inner children; references out of a cluster = (slot of the even node << 4) | mask of its inner children.  Built on demand
  int root_entry4c = 0;       // ... the root's reference in that form
  uint4 *d_nodes4q = nullptr; // the same nodes laid out for the quad-per-ray traversal: piece s = child s (quad_kernel.inc)
  float4 *d_triq = nullptr;   // leaf blocks, transposed (lbvh.hip k_emit_trisq)
  int leaf_max = 2;           // triangles per leaf this mesh was built with
  size_t nLeaves = 0;
  float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
  float build_ms = 0.f;
  float sah_inner = 0.f;      // sum over the inner nodes of area(node) / area(root)
  bool packet_ok = false;     // coherent lists over this mesh are traversed a packet per wave (k_packet)
};

struct gvt_hip_queue {
  float4 *d_planes = nullptr; // 4 float4 planes of `cap` + the uint32 stream-word plane (GVT_QUEUE_BYTES_PER_RAY * cap)
  size_t cap = 0;
  size_t size = 0;            // host mirror, always valid between API calls
  unsigned *d_count = nullptr;
};

struct gvt_hip_top {
  size_t n = 0;
  std::vector<int> order;   // DFS leaf order of the reference's top-level BVH
  std::vector<float> h_lo, h_hi; // the instance boxes as given (host copies: the projection of a rank's instances onto the film)
  float4 *d_lo = nullptr;   // in `order` order: (lo.xyz, inst id)
  float4 *d_hi = nullptr;
  float4 *d_nlo = nullptr, *d_nhi = nullptr; // the BVH's nodes (gvt_device.h TopDev), root = 0
  size_t n_nodes = 0;
  TopDev dev() const { return TopDev{ d_lo, d_hi, (int)n, d_nlo, d_nhi, (int)n_nodes }; }
  unsigned *d_hist = nullptr; // n counters
  void **d_qdesc = nullptr;   // device array of queue descriptors
  unsigned *h_hist = nullptr; // pinned staging for the two (asynchronous copies, no pageable bounce)
  void *h_qdesc = nullptr;
  std::vector<unsigned char> qdesc_uploaded; // what d_qdesc holds when the asynchronous shuffle last uploaded it
};

struct gvt_hip_fb {
  int w = 0, h = 0;
  float *d_rgba = nullptr;
};

struct QueueDesc { // device-visible view of one destination queue
  float4 *planes;
  unsigned long long cap;
  unsigned *count;
  unsigned keep;
};

// Terminal rule of shuffleRays (TracerBase.h:396-400) applied where the any-hit kernel retires an un-occluded shadow ray: a ray that
// meets no other instance is not appended to moved_rays but deposits color*w in the framebuffer at once (fb == nullptr: disabled).
struct TermSink {
  TopDev top; // the instance set (gvt_hip_top)
  int from;
  float *fb;
  unsigned n_pix;
};
struct TraceParams {
  Mat4 m, minv;
  Mat3 normi;
  int normal_mode;
  uint32_t seed;
  int n_lights;
  TermSink sink;
  int update_in_place; // Adapter::trace updates rayList in place; device-queue callers clear the list afterwards and skip that write
  int carried_rng;     // device-queue callers: a ray's RNG stream is the word it carries (gvt_device.h, plane 4)
  // merged chains of the schedulers: rays that leave an instance without a hit go on into the next instance of this rank inside the launch (trace_lane.inc MultiSrc)
  int hop;              // 0 no, 2 yes, 1 = the closest-hit launch only, and only while it still has rays to hand out (MultiSrc::hop_early)
  const int *hop_owner; // instance -> rank on the device (null: all local)
  int hop_rank;
};

// what k_shade needs of a mesh (shading attributes in their reference shapes)
struct MeshView {
  const float4 *slots;      // triangle slots in leaf order (gvt_hip_mesh::d_tri) and primID -> slot
  const unsigned *slot_of;
  const float *verts;
  const int *tris;
  const float *normals;
  const float *vcolors;
  const gvt_hip_material *materials;
  unsigned n_mat;
  const int *face_mat;
  gvt_hip_material mat; // Mesh::mat
};

// ---- merged ("wave") launches: the rays of ALL local instance queues of a scheduler round go through ONE closest-hit launch, one
// shade launch and one any-hit launch.  A launch's ray index space [0, n_total) is the concatenation of the queues (segments); every
// ray finds its segment (-> queue planes, instance) by a binary search over the segment starts, and its instance's transform,
// acceleration structure and shading attributes in a per-instance table.  Replaces one adapter call per non-empty queue (each with
// its own latency-bound traversal tail and read-backs) by one launch chain per round.
struct WaveSeg { // one local queue: rays [begin, begin + n) of the launch
  float4 *planes;
  unsigned long long cap;
  unsigned begin, n;
  int inst, pad;
};
struct WaveInst { // per instance (Adapter::trace arguments m / minv / normi + the adapter's mesh)
  Mat4 minv;
  Mat3 normi;
  int root_entry4c;      // the root's reference in the cluster layout (nodes4c)
  int pad[2];
  const uint4 *nodes4;
  const float4 *tris;
  const uint4 *nodes4c;  // cluster layout for the wave-per-ray traversals (null: not built; they walk nodes4 then)
  const uint4 *nodes4q;  // quad layouts (null: the mesh was built without them)
  const float4 *trisq;
  MeshView mv;
};
struct WaveSet {
  const WaveSeg *segs;
  const WaveInst *insts;
  int n_seg;
  int quad_ok; // every instance that can be traced here carries the quad layouts (nodes4q / trisq)
  int n_inst;  // rows of `insts` (0: unknown; the kernels then read the tables from global memory)
};

// lbvh.hip
int build_lbvh(gvt_hip_mesh *M);
int trav_overflow_fetch_async();
int trav_overflow_result();
int build_nodes4(gvt_hip_mesh *M); // lazily, when the wide4 option is on
int build_nodes4c(gvt_hip_mesh *M); // the cluster layout of the 4-wide nodes (on demand: tracers with several instances / ranks)
int sort_pairs_u32(unsigned *keys_in, unsigned *keys_out, unsigned *vals_in, unsigned *vals_out, size_t n, int end_bit);
// trace.hip
int queue_reserve(gvt_hip_queue *q, size_t cap);
int trace_core(gvt_hip_mesh *M, RayPlanes in, size_t n, uint64_t index_base, gvt_hip_queue *out, const TraceParams &P,
               const gvt_hip_light *lights_host);
int launch_closest(gvt_hip_mesh *M, RayPlanes q, const unsigned *idx, size_t n, bool xform, const Mat4 &minv, float tnear,
                   gvt_hip_hit *d_hits, bool counter_is_zero = false);
int launch_visit_stats(gvt_hip_mesh *M, RayPlanes q, size_t n, float tnear, unsigned *d_out);
int launch_wide_visit_stats(gvt_hip_mesh *M, RayPlanes q, size_t n, float tnear, const unsigned char *d_marks, unsigned *d_out);
int wide_root_marks(gvt_hip_mesh *M, int width, unsigned char *d_marks, size_t *n_wide); // lbvh.hip
int launch_any_flags(gvt_hip_mesh *M, RayPlanes q, size_t n, bool xform, const Mat4 &minv, float tnear, int *d_flags);
int set_device_u32(unsigned *p, unsigned v); // stream-ordered store of a host-known value
// one round's merged launch chain (no host round trip inside); `out` must have room for n_total * (1 + n_lights * passes) rays and
// is filled from slot 0 (its device count is reset); d_out_from receives the source instance of every ray appended
struct WaveSingle { // the launch has ONE segment: its queue planes and instance, for the single-mesh kernels
  RayPlanes planes;
  gvt_hip_mesh *mesh;
  Mat4 minv;
  Mat3 normi;
  int inst;
  int coherent; // the queue holds camera rays in tile order, straight from the filter: packet traversal (k_packet)
  const unsigned *n_dev; // the first pass's ray count lives in device memory (the queue's count word; n_total is only its bound)
  int pass0_begun;       // the producer of the queue (k_cam1_scatter) has already done k_wave_pass_begin's pass-0 resets
};
// defer_end: the caller's next kernel (k_round_report) does k_wave_end's work
// n_dev0_multi (merged kernels): device word holding the length of the merged list where only the device knows it; n_total is then a bound
int wave_trace_chain(const WaveSet &W, size_t n_total, int passes, gvt_hip_queue *out, int *d_out_from, const TraceParams &P,
                     const gvt_hip_light *lights_host, const WaveSingle *single, unsigned *const *d_count_ptr, const unsigned char *d_mask, int n_inst,
                     bool defer_end = false, const unsigned *n_dev0_multi = nullptr, bool multi_packets = false, bool simple_meshes = false);
// multi_packets (merged kernels, first pass): the queues hold camera rays in tile order over packet-friendly meshes -- closest hits through k_packet_multi
int finish_round(const WaveSet &W, size_t n_total, const TraceParams &P, const gvt_hip_light *lights_host, const void *d_qdesc, const int *d_owner, int rank,
                 unsigned *d_queue_overflow, unsigned *const *d_count_ptr, const unsigned char *d_mask, const unsigned *spec_words = nullptr);
bool finish_lights_resident(const gvt_hip_light *lights_host, int nL); // the context's device copy of the light list is this list (trace.hip)
int convert_aos_to_planes(const gvt_hip_ray *d_src, size_t n, RayPlanes dst, size_t dst_off, bool keep_state);
int convert_planes_to_aos(RayPlanes src, size_t src_off, size_t n, gvt_hip_ray *d_dst);
int convert_od_to_planes(const float *d_org, const float *d_dir, size_t n, RayPlanes dst);
