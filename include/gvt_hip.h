/*
 * gvt_hip.h -- C ABI of the MI355X (gfx950) engine adapter for GraviT: gvt::render::adapter::hip.
 *
 * This is the drop-in boundary.  Every entry point is what a GraviT-side binding for the
 * adapter path would call; the reference interface each one replaces is cited as file:line
 * relative to the GraviT source tree.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - Rays are the reference's 80-byte `gvt::render::actor::Ray` image (actor/Ray.h:68-96),
 *     16-byte aligned, exactly what Ray::pack() puts on the wire (Ray.h:161-174).
 *   - Materials are the reference's 92-byte `Material` POD (primitives/Material.h:59-90).
 *   - Matrices are glm column-major: m[16], minv[16], normi[9] (api.cpp:303-308).
 *   - Lights are 64-byte tagged PODs (the CPU reference uses virtual classes, scene/Light.h:46-104;
 *     POD precedent adapter/optix/Light.cuh:44-112).
 *   - Triangles are 0-based vertex triples (what the adapters read through std::get<>,
 *     EmbreeMeshAdapter.cpp:152-155).
 *   - Every function returning int returns 0 on success and a negative code on error; the
 *     message is available from gvt_hip_last_error().  Nothing calls exit() (the reference's
 *     Embree error handler does, EmbreeMeshAdapter.cpp:90-123).
 *   - All work of a context is issued on that context's HIP stream (gvt_hip_set_stream); a process has a default context and may
 *     create more (gvt_hip_ctx_create), one per thread; calls on one mesh/queue are serialised by the caller, as the reference's
 *     schedulers do (ImageTracer.h:241-248).
 */
#ifndef GVT_HIP_H
#define GVT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision of this header: bumped whenever an entry point changes meaning or an out-struct (gvt_hip_mesh_info, gvt_hip_stats, gvt_hip_frame_stats)
 * changes size.  A binding compares it with gvt_hip_abi_version() of the library it loaded before its first call (gravit_amd/capi.py, HipMeshAdapter.cpp do):
 * a caller built against another revision would have out-structs written past their end or flags read with another meaning, silently.
 *   6: gvt_hip_queue_append keeps its round-4 meaning (4th argument: the source is device memory); the flag word lives in gvt_hip_queue_append_flags */
#define GVT_HIP_ABI_VERSION 6
int gvt_hip_abi_version(void); /* GVT_HIP_ABI_VERSION the library was built from; no device access */

#define GVT_HIP_OK 0
#define GVT_HIP_ERR_INVALID (-1)  /* bad argument */
#define GVT_HIP_ERR_DEVICE (-2)   /* HIP runtime error */
#define GVT_HIP_ERR_CAPACITY (-3) /* output buffer / queue too small; nothing lost, see call */
#define GVT_HIP_ERR_NODEVICE (-4) /* no gfx950 device visible */
#define GVT_HIP_ERR_TIMEOUT (-5)  /* ray exchange: a peer did not answer within the communicator's deadline; the communicator is dead */
#define GVT_HIP_ERR_PEER (-6)     /* ray exchange: another rank reported an error in its announce; every rank leaves the frame with this */

#define GVT_HIP_NORMALS_FLAT 0   /* EmbreeMeshAdapter.cpp:75,520-522 (FLAT_SHADING) */
#define GVT_HIP_NORMALS_SMOOTH 1 /* EmbreeMeshAdapter.cpp:505-518, EmbreeStreamMeshAdapter.cpp:708, OptixMeshAdapter.cu:298 */

typedef struct gvt_hip_ray { /* actor/Ray.h:68-96 */
  float origin[3];
  float t_min;
  float direction[3];
  float t_max;
  float color[3];
  float t;
  int32_t id;
  int32_t depth;
  float w;
  int32_t type; /* 0 PRIMARY, 1 SHADOW, 2 SECONDARY */
  float pad[4];
} gvt_hip_ray;

typedef struct gvt_hip_material { /* primitives/Material.h:59-90 */
  int32_t type;                 /* 0 LAMBERT, 1 PHONG, 2 BLINN, 3 EMBREE_MATERIAL_METAL, 4 _VELVET, 5 _MATTE (Material.h:50-57) */
  float ka[3], ks[3], kd[3];
  float alpha;
  float eta[3], k[3];
  float roughness;
  float horizonScatteringColor[3];
  float backScattering, horizonScatteringFallOff;
} gvt_hip_material;

#define GVT_HIP_LIGHT_POINT 0
#define GVT_HIP_LIGHT_AREA 1
#define GVT_HIP_LIGHT_AMBIENT 2
typedef struct gvt_hip_light { /* scene/Light.h:46-104 flattened */
  int32_t type;
  float position[3];
  float color[3];
  float normal[3];
  float width, height;
  float pad[4];
} gvt_hip_light;

typedef struct gvt_hip_hit {
  float t;
  int32_t prim; /* -1 = miss */
  float u, v;
} gvt_hip_hit;

typedef struct gvt_hip_mesh gvt_hip_mesh;   /* one adapter instance: replaces EmbreeMeshAdapter */
typedef struct gvt_hip_queue gvt_hip_queue; /* device-resident RayVector */
typedef struct gvt_hip_top gvt_hip_top;     /* top-level instance set (accel/BVH.h) */
typedef struct gvt_hip_fb gvt_hip_fb;       /* float RGBA framebuffer (IceTComposite) */

/* ---- process / device ---- */
typedef struct gvt_hip_ctx gvt_hip_ctx;       /* a stream + scratch arenas + counters + statistics + knobs */
int gvt_hip_init(int device);                 /* select device, create the process's default context */
/* Additional contexts, e.g. one per rank when several ranks of a scheduler share a process (the reference runs several MPI ranks per
 * node): every API call runs on the calling thread's current context (NULL = the default one).  Device objects (meshes, queues,
 * framebuffers, top-level sets) may be used from any context of their device, by one context at a time. */
gvt_hip_ctx *gvt_hip_ctx_create(int device);
int gvt_hip_ctx_make_current(gvt_hip_ctx *);  /* for the calling thread; also selects the context's device (hipSetDevice) */
void gvt_hip_ctx_destroy(gvt_hip_ctx *);
int gvt_hip_set_stream(void *hip_stream);     /* issue all later work on this hipStream_t (NULL = own stream) */
int gvt_hip_synchronize(void);
const char *gvt_hip_last_error(void);

/* ---- adapter construction: EmbreeMeshAdapter::EmbreeMeshAdapter (EmbreeMeshAdapter.cpp:125-162) ----
 * Copies everything (no borrowed pointers), generates vertex normals when vnormals==NULL
 * (Mesh::generateNormals, Mesh.cpp:116-154) and builds the LBVH on the device. */
gvt_hip_mesh *gvt_hip_mesh_create(const float *verts, size_t nV, const int32_t *tris, size_t nT,
                                  const float *vnormals /* nV*3 or NULL */, const float *vcolors /* nV*3 or NULL */,
                                  const gvt_hip_material *materials, size_t nMat,
                                  const int32_t *face_mat /* nT or NULL, -1 = none */,
                                  const gvt_hip_material *mesh_mat /* Mesh::mat; NULL = Material() */);
void gvt_hip_mesh_destroy(gvt_hip_mesh *);

typedef struct gvt_hip_mesh_info {
  uint64_t n_tris, n_verts, n_nodes, n_leaves;
  float bbox_lo[3], bbox_hi[3];
  float build_ms; /* device time of the LBVH build */
  uint32_t max_leaf;
  uint32_t packet;   /* 1: the builder found the mesh packet-friendly -- coherent lists (camera rays in tile order and their shadow rays) are traversed a
                        packet of 64 per wave (k_packet); 0: one lane per ray.  The reference picks its packet width per build too (EmbreeMeshAdapter.cpp:50-74) */
  uint64_t bytes_nodes, bytes_tris;
  float sah_inner;   /* the statistic behind `packet`: sum over the inner nodes of area(node) / area(root) = inner nodes a random line through the mesh's
                        box pierces.  A surface stays at a few dozen whatever its triangle count, a volume-filling soup grows with N^(1/3) */
  float pad;
} gvt_hip_mesh_info;
int gvt_hip_mesh_get_info(const gvt_hip_mesh *, gvt_hip_mesh_info *);
/* vertex normals in use (device -> host copy), nV*3 floats */
int gvt_hip_mesh_get_normals(const gvt_hip_mesh *, float *out);

/* ---- Adapter::trace (Adapter.h:82-84; EmbreeMeshAdapter.cpp:625-660) ----
 * rays[begin,end) are traced (end==0 -> n_rays, EmbreeMeshAdapter.cpp:642) and updated in place like the
 * reference's rayList (t, and origin/direction/w/depth/type on a bounce).  rays_out receives every
 * PRIMARY/SECONDARY ray that missed and every un-occluded SHADOW ray (moved_rays); order unspecified.
 * If cap is too small: returns GVT_HIP_ERR_CAPACITY, *n_out = needed count, rays_out untouched. */
int gvt_hip_trace(gvt_hip_mesh *, gvt_hip_ray *rays, size_t n_rays, size_t begin, size_t end, gvt_hip_ray *rays_out,
                  size_t cap, size_t *n_out, const float m[16], const float minv[16], const float normi[9],
                  const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed);
/* The same call with flags.  GVT_HIP_TRACE_NO_WRITEBACK: `rays` is only read.  Both of the reference's schedulers clear the traced
 * queue right after the call (ImageTracer.h:248, DomainTracer.h:316), so the rayList's in-place update is never observed there;
 * skipping it saves a third of the call's PCIe traffic (80 B per ray back to the host). */
#define GVT_HIP_TRACE_NO_WRITEBACK 1u
int gvt_hip_trace_ex(gvt_hip_mesh *, gvt_hip_ray *rays, size_t n_rays, size_t begin, size_t end, gvt_hip_ray *rays_out,
                     size_t cap, size_t *n_out, const float m[16], const float minv[16], const float normi[9],
                     const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed, uint32_t flags);

/* ---- the Embree queries the adapter is built on (rtcIntersect / rtcOccluded,
 *      EmbreeMeshAdapter.cpp:474,375): object-space rays, t in (tnear, FLT_MAX) ---- */
int gvt_hip_intersect(gvt_hip_mesh *, const float *org /* n*3 */, const float *dir /* n*3 */, size_t n, float tnear,
                      gvt_hip_hit *hits_out);
int gvt_hip_occluded(gvt_hip_mesh *, const float *org, const float *dir, size_t n, float tnear, int32_t *occluded_out);

/* ---- device-resident RayVector (actor/Ray.h:189) ---- */
gvt_hip_queue *gvt_hip_queue_create(size_t capacity);
void gvt_hip_queue_destroy(gvt_hip_queue *);
int gvt_hip_queue_reserve(gvt_hip_queue *, size_t capacity); /* grows, keeps contents */
int gvt_hip_queue_clear(gvt_hip_queue *);
int gvt_hip_queue_size(gvt_hip_queue *, size_t *n);          /* host-side count, exact between API calls; no device access */
/* append n 80-byte rays.  flags: GVT_HIP_APPEND_DEVICE -- `rays` is device memory (default: host); GVT_HIP_APPEND_KEEP_STATE -- the rays are rays this
 * library exported (gvt_hip_queue_export, the wire image of an exchange) and bytes 64..79 carry their state: the RNG stream word and the
 * instances already crossed without a hit (known misses).  Without it the rays are FRESH, whatever those bytes hold: a GraviT host leaves them
 * uninitialised (Ray.h:106-116 never writes data[64..79]), and stack garbage read as a known-miss list would make the scheduler walk a ray
 * through an instance without tracing it. */
#define GVT_HIP_APPEND_DEVICE 1
#define GVT_HIP_APPEND_KEEP_STATE 2
int gvt_hip_queue_append_flags(gvt_hip_queue *, const gvt_hip_ray *rays, size_t n, int flags);
/* The entry point of the earlier revisions, with the meaning it had there: src_on_device != 0 -- `rays` is a wire image in device memory (what an exchange
 * delivered; this library exported it): its state bytes 64..79 are kept, = GVT_HIP_APPEND_DEVICE | GVT_HIP_APPEND_KEEP_STATE; src_on_device == 0 -- host
 * rays, taken as FRESH (= flags 0); any other value: GVT_HIP_ERR_INVALID (a revision-5 caller's flag word is refused, not misread).  A caller written
 * against the boolean keeps working; new code uses gvt_hip_queue_append_flags. */
int gvt_hip_queue_append(gvt_hip_queue *, const gvt_hip_ray *rays, size_t n, int src_on_device);
/* copy the queue out as 80-byte rays (host or device destination) */
int gvt_hip_queue_export(gvt_hip_queue *, gvt_hip_ray *dst, size_t cap, size_t *n, int dst_on_device);
/* Adapter::trace on device queues: consumes q_in (left empty, like ImageTracer.h:248), appends to q_out */
int gvt_hip_trace_queue(gvt_hip_mesh *, gvt_hip_queue *q_in, gvt_hip_queue *q_out, const float m[16], const float minv[16],
                        const float normi[9], const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed);
/* The same call, with the terminal rule of the shuffleRays that would follow (TracerBase.h:396-400) applied inside the adapter: a
 * ray the adapter would append to q_out (an un-occluded shadow ray, or a ray that leaves the mesh without a hit) but that meets no
 * instance other than from_inst is not appended: a SHADOW ray that carries colour deposits color*w into fb on the spot, any other
 * such ray is dropped -- exactly what the shuffle would have done with it.  All other rays reach q_out as before, so
 * gvt_hip_shuffle(top, q_out, from_inst, ...) afterwards completes the step.  top == NULL or fb == NULL: plain gvt_hip_trace_queue. */
int gvt_hip_trace_queue_sink(gvt_hip_mesh *, gvt_hip_queue *q, gvt_hip_queue *moved, const float m[16], const float minv[16],
                             const float normi[9], const gvt_hip_light *lights, size_t n_lights, int normal_mode, uint32_t seed,
                             gvt_hip_top *top, int from_inst, gvt_hip_fb *fb);

/* ---- gvtPerspectiveCamera::generateRays (gvtCamera.cpp:89-171, 233-312): fills q with W*H*samples^2 rays ---- */
int gvt_hip_camera_generate(gvt_hip_queue *q, const float eye[3], const float focus[3], const float up[3], float fov,
                            int width, int height, int samples, int depth, float jitter_window_size);
/* The same rays, listed tile by tile (tile = 8: 8x8 pixels, i.e. one 64-wide wavefront per tile; 0 = the reference's
 * pixel-major order).  A RayVector's order carries no meaning in the reference (TBB chunks, moved_rays under a mutex);
 * the schedulers use the tiled list because the shuffle below keeps list order and coherent packets traverse faster. */
int gvt_hip_camera_generate_tiled(gvt_hip_queue *q, const float eye[3], const float focus[3], const float up[3], float fov,
                                  int width, int height, int samples, int depth, float jitter_window_size, int tile);

/* ---- top-level instance set + shuffleRays (accel/BVH.h:61-135, actor/RayPacket.h:83-211,
 *      algorithm/TracerBase.h:325-343,392-414) ----
 * inst_lo/hi: world AABBs (api.cpp:309-312).  The instances are tested in the order the reference's
 * BVH visits its leaves (BVH.cpp:77-171), so ties resolve identically. */
gvt_hip_top *gvt_hip_top_create(const float *inst_lo, const float *inst_hi, size_t n_inst);
void gvt_hip_top_destroy(gvt_hip_top *);
int gvt_hip_top_order(const gvt_hip_top *, int32_t *order_out /* n_inst */);
/* shuffleRays(rays, from): consumes q_in (also when it fails: q_in is then cleared, its rays are lost); a ray whose next instance is i is advanced
 * (origin += dir * t * 0.95f) and appended to queues[i] -- unless keep_mask!=NULL and keep_mask[i]==0
 * (Tracer<DomainScheduler>::shuffleDropRays, DomainTracer.h:148-183); SHADOW rays that hit no further
 * instance deposit color*w into fb (fb may be NULL only if deposit is impossible, i.e. from<0 drop mode). */
int gvt_hip_shuffle(gvt_hip_top *, gvt_hip_queue *q_in, int from, gvt_hip_queue *const *queues /* n_inst */,
                    const uint8_t *keep_mask, gvt_hip_fb *fb);
/* sizes of n queues in one round trip */
int gvt_hip_queue_sizes(gvt_hip_queue *const *queues, size_t n, uint64_t *sizes_out);

/* ---- Tracer<ImageScheduler>::operator() (algorithm/ImageTracer.h:127-269) for one rank, natively: clearBuffer, generateRays,
 *      FilterRaysLocally, then until every queue is empty: pick the fullest queue (first strictly largest, :159-173), adapter->trace,
 *      shuffleRays.  meshes[i] / m / minv / normi are per instance (adapters may repeat: adapterCache, :184-233).  The caller
 *      owns the queues (n_inst of them), q_moved and fb; on return the queues are empty and fb holds the frame.  q_cam is not used
 *      any more (generateRays is fused into FilterRaysLocally, gvt_hip_camera_filter) and may be NULL. ---- */
typedef struct gvt_hip_camera {
  float eye[3], focus[3], up[3];
  float fov;
  int32_t width, height, samples, depth;
  float jitter_window_size;
} gvt_hip_camera;
/* generateRays (gvtCamera.cpp:233-312) + FilterRaysLocally (ImageTracer.h:111-125 / shuffleDropRays, DomainTracer.h:148-183) in one
 * step: every camera ray is appended to the queue of the instance it enters first (advanced like gvt_hip_shuffle does), without the
 * W*H*samples^2 list ever being written to memory.  tile as in gvt_hip_camera_generate_tiled. */
int gvt_hip_camera_filter(gvt_hip_top *, const gvt_hip_camera *cam, int tile, gvt_hip_queue *const *queues, const uint8_t *keep_mask);
int gvt_hip_image_frame(gvt_hip_top *, gvt_hip_mesh *const *meshes, const float *m /* n_inst*16 */, const float *minv, const float *normi /* n_inst*9 */,
                        size_t n_inst, const gvt_hip_light *lights, size_t n_lights, int normal_mode, const gvt_hip_camera *cam,
                        gvt_hip_queue *const *queues, gvt_hip_queue *q_cam, gvt_hip_queue *q_moved, gvt_hip_fb *fb, uint64_t *adapter_calls);

/* ---- the schedulers as native loops (csrc/domain.hip): Tracer<ImageScheduler>::operator() (ImageTracer.h:127-269) and
 *      Tracer<DomainScheduler>::operator() with SendRays (DomainTracer.h:185-496); asynchronous variant tracer/Domain/DomainTracer.cpp:109-192.
 *      Every round, all non-empty local queues go through ONE merged launch chain (closest hit, shade, any hit) and one shuffle, with
 *      one host synchronisation per round; the Domain scheduler's ray exchange is RCCL point-to-point on its own stream (announce =
 *      SendRays' count exchange + the termination vote, then the payload in the reference's wire format
 *      [int32 queueId][int32 nRays][80-byte Ray x nRays], :441-455). ---- */
typedef struct gvt_hip_comm gvt_hip_comm;     /* one rank's endpoint of the ray exchange */
typedef struct gvt_hip_hub gvt_hip_hub;       /* rendezvous of in-process ranks */
typedef struct gvt_hip_tracer gvt_hip_tracer; /* a Tracer<...> object for one rank */
/* RCCL transport: rank 0 calls gvt_hip_comm_unique_id, the 128 bytes reach the other ranks out of band (e.g. a torch.distributed or
 * MPI broadcast), every rank calls gvt_hip_comm_create on its own device.  Collective.  The library is resolved at first use: GVT_HIP_RCCL_LIB
 * (a path: a particular build, or a stand-in with the same entry points -- tests/fake_rccl) if set, else librccl.so.1 as the process or the system has it. */
int gvt_hip_comm_unique_id(unsigned char id[128]);
gvt_hip_comm *gvt_hip_comm_create(const unsigned char id[128], int rank, int world);
/* In-process transport: the ranks are threads of one process (one context each, gvt_hip_ctx_create) sharing a device; transfers are
 * device-to-device copies ordered by events.  Same protocol, no RCCL: several ranks per node on one GPU, and the way the Domain
 * scheduler's multi-rank control flow is exercised on a one-GPU machine. */
gvt_hip_hub *gvt_hip_hub_create(int world);
void gvt_hip_hub_abort(gvt_hip_hub *);        /* a rank failed: wake the ranks blocked in an exchange (they return an error) */
void gvt_hip_hub_destroy(gvt_hip_hub *);
gvt_hip_comm *gvt_hip_comm_create_local(gvt_hip_hub *, int rank);
void gvt_hip_comm_destroy(gvt_hip_comm *);
/* loop-back check on this rank: grouped send-to-self / receive-from-self of `bytes` bytes (+ an in-place ncclReduce under RCCL) through the
 * very calls the frame loop makes; 0 = the payload came back intact */
int gvt_hip_comm_selftest(gvt_hip_comm *, size_t bytes);
int gvt_hip_comm_rank(const gvt_hip_comm *);
/* ranks the transport itself counts (ncclCommCount; the hub's world): gvt_hip_comm_create fails unless it equals `world` */
int gvt_hip_comm_count(const gvt_hip_comm *);
/* compute units reserved for the communicator's own stream (knob "comm_cus" of the creating context at gvt_hip_comm_create[_local]; 0: none) */
int gvt_hip_comm_reserved_cus(const gvt_hip_comm *);
/* deadline of every blocking point of an exchange, in milliseconds (default: GVT_HIP_EXCHANGE_TIMEOUT_MS or 20000).  A rank that waits
 * longer returns GVT_HIP_ERR_TIMEOUT with its last announce in gvt_hip_last_error(), aborts the communicator (ncclCommAbort / hub
 * abort) and must not use it again: the process reports and exits non-zero. */
int gvt_hip_comm_set_deadline_ms(gvt_hip_comm *, int ms);
int gvt_hip_comm_world(const gvt_hip_comm *);

/* The tracer copies matrices, lights and camera, creates its own per-instance queues, and borrows top, meshes and fb. */
gvt_hip_tracer *gvt_hip_tracer_create(gvt_hip_top *, gvt_hip_mesh *const *meshes, const float *m /* n_inst*16 */, const float *minv,
                                      const float *normi /* n_inst*9 */, size_t n_inst, const gvt_hip_light *lights, size_t n_lights,
                                      int normal_mode, const gvt_hip_camera *cam, gvt_hip_fb *fb);
void gvt_hip_tracer_destroy(gvt_hip_tracer *);
int gvt_hip_tracer_set_camera(gvt_hip_tracer *, const gvt_hip_camera *cam);
/* mpiInstanceMap (DomainTracer.h:115-144): owner[i] = rank holding instance i.  comm == NULL (or never called): one rank, Image scheduler. */
int gvt_hip_tracer_set_domains(gvt_hip_tracer *, const int32_t *owner /* n_inst */, gvt_hip_comm *comm);
#define GVT_HIP_FRAME_BSP 1          /* Domain: trace until the local queues are dry, then exchange (Tracer<DomainScheduler>); default:
                                        one local chain per exchange, the payload moving while the next chain runs (asynchronous tracer) */
#define GVT_HIP_FRAME_NO_COMPOSITE 2 /* leave the per-rank framebuffers un-reduced */
#define GVT_HIP_FRAME_IMAGE 8        /* several ranks, Image scheduler (ImageTracer.h:111-125): the scene is replicated (every instance's mesh on every
                                        rank), each rank traces its contiguous portion of the camera's rays to the end, then the composite */
#define GVT_HIP_FRAME_FULL_REDUCE 4  /* composite by a sum-reduce of whole frames (ncclReduce) instead of each rank's written rectangle */
typedef struct gvt_hip_frame_stats {
  uint64_t rounds;       /* exchanges (Domain) */
  uint64_t chains;       /* merged launch chains on this rank */
  uint64_t host_syncs;   /* host synchronisations of the frame */
  uint64_t rays_sent;    /* rays this rank sent to other ranks */
  uint64_t rays_closest; /* rays this rank pushed through the closest-hit kernel */
  uint64_t rays_any;     /* ... through the any-hit kernel */
  uint64_t packets_bailed; /* 64-ray packets the packet traversal handed to the one-lane-per-ray kernels (stack / step budget) */
  uint64_t bytes_sent;   /* payload bytes this rank sent to other ranks (wire format, DomainTracer.h:441-455), composite excluded */
  /* where a multi-rank frame's time goes (HIP events on the compute / communication streams, host clock for the waits); 0 on one rank, and 0
   * unless gvt_hip_set_option("frame_timing", 1): the breakdown costs five more event calls per exchange */
  double ms_chain;       /* local launch chains (incl. unpacking what arrived) */
  double ms_announce;    /* announce exchanges: group of sends / receives + the report's device-to-host copy */
  double ms_payload;     /* payload exchanges, posting to arrival (overlaps the next chain) */
  double ms_composite;   /* framebuffer composite on rank 0 */
  double ms_host_wait;   /* host time blocked in the exchanges' bounded waits */
  uint64_t exchanges;    /* transport groups (ncclGroupStart .. End) this rank issued in the frame, composite included: one per tick when every pair's
                            payload rode inside the announce ("inline_kb"), two where a payload exchange followed */
  uint64_t rays_inline;  /* of rays_sent: rays that travelled inside an announce */
} gvt_hip_frame_stats;
/* One frame: clearBuffer, generateRays + FilterRaysLocally / shuffleDropRays, rounds until every queue of every rank is empty, then
 * (Domain) the composite: the sum of the ranks' float framebuffers on rank 0 (IceTComposite.cpp:84-101).  Collective under a comm. */
int gvt_hip_tracer_frame(gvt_hip_tracer *, int flags, gvt_hip_frame_stats *stats);

/* ---- framebuffer: IceTComposite (composite/IceTComposite.cpp:79-157) ---- */
gvt_hip_fb *gvt_hip_fb_create(int width, int height);
void gvt_hip_fb_destroy(gvt_hip_fb *);
int gvt_hip_fb_clear(gvt_hip_fb *);                              /* reset(), :79-82 */
int gvt_hip_fb_download(gvt_hip_fb *, float *rgba /* W*H*4 */, int clamp); /* clamp: c>1 -> 1 (localAdd :111-117) */
void *gvt_hip_fb_device_ptr(gvt_hip_fb *);                       /* float[W*H*4], un-clamped sums: for the RCCL reduce */
int gvt_hip_fb_write_ppm_bytes(gvt_hip_fb *, unsigned char *rgb /* W*H*3, rows flipped, (uchar)(c*255) :119-157 */);

/* ---- measurement ---- */
typedef struct gvt_hip_stats {
  uint64_t rays_closest; /* rays pushed through the closest-hit kernel */
  uint64_t rays_any;     /* rays pushed through the any-hit kernel */
  uint64_t rays_shaded, rays_forwarded, trace_calls;
  /* HIP-event time per kernel class, ms, accumulated while profiling is enabled */
  double ms_closest, ms_any, ms_shade, ms_convert, ms_shuffle, ms_camera, ms_build;
  uint64_t launches_closest, launches_any;
  double ms_sort; /* ray sorting inside the adapter */
  double ms_long; /* k_long_closest: parked long rays, a wave per ray (not part of ms_closest) */
} gvt_hip_stats;
int gvt_hip_profile(int enable);           /* 1: bracket every kernel with HIP events on the launch stream; 2: the traversal kernels only; 3 / 4: the
                                            * closest-hit / any-hit launches only (an event pair costs a few microseconds of the stream); 0: off */
int gvt_hip_stats_read(gvt_hip_stats *);   /* synchronises */
int gvt_hip_stats_reset(void);
/* diagnostic, not on the hot path: per-ray visit counts of the closest-hit traversal over n object-space rays:
 * counts[3*j + 0..2] = inner-node visits / leaf visits / triangle tests of ray j. */
int gvt_hip_visit_stats(gvt_hip_mesh *, const float *org, const float *dir, size_t n, float tnear, uint32_t *counts);
/* diagnostic, not on the hot path: the number of nodes a `width`-wide collapse of the mesh's tree (width 2..8, the collapse rule of the
 * 4-wide layout the traversal uses) would make each ray visit; *n_wide_nodes (optional) = nodes of that collapse. */
int gvt_hip_wide_visit_stats(gvt_hip_mesh *, const float *org, const float *dir, size_t n, float tnear, int width, uint32_t *counts /* n */,
                             uint64_t *n_wide_nodes);
/* measurement: the traversal layout as the kernels walk it -- (bytes_nodes / 64 - n_nodes) compressed 4-wide nodes of 64 bytes {origin.xyz, step.x | qlo.x[4], qhi.x[4],
 * qlo.y[4], qhi.y[4] | qlo.z[4], qhi.z[4], ref0, ref1 | ref2, ref3, step.y, step.z} (child box plane = origin + q * step; ref >= 0: node, < 0: leaf, ~ref =
 * first slot << 3 | triangles) and n_tris triangle slots of 64 bytes {v0, primID | e1 = v0 - v1, . | e2 = v2 - v0, . | .} in leaf order -- for the SIMD CPU
 * baseline of bench.py (oracle/simd_baseline.c) */
int gvt_hip_mesh_download_wide(gvt_hip_mesh *, void *nodes4, size_t n_nodes4, void *slots, size_t n_slots);
/* ... and the CLUSTER layout of the same nodes (built on first use; the Domain scheduler's small rounds, k_finish, walk it -- knob finish_clusters): a permutation of
 * nodes4 in which every node of an even level is followed by its inner children, the references of the odd-level nodes to inner nodes being (slot << 4) | mask
 * of the grandchild's inner children; *root_entry = the root's (slot 0).  rc 0 with *root_entry = -1: the mesh has no such layout (too deep or too large), the
 * traversals walk nodes4 */
int gvt_hip_mesh_download_clusters(gvt_hip_mesh *, void *nodes4c, size_t n_nodes4, int32_t *root_entry);
/* diagnostics behind tools/wide_dp.py (is a cost-optimal wide collapse of the tree worth building?): the binary LBVH as built -- n_nodes x 64 bytes,
 * per node {c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y | c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y | c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z | child0, child1 (int32; < 0: leaf), -, -} --
 * and the visits of a collapse the CALLER chose: marks[k] = 1 where binary node k is the root of a wide node, counts[j] = marked nodes ray j visits */
int gvt_hip_mesh_download_nodes(gvt_hip_mesh *, void *out, size_t n_nodes);
/* ... and back: a tree of the caller's over the SAME leaves (same node count, root = node 0) for the visit-count diagnostics only -- the 4-wide layout the
 * product kernels traverse is not rebuilt from it */
int gvt_hip_mesh_upload_nodes(gvt_hip_mesh *, const void *in, size_t n_nodes);
int gvt_hip_marked_visit_stats(gvt_hip_mesh *, const float *org, const float *dir, size_t n, float tnear, const unsigned char *marks /* n_nodes */, uint32_t *counts /* n */);
/* diagnostic, not on the hot path: include/gvt_math.h evaluated on the device, element-wise over n floats -- kind 0 gvt_sinf(x),
 * 1 gvt_cosf(x), 2 (float)gvt_acos(sqrt(1.0 - x)) -- so that a test can compare the device's bits with the host's for the
 * functions the bounce path (EmbreeMeshAdapter.cpp:289-318) is built on. */
int gvt_hip_math_probe(int kind, const float *in, size_t n, float *out);
/* Knobs of the library (gvt_internal.h `struct Knobs` lists them with their defaults; ("defaults", 0) restores all of them).  Results never
 * depend on them -- except "skip_known" = 1, an opt-in approximation of the shuffle rule (below).  The shipped surface, 23 knobs:
 *   behaviour    "skip_known"   0 (default): the reference's hop-by-hop shuffleRays, ray for ray -- 1: the known-miss shortcut (a ray is not traced / sent again into an
 *                               instance it has already crossed without a hit on the same straight segment).  The shortcut saves the hand-back hops between
 *                               overlapping boxes but is an APPROXIMATION: where a re-trace from the advanced origin would flip an edge-grazing triangle test,
 *                               the reference finds a hit the shortcut skips (measured: 2 of 147,456 pixels of the hall in 8 slabs; none on the soup tiles)
 *                "term_sink"    1: gvt_hip_trace_queue_sink applies shuffleRays' terminal rule inside the kernels -- 0: every moved ray goes through the shuffle
 *                "sort_rays"    1: Morton-sort a list before traversal (pays on incoherent lists; off)
 *                "frame_timing" 1: fill gvt_hip_frame_stats' per-phase milliseconds (five more event calls per exchange)
 *                "packet"       closest hits of coherent lists (camera rays in tile order) a packet of 64 rays per wave: 0 never, 1 (default) on meshes the builder
 *                               found packet-friendly (gvt_hip_mesh_info::packet), in launches of at least "packet_min_rays" rays, 2 always (and shadow rays too)
 *                "shadow_order" 1 (default): a single-mesh round with one light lists its shadow rays by how many node steps the primaries of their 64-ray tile
 *                               took, the longest first (a shadow ray's step count follows its primary's), so that the any-hit launch's drain is left to
 *                               short rays -- in launches of at least "shadow_order_min_rays" rays; 0: in arrival order.  Results do not depend on it
 *   build time   "leaf_max"     triangles per leaf of meshes created afterwards (1..4, default 2)
 *                "packet_sah_max"  meshes created afterwards are packet-friendly when gvt_hip_mesh_info::sah_inner is at most this (default 128)
 *   budgets      "long_steps" / "long_min_rays" / "long_auto"  closest hit: node steps after which a ray is parked for a whole wave (0: never), launches it
 *                               applies to, and whether gvt_hip_tracer_frame raises it from frame to frame on scenes that park more than 0.3 % of their rays
 *                "small_rays" / "finish_rays"    rounds of at most so many rays: a wave per ray / the whole round in one launch (k_finish)
 *                "finish_auto"                   1: on one rank with several instances gvt_hip_tracer_frame times a few frames with and without k_finish and keeps
 *                                                the faster (a hop that is a long traversal of its own is better served by per-hop rounds)
 *                "round_room_mb"                 memory a round's worst-case reservation may add before it falls back to exact growth
 *                "payload_overlap_kb"            Domain scheduler: payloads of at least this size move on the communicator's own stream beside the next chain
 *                "inline_kb"                     Domain scheduler: a pair's payload of at most this many KiB per tick travels inside the announce message (one exchange
 *                                                per tick; default 16; the same on every rank; 0: the two-step exchange of SendRays, counts then rays)
 *                "comm_cus"                      Domain scheduler: compute units reserved for the communicator's own stream (CU mask; the persistent traversal grids
 *                                                are sized for the rest); set before gvt_hip_comm_create; default 0
 *                "spec_ticks"                    Domain scheduler, asynchronous ticks: 1 (default) = the next tick's small round and report are enqueued behind the
 *                                                current exchange before its result is read (the device voids them when the result needs the host); 0 = never
 *                "hop_local"                     Domain / Image scheduler, merged launches over several instances: a ray that leaves its instance without a hit and has another
 *                                                instance of THIS rank ahead goes on there inside the traversal launch (shuffleRays' rule applied by the lane) instead of waiting
 *                                                for the next round -- 0 never, 2 always, 3 early: in the closest-hit launch only and only while it still has rays to hand out (a ray that
 *                                                goes on during the launch's drain stretches its tail), 1 (default) per tracer: never / early / always timed like finish_auto on one rank, on several
 *                                                by the meshes' kind (surfaces always, volume-filling soups never).  Results never depend on it
 *                "finish_clusters"               small rounds of several instances (k_finish, a wave per ray): 1 (default) = walk the cluster layout of the 4-wide nodes
 *                                                (two tree levels per memory round trip; built per mesh when such a tracer is created, + 64 bytes per node); 0 = the plain nodes
 *                "comm_stream"                   Domain scheduler: 1 = every exchange of a frame on the communicator's own stream, ordered against the compute stream
 *                                                by events (also GVT_HIP_COMM_STREAM in the environment); default 0: on the compute stream, large payloads beside it
 *                "abi_lanes" / "abi_chunk"     gvt_hip_trace on a host RayVector: pipeline lanes (0: one shot), rays per chunk
 *   test hook    "inject_fail_tick"
 * Everything else -- the tuned constants of the kernels (refill / phase thresholds, grid sizes, drain sharing ...) and the variants that were
 * measured and lost ("trav_kernel" = 0, "wide4" = 0, "coop_fetch", "fused", "packet", "quad", merged kernels for one queue, compacted shadow
 * slots, non-lean frames, camera rays in generateRays' pixel-major order ("camera_tile" = 0) ...: EXPERIMENTS.md) -- can be moved only in the experiments build of the library (libgvt_hip_exp.so,
 * -DGVT_EXPERIMENTS), where the knob sweeps run; the shipped library answers GVT_HIP_ERR_INVALID when one of them is switched away from its default. */
int gvt_hip_set_option(const char *name, int value);
/* 1 in the experiments build (every variant behind its knob), 0 in the shipped library */
int gvt_hip_is_experiments_build(void);
/* diagnostic: the calling thread's context counter words (32) after a stream synchronisation; [3] = rays the last closest-hit launch
   parked for k_long_closest, [8] = traversal overflow flags */
int gvt_hip_counters_peek(uint32_t out[32]);

#ifdef __cplusplus
}
#endif
#endif
