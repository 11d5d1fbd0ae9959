// sched.hip -- the steps either side of Adapter::trace, kept on the device so that ray queues never
// cross PCIe between adapter calls:
//   k_camera        gvtPerspectiveCamera::generateRays       (data/scene/gvtCamera.cpp:233-312)
//   k_top_classify  BVH::intersect + RayPacketIntersection   (data/accel/BVH.h:61-135, actor/RayPacket.h:83-211)
//   k_top_scatter   AbstractTrace::shuffleRays, mesh branch  (algorithm/TracerBase.h:392-400) and
//                   Tracer<DomainScheduler>::shuffleDropRays (algorithm/DomainTracer.h:148-183)
//   framebuffer     IceTComposite::localAdd / reset / write  (composite/IceTComposite.cpp:79-157)
#include <algorithm>
#include <cmath>

#include "gvt_internal.h"

namespace {

struct CamArgs {
  V3 eye, u, v, w;
  float vert, horz, wmult, hmult, half_sample, offset, contri;
  int W, H, samples, depth;
  int tile; // 0: rays in the reference's order (pixel-major); 8: pixels enumerated in 8x8 tiles (one wave of rays = one tile)
  unsigned first; // the kernels' ray i is ray `first + i` of the generated list (a rank's portion, ImageTracer.h:112-120)
  // Domain scheduler, shuffleDropRays: a rank keeps only the camera rays whose first domain is one of ITS instances, and those can only
  // come from the pixels its instances' boxes project onto.  rect_on: the list is enumerated over that rectangle only -- 8x8 tiles,
  // row-major over the rectangle's tiles (positions of tiles that reach beyond x1 / y1 hold no ray) -- instead of the whole film.
  int rect_on, rx0, ry0, rx1, ry1, rtpr; // rectangle in pixels (x0, y0 multiples of 8), tiles per row
};

// position in the generated list -> pixel.  tile == 8: the W8 x H8 part of the image that whole 8x8 tiles cover comes first, tile
// by tile, then the right-hand strip and the bottom strip in scanline order -- a bijection on [0, W*H).
__device__ inline unsigned camera_slot_pixel(unsigned slot, int W, int H, int tile) {
  if (tile != 8) return slot;
  const unsigned W8 = (unsigned)W & ~7u, H8 = (unsigned)H & ~7u, n_full = W8 * H8;
  if (slot < n_full) {
    const unsigned t = slot >> 6, k = slot & 63u, tpr = W8 >> 3;
    return ((t / tpr) * 8u + (k >> 3)) * (unsigned)W + (t % tpr) * 8u + (k & 7u);
  }
  unsigned r = slot - n_full;
  const unsigned Wr = (unsigned)W - W8;
  if (r < Wr * H8) return (r / Wr) * (unsigned)W + W8 + r % Wr;
  r -= Wr * H8;
  return (H8 + r / (unsigned)W) * (unsigned)W + r % (unsigned)W;
}

// rect_on: pixel of list position `slot`, or 0xffffffff where the rectangle's last tile column / row reaches beyond it
__device__ inline unsigned camera_rect_pixel(const CamArgs &A, unsigned slot) {
  const unsigned t = slot >> 6, k = slot & 63u;
  const unsigned px = (unsigned)A.rx0 + (t % (unsigned)A.rtpr) * 8u + (k & 7u), py = (unsigned)A.ry0 + (t / (unsigned)A.rtpr) * 8u + (k >> 3);
  return (px < (unsigned)A.rx1 && py < (unsigned)A.ry1) ? py * (unsigned)A.W + px : 0xffffffffu;
}
__device__ inline bool camera_slot_valid(const CamArgs &A, unsigned long long ridx) {
  return !A.rect_on || camera_rect_pixel(A, (unsigned)(ridx / (unsigned)(A.samples * A.samples))) != 0xffffffffu;
}
__device__ inline RayRec camera_ray(const CamArgs &A, unsigned long long ridx) {
  const unsigned samples2 = (unsigned)(A.samples * A.samples);
  unsigned pix = A.rect_on ? camera_rect_pixel(A, (unsigned)(ridx / samples2)) : camera_slot_pixel((unsigned)(ridx / samples2), A.W, A.H, A.tile);
  if (pix == 0xffffffffu) pix = 0u; // (a position without a ray: the callers skip it, camera_slot_valid)
  const unsigned sub = (unsigned)(ridx % samples2);
  const int k = (int)(sub / (unsigned)A.samples), ww = (int)(sub % (unsigned)A.samples);
  const int i = (int)(pix % (unsigned)A.W), j = (int)(pix / (unsigned)A.W);
  // float(i)*wmult - 1.0 : the double subtraction of the reference is exact here, so the float form has the same bits
  const float x0 = (float)i * A.wmult - 1.0f, y0 = (float)j * A.hmult - 1.0f;
  float x = x0 + ((float)ww - A.half_sample) * A.offset;
  x *= A.horz;
  float y = y0 + ((float)k - A.half_sample) * A.offset;
  y *= A.vert;
  V3 d;
  d.x = A.u.x * x + A.v.x * y + A.w.x;
  d.y = A.u.y * x + A.v.y * y + A.w.y;
  d.z = A.u.z * x + A.v.z * y + A.w.z;
  RayRec r;
  r.o = A.eye; r.t_min = GVT_RAY_EPSILON;
  r.d = norm3(d); r.t_max = GVT_FLT_MAX;
  r.c = mk3(0.f, 0.f, 0.f); r.t = GVT_FLT_MAX;
  r.id = (int)pix; r.depth = A.depth; r.w = A.contri; r.type = 0;
  r.rng = camera_stream_word((unsigned long long)pix * samples2 + sub);
  r.km[0] = 0u; r.km[1] = 0u; r.km[2] = 0u;
  return r;
}

__global__ __launch_bounds__(256) void k_camera(CamArgs A, RayPlanes q, unsigned long long n) {
  const unsigned long long ridx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ridx >= n) return;
  store_ray(q, ridx, camera_ray(A, ridx));
}

// where the shuffle kernels take their rays from: a queue, or the camera itself (generateRays fused into FilterRaysLocally:
// the W*H*samples^2 list is never written to memory, each kernel regenerates ray i from its index)
struct RaySrc {
  RayPlanes q;
  CamArgs cam;
  int from_cam;
};

// Destination counting and slot allocation are aggregated twice before they reach a global counter: per wave
// (__ballot over equal destinations) and per 1024-thread block (LDS counters), because one global counter word
// sustains only ~90 atomics/us chip-wide (2 M rays: 32 K wave-level atomics on one word cost ~0.35 ms).
#define TOP_BLOCK 1024
// Scan-ordered (deterministic, order-preserving) slots up to this many destinations: one scan block per destination and
// n_dest x 16 LDS words per scatter block.  Beyond it: LDS-aggregated atomics (arrival order) -- since every ray carries its RNG
// stream, queue order no longer influences any result, only the coherence of the next traversal launch.
#define GVT_TOP_ORDERED_MAX 256

// n_dev (optional): the ray count in device memory (a list filled by the kernels just before, no host round trip; n is then only
// the bound the grid was sized for).  from_arr (optional): the source instance per ray (lists that mix rays of several instances).
__global__ __launch_bounds__(TOP_BLOCK) void k_top_classify(RaySrc S, unsigned n, TopDev top,
                                                            int n_inst, int from, int *__restrict__ next_out, float *__restrict__ t_out,
                                                            unsigned *__restrict__ hist, int use_lds, unsigned *__restrict__ blk_cnt,
                                                            const unsigned *__restrict__ n_dev = nullptr, const int *__restrict__ from_arr = nullptr, int skip_known = 0) {
  extern __shared__ unsigned sh_cnt[];
  if (n_dev) n = min(n, *n_dev);
  if (use_lds) {
    for (int d = threadIdx.x; d < n_inst; d += TOP_BLOCK) sh_cnt[d] = 0u;
    __syncthreads();
  }
  const unsigned i = blockIdx.x * TOP_BLOCK + threadIdx.x;
  int next = -1;
  if (i < n) {
    float ret_t;
    float4 a, b;
    if (S.from_cam) { const RayRec r = camera_ray(S.cam, (unsigned long long)S.cam.first + i); a = make_float4(r.o.x, r.o.y, r.o.z, r.t_min); b = make_float4(r.d.x, r.d.y, r.d.z, r.t_max); }
    else { a = S.q.p0[i]; b = S.q.p1[i]; }
    const int fr = from_arr ? from_arr[i] : from;
    if (S.from_cam && !camera_slot_valid(S.cam, (unsigned long long)S.cam.first + i)) { next = -1; ret_t = GVT_FLT_MAX; }
    else if (skip_known && !S.from_cam && S.q.p5 && fr >= 0) {
      // the known-miss shortcut (gvt_device.h): `fr` joins the ray's list; a choice that is on the list is walked through here, the
      // advanced origin left in the source list (consumed by the scatter that follows) and t_out = -1 says "already advanced"
      uint32_t km[3] = { S.q.p5[3 * (size_t)i], S.q.p5[3 * (size_t)i + 1], S.q.p5[3 * (size_t)i + 2] };
      bool walked;
      next = shuffle_walk(a, b, top, fr, km, ret_t, walked);
      if (walked) { S.q.p0[i] = a; ret_t = -1.f; }
      S.q.p5[3 * (size_t)i] = km[0]; S.q.p5[3 * (size_t)i + 1] = km[1]; S.q.p5[3 * (size_t)i + 2] = km[2];
    } else {
      next = top_nearest(a, b, top, fr, ret_t);
    }
    next_out[i] = next;
    t_out[i] = ret_t;
  }
  unsigned long long todo = ballot64(next >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int d = __shfl(next, leader);
    const unsigned long long m = ballot64(next == d);
    if ((int)lane_id() == leader) {
      if (use_lds) atomicAdd(&sh_cnt[d], (unsigned)__popcll(m)); else atomicAdd(&hist[d], (unsigned)__popcll(m));
    }
    todo &= ~m;
  }
  if (use_lds) {
    __syncthreads();
    for (int d = threadIdx.x; d < n_inst; d += TOP_BLOCK) {
      if (hist && sh_cnt[d]) atomicAdd(&hist[d], sh_cnt[d]);
      if (blk_cnt) blk_cnt[(size_t)d * gridDim.x + blockIdx.x] = sh_cnt[d]; // ordered mode: this block's rays per destination
    }
  }
}

// Ordered mode (few destinations): exclusive scan of the per-block counts of one destination, offset by the queue's fill,
// so that the scatter can place every ray at a slot that depends only on its index in the input list -- queues keep the order
// of the list they were filled from (camera rays stay in pixel order; no sort is needed in front of the traversal) and the
// result of a shuffle is deterministic.  One block per destination.
__global__ __launch_bounds__(TOP_BLOCK) void k_top_scan(unsigned *__restrict__ blk_cnt, unsigned n_blk, const QueueDesc *__restrict__ queues,
                                                        unsigned *__restrict__ totals) {
  __shared__ unsigned sh_w[TOP_BLOCK / 64];
  __shared__ unsigned sh_run;
  const int d = blockIdx.x;
  unsigned *row = blk_cnt + (size_t)d * n_blk;
  __shared__ unsigned sh_start;
  if (threadIdx.x == 0) { sh_run = *queues[d].count; sh_start = sh_run; }
  __syncthreads();
  for (unsigned b0 = 0; b0 < n_blk; b0 += TOP_BLOCK) {
    const unsigned b = b0 + threadIdx.x;
    const unsigned v = b < n_blk ? row[b] : 0u;
    unsigned incl = v;
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if ((int)lane_id() >= o) incl += u; }
    if (lane_id() == 63) sh_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (unsigned w = 0; w < (threadIdx.x >> 6); w++) woff += sh_w[w];
    const unsigned run = sh_run;
    if (b < n_blk) row[b] = run + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == TOP_BLOCK - 1) sh_run = run + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (queues[d].keep) *queues[d].count = sh_run;
    if (totals) totals[d] = sh_run - sh_start;
  }
}

__global__ __launch_bounds__(TOP_BLOCK) void k_top_scatter(RaySrc S, unsigned n, const int *__restrict__ next_in, const float *__restrict__ t_in,
                                                           const QueueDesc *__restrict__ queues, int n_inst, float *__restrict__ fb, unsigned n_pix,
                                                           int use_lds, const unsigned *__restrict__ blk_base, const unsigned *__restrict__ n_dev = nullptr,
                                                           unsigned *__restrict__ overflow = nullptr) {
  extern __shared__ unsigned sh[]; // [0,n_inst): rays of this block per destination, [n_inst,2n_inst): their base slot
  if (n_dev) n = min(n, *n_dev);
  unsigned *sh_cnt = sh, *sh_base = sh + n_inst; // ordered mode: sh[w * n_inst + d] = rays of wave w for destination d
  if (blk_base) {
    for (int k = threadIdx.x; k < n_inst * (TOP_BLOCK / 64); k += TOP_BLOCK) sh[k] = 0u;
    __syncthreads();
  } else if (use_lds) {
    for (int d = threadIdx.x; d < n_inst; d += TOP_BLOCK) sh_cnt[d] = 0u;
    __syncthreads();
  }
  const unsigned i = blockIdx.x * TOP_BLOCK + threadIdx.x;
  int next = -1;
  RayRec r;
  if (i < n) {
    next = next_in[i];
    r = S.from_cam ? camera_ray(S.cam, (unsigned long long)S.cam.first + i) : load_ray(S.q, i);
    if (next >= 0) {
      const float t_adv = t_in[i];
      if (t_adv >= 0.f) r.o = add3(r.o, scl3(r.d, t_adv * 0.95f)); // TracerBase.h:393 (< 0: k_top_classify has walked the ray through known misses)
    } else if (fb && r.type == 1 && len3(r.c) > 0.f) { // TracerBase.h:396-400 -> localAdd
      if ((unsigned)r.id < n_pix) {
        const V3 c = scl3(r.c, r.w);
        float *px = fb + (size_t)4 * (unsigned)r.id;
        atomicAdd(px + 0, c.x); atomicAdd(px + 1, c.y); atomicAdd(px + 2, c.z); atomicAdd(px + 3, 1.f);
      }
    }
    if (next >= 0 && !queues[next].keep) next = -1; // shuffleDropRays: not this rank's domain
  }
  unsigned local = 0; // use_lds: offset inside the block's share of the destination; else: final slot
  unsigned long long todo = ballot64(next >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int d = __shfl(next, leader);
    const unsigned long long m = ballot64(next == d);
    unsigned base = 0;
    if (blk_base) {
      if ((int)lane_id() == leader) sh[(threadIdx.x >> 6) * n_inst + d] = (unsigned)__popcll(m);
    } else {
      if ((int)lane_id() == leader) base = use_lds ? atomicAdd(&sh_cnt[d], (unsigned)__popcll(m)) : atomicAdd(queues[d].count, (unsigned)__popcll(m));
      base = __shfl(base, leader);
    }
    if (next == d) local = base + lanes_below(m);
    todo &= ~m;
  }
  if (blk_base) {
    __syncthreads();
    if (next >= 0) {
      for (unsigned w = 0; w < (threadIdx.x >> 6); w++) local += sh[w * n_inst + next];
      local += blk_base[(size_t)next * gridDim.x + blockIdx.x];
    }
  } else if (use_lds) {
    __syncthreads();
    for (int d = threadIdx.x; d < n_inst; d += TOP_BLOCK)
      if (sh_cnt[d]) sh_base[d] = atomicAdd(queues[d].count, sh_cnt[d]);
    __syncthreads();
    if (next >= 0) local += sh_base[next];
  }
  if (next >= 0) {
    const QueueDesc Q = queues[next];
    if (local < Q.cap) store_ray(make_planes(Q.planes, Q.cap), local, r);
    else if (overflow) atomicOr(overflow, 1u); // a destination without room: reported by the caller, never silent
  }
}

// ---- ONE instance on ONE rank (the one-domain frame): clearBuffer + generateRays + FilterRaysLocally + the counter resets the
// round's launch chain starts from, in two launches and without the classify / scatter hand-over arrays.  k_cam1_count: every thread
// generates its ray, tests it against the instance box and zeroes its pixel of the framebuffer; a block leaves its number of
// entering rays.  k_cam1_scatter: a block's first slot is the sum of the counts in front of it (a few thousand words from L2), its
// rays are generated again and stored in list order -- the queue keeps the camera's (tile) order exactly as the three-kernel
// shuffle did -- and the last block publishes the total and performs k_wave_pass_begin's pass-0 resets.
__global__ __launch_bounds__(TOP_BLOCK) void k_cam1_count(CamArgs A, unsigned n, TopDev top, unsigned *__restrict__ blk_cnt, float4 *__restrict__ fb, unsigned n_pix,
                                                          unsigned *__restrict__ c, unsigned *__restrict__ ovf) {
  __shared__ unsigned sh_w[TOP_BLOCK / 64];
  const unsigned i = blockIdx.x * TOP_BLOCK + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const RayRec r = camera_ray(A, (unsigned long long)i);
    float ret_t;
    hit = camera_slot_valid(A, (unsigned long long)i) && top_nearest(make_float4(r.o.x, r.o.y, r.o.z, r.t_min), make_float4(r.d.x, r.d.y, r.d.z, r.t_max), top, -1, ret_t) >= 0;
    if (fb && i % (unsigned)(A.samples * A.samples) == 0u && (unsigned)r.id < n_pix) fb[(unsigned)r.id] = make_float4(0.f, 0.f, 0.f, 0.f); // clearBuffer (whole-film enumeration only)
  }
  const unsigned long long m = ballot64(hit);
  if (lane_id() == 0) sh_w[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int w = 0; w < TOP_BLOCK / 64; w++) tot += sh_w[w];
    blk_cnt[blockIdx.x] = tot;
  }
  if (blockIdx.x == 0 && threadIdx.x < 5) c[16 + threadIdx.x] = 0u; // the frame's ray totals and parked-ray total (k_zero_totals)
  if (blockIdx.x == 0 && threadIdx.x == 4) { *ovf = 0u; c[9] = 0u; }
}
__global__ __launch_bounds__(TOP_BLOCK) void k_cam1_scatter(CamArgs A, unsigned n, TopDev top, const unsigned *__restrict__ blk_cnt, QueueDesc Q,
                                                            unsigned *__restrict__ overflow, unsigned *__restrict__ c, unsigned *__restrict__ moved_count) {
  __shared__ unsigned sh_w[TOP_BLOCK / 64], sh_p[TOP_BLOCK / 64];
  unsigned part = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += TOP_BLOCK) part += blk_cnt[b];
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
  const unsigned i = blockIdx.x * TOP_BLOCK + threadIdx.x;
  bool hit = false;
  RayRec r;
  if (i < n) {
    r = camera_ray(A, (unsigned long long)i);
    float ret_t;
    hit = camera_slot_valid(A, (unsigned long long)i) && top_nearest(make_float4(r.o.x, r.o.y, r.o.z, r.t_min), make_float4(r.d.x, r.d.y, r.d.z, r.t_max), top, -1, ret_t) >= 0;
    if (hit) r.o = add3(r.o, scl3(r.d, ret_t * 0.95f)); // TracerBase.h:393
  }
  const unsigned long long m = ballot64(hit);
  if (lane_id() == 0) { sh_w[threadIdx.x >> 6] = (unsigned)__popcll(m); sh_p[threadIdx.x >> 6] = part; }
  __syncthreads();
  unsigned base = 0, mine = 0;
  for (int w = 0; w < TOP_BLOCK / 64; w++) { base += sh_p[w]; if (w < (int)(threadIdx.x >> 6)) mine += sh_w[w]; }
  if (hit) {
    const unsigned slot = base + mine + lanes_below(m);
    RayPlanes qp = make_planes(Q.planes, Q.cap);
    qp.p5 = nullptr; // a one-instance scene: no other instance a ray could have missed
    if (slot < Q.cap) store_ray(qp, slot, r);
    else atomicOr(overflow, 1u);
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { // the list is complete behind this block
    unsigned total = base;
    for (int w = 0; w < TOP_BLOCK / 64; w++) total += sh_w[w];
    *Q.count = total;
    unsigned long long *tot = (unsigned long long *)(c + 16); // k_wave_pass_begin, pass 0 (trace.hip)
    *moved_count = 0u; c[2] = 0u; c[5] = 0u; tot[0] += total;
    c[0] = 0u; c[1] = 0u; c[3] = 0u; c[4] = 0u; c[6] = 0u;
    for (int k = 0; k < SHADOW_CLASSES; k++) c[SHADOW_CLS_WORD + k] = 0u;
  }
}

__global__ __launch_bounds__(256) void k_fb_clamp(const float *__restrict__ src, float *__restrict__ dst, unsigned long long n4, int clamp) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = ((const float4 *)src)[i];
  if (clamp) { // IceTComposite::localAdd :111-117: c > 1 -> 1 (sums of non-negative terms: clamp-at-end == clamp-per-add)
    v.x = v.x > 1.f ? 1.f : v.x; v.y = v.y > 1.f ? 1.f : v.y; v.z = v.z > 1.f ? 1.f : v.z;
  }
  ((float4 *)dst)[i] = v;
}

inline unsigned blocks_for(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }

} // namespace

extern "C" int gvt_hip_camera_generate(gvt_hip_queue *q, const float eye[3], const float focus[3], const float up[3], float fov, int W,
                                       int H, int samples, int depth, float jitterF) {
  return gvt_hip_camera_generate_tiled(q, eye, focus, up, fov, W, H, samples, depth, jitterF, 0);
}

// buildTransform, RIGHT_HAND_CAMERA (gvtCamera.cpp:89-139); host float arithmetic, same order as the reference
static CamArgs make_cam_args(const float eye[3], const float focus[3], const float up[3], float fov, int W, int H, int samples, int depth, float jitterF,
                             int tile) {
  CamArgs A;
  V3 e = ld3(eye), f = ld3(focus), upv = ld3(up);
  V3 w = norm3(sub3(f, e));
  V3 v = norm3(upv);
  V3 u;
  u.x = w.y * v.z - w.z * v.y; u.y = w.z * v.x - w.x * v.z; u.z = w.x * v.y - w.y * v.x;
  u = norm3(u);
  V3 up2;
  up2.x = u.y * w.z - w.y * u.z; up2.y = u.z * w.x - w.z * u.x; up2.z = u.x * w.y - w.x * u.y;
  v = norm3(up2);
  const int jitterWindowSize = (int)jitterF; // setJitterWindowSize(int) truncates (gvtCamera.cpp:200)
  const float aspectRatio = (float)W / (float)H;
  A.eye = e; A.u = u; A.v = v; A.w = w;
  A.vert = tanf((float)(fov * 0.5));
  A.horz = tanf((float)(fov * 0.5)) * aspectRatio;
  const float divider = (float)samples;
  A.offset = (float)((1.0 / divider) * jitterWindowSize);
  A.wmult = 2.f / (float)(W - 1);
  A.hmult = 2.f / (float)(H - 1);
  A.half_sample = samples * 0.5f;
  A.contri = 1.f / (samples * samples);
  A.W = W; A.H = H; A.samples = samples; A.depth = depth; A.tile = tile; A.first = 0u;
  A.rect_on = 0; A.rx0 = 0; A.ry0 = 0; A.rx1 = W; A.ry1 = H; A.rtpr = 1;
  return A;
}

extern "C" int gvt_hip_camera_generate_tiled(gvt_hip_queue *q, const float eye[3], const float focus[3], const float up[3], float fov, int W,
                                             int H, int samples, int depth, float jitterF, int tile) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!q || W < 2 || H < 2 || samples < 1 || (tile != 0 && tile != 8)) { set_error("camera_generate: bad arguments"); return GVT_HIP_ERR_INVALID; }
  const size_t n = (size_t)W * H * samples * samples;
  int rc = queue_reserve(q, n);
  if (rc) return rc;
  const CamArgs A = make_cam_args(eye, focus, up, fov, W, H, samples, depth, jitterF, tile);
  Ctx &C = gctx();
  {
    ProfScope ps(KC_CAMERA);
    k_camera<<<blocks_for(n), 256, 0, C.stream>>>(A, make_planes(q->d_planes, q->cap), n);
  }
  HIPCHK(hipGetLastError());
  q->size = n;
  return set_device_u32(q->d_count, (unsigned)n);
}

// ---- top-level BVH order: accel/BVH.cpp:77-216 restated on the host (tiny: <= #domains) ----
namespace {
struct TopNode { float lo[3], hi[3]; int left, right; }; // left >= 0: inner (children); left < 0: leaf of `right` instances from ~left
struct TopBuild {
  const float *lo, *hi;
  std::vector<int> set, sorted;
  std::vector<TopNode> nodes;
  float centroid(int inst, int ax) const { return 0.5f * lo[3 * inst + ax] + 0.5f * hi[3 * inst + ax]; } // BBox.cpp:128
  static float fmn(float a, float b) { return (a < b) ? a : b; }
  static float fmx(float a, float b) { return (a > b) ? a : b; }
  void merge(float l[3], float h[3], int q) const {
    for (int k = 0; k < 3; k++) { l[k] = fmn(lo[3 * q + k], l[k]); h[k] = fmx(hi[3 * q + k], h[k]); }
  }
  static float area(const float l[3], const float h[3]) { // BBox.cpp:130-133
    float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    return (2.f * (dx * dy + dy * dz + dz * dx));
  }
  float split_point(int ax, int start, int end) const { // BVH.cpp:173-216
    float minCost = GVT_FLT_MAX, splitPoint = 0.f;
    for (int i = start; i < end; ++i)
      for (int e = 0; e < 2; ++e) {
        float edge = (e == 0) ? lo[3 * set[i] + ax] : hi[3 * set[i] + ax];
        float ll[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, lh[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
        float rl[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, rh[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
        int leftCount = 0;
        for (int j = start; j < end; ++j) {
          if (centroid(set[j], ax) < edge) { ++leftCount; merge(ll, lh, set[j]); } else merge(rl, rh, set[j]);
        }
        int rightCount = end - start - leftCount;
        float cost = (float)(0.5 + (area(ll, lh) * leftCount) + (area(rl, rh) * rightCount));
        if (cost < minCost) { minCost = cost; splitPoint = edge; }
      }
    return splitPoint;
  }
  int leaf(int me, int start, int end) {
    nodes[me].left = ~(int)sorted.size(); nodes[me].right = end - start;
    for (int i = start; i < end; ++i) sorted.push_back(set[i]);
    return me;
  }
  int build(int start, int end) { // BVH.cpp:77-171 (LEAF_SIZE 1); returns the node's index
    float l[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, h[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
    for (int i = start; i < end; ++i) merge(l, h, set[i]);
    const int me = (int)nodes.size();
    nodes.push_back(TopNode{ { l[0], l[1], l[2] }, { h[0], h[1], h[2] }, 0, 0 });
    if (end - start <= 1) return leaf(me, start, end);
    float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    int ax = (dx > dy && dx > dz) ? 0 : (dy > dz) ? 1 : 2; // BBox.cpp:117-126
    float sp = split_point(ax, start, end);
    int first = start, last = end; // std::partition (libstdc++ bidirectional __partition)
    for (;;) {
      bool done = false;
      for (;;) { if (first == last) { done = true; break; } else if (centroid(set[first], ax) < sp) ++first; else break; }
      if (done) break;
      --last;
      for (;;) { if (first == last) { done = true; break; } else if (!(centroid(set[last], ax) < sp)) --last; else break; }
      if (done) break;
      std::swap(set[first], set[last]);
      ++first;
    }
    int splitIdx = first;
    if (splitIdx == start || splitIdx == end) return leaf(me, start, end);
    const int a = build(start, splitIdx);
    const int b = build(splitIdx, end);
    nodes[me].left = a; nodes[me].right = b;
    return me;
  }
};
} // namespace

extern "C" gvt_hip_top *gvt_hip_top_create(const float *inst_lo, const float *inst_hi, size_t n) {
  if (ensure_init()) return nullptr;
  if (!inst_lo || !inst_hi) { set_error("top_create: null boxes"); return nullptr; }
  gvt_hip_top *T = new gvt_hip_top();
  T->n = n;
  TopBuild B{ inst_lo, inst_hi, {}, {}, {} };
  B.set.resize(n);
  for (size_t i = 0; i < n; i++) B.set[i] = (int)i;
  if (n) B.build(0, (int)n);
  T->order = B.sorted;
  T->h_lo.assign(inst_lo, inst_lo + 3 * n); T->h_hi.assign(inst_hi, inst_hi + 3 * n);
  std::vector<float4> lo(n ? n : 1), hi(n ? n : 1);
  for (size_t k = 0; k < n; k++) {
    int q = T->order[k];
    lo[k] = make_float4(inst_lo[3 * q], inst_lo[3 * q + 1], inst_lo[3 * q + 2], __builtin_bit_cast(float, q));
    hi[k] = make_float4(inst_hi[3 * q], inst_hi[3 * q + 1], inst_hi[3 * q + 2], 0.f);
  }
  bool ok = hipMalloc((void **)&T->d_lo, sizeof(float4) * (n ? n : 1)) == hipSuccess &&
            hipMalloc((void **)&T->d_hi, sizeof(float4) * (n ? n : 1)) == hipSuccess &&
            hipMalloc((void **)&T->d_hist, sizeof(unsigned) * (n ? n : 1)) == hipSuccess &&
            hipMalloc((void **)&T->d_qdesc, sizeof(QueueDesc) * (n ? n : 1)) == hipSuccess &&
            hipHostMalloc((void **)&T->h_hist, sizeof(unsigned) * (n ? n : 1), hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void **)&T->h_qdesc, sizeof(QueueDesc) * (n ? n : 1), hipHostMallocDefault) == hipSuccess;
  if (ok && n) {
    ok = hipMemcpy(T->d_lo, lo.data(), sizeof(float4) * n, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(T->d_hi, hi.data(), sizeof(float4) * n, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok && !B.nodes.empty()) { // the tree itself, for larger sets (gvt_device.h top_nearest)
    const size_t nn = B.nodes.size();
    std::vector<float4> nlo(nn), nhi(nn);
    for (size_t k = 0; k < nn; k++) {
      const TopNode &N = B.nodes[k];
      nlo[k] = make_float4(N.lo[0], N.lo[1], N.lo[2], __builtin_bit_cast(float, N.left));
      nhi[k] = make_float4(N.hi[0], N.hi[1], N.hi[2], __builtin_bit_cast(float, N.right));
    }
    ok = hipMalloc((void **)&T->d_nlo, sizeof(float4) * nn) == hipSuccess && hipMalloc((void **)&T->d_nhi, sizeof(float4) * nn) == hipSuccess &&
         hipMemcpy(T->d_nlo, nlo.data(), sizeof(float4) * nn, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(T->d_nhi, nhi.data(), sizeof(float4) * nn, hipMemcpyHostToDevice) == hipSuccess;
    T->n_nodes = nn;
  }
  if (!ok) { set_error("top_create: device allocation failed"); gvt_hip_top_destroy(T); return nullptr; }
  return T;
}
extern "C" void gvt_hip_top_destroy(gvt_hip_top *T) {
  if (!T) return;
  hipFree(T->d_lo); hipFree(T->d_hi); hipFree(T->d_nlo); hipFree(T->d_nhi); hipFree(T->d_hist); hipFree(T->d_qdesc); hipHostFree(T->h_hist); hipHostFree(T->h_qdesc);
  delete T;
}
extern "C" int gvt_hip_top_order(const gvt_hip_top *T, int32_t *out) {
  if (!T || !out) { set_error("top_order: null"); return GVT_HIP_ERR_INVALID; }
  for (size_t i = 0; i < T->n; i++) out[i] = T->order[i];
  return 0;
}

// shuffleRays over n rays taken from S (a queue's planes or the camera)
static int shuffle_impl(gvt_hip_top *T, const RaySrc &in, size_t n, int from, gvt_hip_queue *const *queues, const uint8_t *keep_mask, gvt_hip_fb *fb,
                        const int *from_arr = nullptr) {
  Ctx &C = gctx();
  if (!n) return 0;
  hipStream_t st = C.stream;
  int *d_next = (int *)scratch_get(6, sizeof(int) * n);
  float *d_t = (float *)scratch_get(7, sizeof(float) * n);
  if (!d_next || !d_t) return GVT_HIP_ERR_DEVICE;
  const size_t nI = T->n;
  const int use_lds = (C.top_lds && nI > 0 && nI <= 4096) ? 1 : 0; // LDS counters per destination; beyond that straight to the global ones
  const unsigned n_blk = blocks_for(n, TOP_BLOCK);
  unsigned *d_blk = nullptr; // ordered mode: [destination][block] counts, then base slots
  if (C.top_ordered && use_lds && nI <= GVT_TOP_ORDERED_MAX) {
    d_blk = (unsigned *)scratch_get(14, sizeof(unsigned) * nI * n_blk);
    if (!d_blk) return GVT_HIP_ERR_DEVICE;
  }
  // When every kept queue already has room for all n rays the scatter is launched right behind the classification and the
  // per-destination totals are read back once, at the end; exact growth needs them on the host before the scatter.
  unsigned *hist = T->h_hist;
  QueueDesc *desc = (QueueDesc *)T->h_qdesc;
  bool roomy = true;
  for (size_t i = 0; i < nI; i++) {
    const bool keep = !keep_mask || keep_mask[i];
    if (!keep) continue;
    if (nI <= 16 && queues[i]->cap < queues[i]->size + n) { // few domains: worst-case room is cheap next to 288 GB of HBM
      int rc = queue_reserve(queues[i], queues[i]->size + n);
      if (rc) return rc;
    }
    if (queues[i]->cap < queues[i]->size + n) roomy = false;
  }
  const bool scan_totals = d_blk && roomy; // ordered mode: k_top_scan leaves the totals in d_hist, no atomics and no memset needed
  if (nI && !scan_totals) HIPCHK(hipMemsetAsync(T->d_hist, 0, sizeof(unsigned) * nI, st));
  {
    ProfScope ps(KC_SHUFFLE);
    k_top_classify<<<n_blk, TOP_BLOCK, use_lds ? sizeof(unsigned) * nI : 0, st>>>(in, (unsigned)n, T->dev(), (int)nI, from, d_next, d_t,
                                                                                scan_totals ? nullptr : T->d_hist, use_lds, d_blk, nullptr, from_arr, C.skip_known);
  }
  HIPCHK(hipGetLastError());
  if (!roomy) {
    if (nI) HIPCHK(hipMemcpyAsync(hist, T->d_hist, sizeof(unsigned) * nI, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  for (size_t i = 0; i < nI; i++) {
    const bool keep = !keep_mask || keep_mask[i];
    gvt_hip_queue *Q = queues[i];
    if (!roomy && keep && hist[i]) {
      int rc = queue_reserve(Q, Q->size + hist[i] + (hist[i] >> 2)); // head-room: later frames take the single-sync path
      if (rc) return rc;
    }
    desc[i].planes = Q->d_planes; desc[i].cap = Q->cap; desc[i].count = Q->d_count; desc[i].keep = keep ? 1u : 0u;
  }
  if (nI) HIPCHK(hipMemcpyAsync(T->d_qdesc, desc, sizeof(QueueDesc) * nI, hipMemcpyHostToDevice, st));
  T->qdesc_uploaded.clear(); // (the asynchronous shuffle's upload cache no longer describes d_qdesc)
  {
    ProfScope ps(KC_SHUFFLE);
    if (d_blk) k_top_scan<<<(unsigned)nI, TOP_BLOCK, 0, st>>>(d_blk, n_blk, (const QueueDesc *)T->d_qdesc, scan_totals ? T->d_hist : nullptr);
    const size_t lds = d_blk ? sizeof(unsigned) * nI * (TOP_BLOCK / 64) : (use_lds ? 2 * sizeof(unsigned) * nI : 0);
    k_top_scatter<<<n_blk, TOP_BLOCK, lds, st>>>(in, (unsigned)n, d_next, d_t, (const QueueDesc *)T->d_qdesc, (int)nI, fb ? fb->d_rgba : nullptr,
                                               fb ? (unsigned)(fb->w * fb->h) : 0u, use_lds, d_blk);
  }
  HIPCHK(hipGetLastError());
  if (roomy && nI) HIPCHK(hipMemcpyAsync(hist, T->d_hist, sizeof(unsigned) * nI, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st)); // desc/hist are read by the copies above
  for (size_t i = 0; i < nI; i++)
    if (desc[i].keep) queues[i]->size += hist[i];
  return 0;
}

// shuffleRays of a device-resident list whose length only the device knows (q_in's count word), launched without any host round
// trip: every destination must already have room for its current rays + n_ub more.  Destination counts advance on the device only;
// the caller learns them from its next read-back.  from_arr: source instance per ray (merged rounds), else `from` for all.  Kernel order on the stream:
// classify -> (scan) -> scatter.
// d_qdesc: the queue descriptors already on the device (uploaded by the caller with its other per-round tables), or null.
static int shuffle_async_src(gvt_hip_top *T, const RaySrc &S, size_t n_ub, const unsigned *n_dev, const int *from_arr, int from, gvt_hip_queue *const *queues,
                             const uint8_t *keep_mask, gvt_hip_fb *fb, unsigned *d_overflow, const void *d_qdesc);
int shuffle_async(gvt_hip_top *T, gvt_hip_queue *q_in, size_t n_ub, const int *from_arr, int from, gvt_hip_queue *const *queues, const uint8_t *keep_mask,
                  gvt_hip_fb *fb, unsigned *d_overflow, const void *d_qdesc) {
  RaySrc S{};
  S.q = make_planes(q_in->d_planes, q_in->cap);
  S.from_cam = 0;
  return shuffle_async_src(T, S, n_ub, q_in->d_count, from_arr, from, queues, keep_mask, fb, d_overflow, d_qdesc);
}
// generateRays + FilterRaysLocally without a read-back: every kept queue must have room for all W*H*samples^2 rays; the queue counts
// advance on the device only (the caller's first launch chain reads its ray count from there)
// [first, first + count) of the generated list (count == 0: all of it): a rank's portion under the multi-rank Image scheduler
static bool camera_keep_rect(const gvt_hip_top *T, CamArgs &A, const uint8_t *keep_mask);
// rect: enumerate only the film rectangle the kept instances project onto (the whole list, first == count == 0)
int camera_filter_async(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, gvt_hip_queue *const *queues, const uint8_t *keep_mask, unsigned *d_overflow,
                        size_t first, size_t count, bool rect) {
  const size_t n_all = (size_t)cam->width * cam->height * cam->samples * cam->samples;
  if (n_all > 0xffffffffull) { set_error("camera_filter: more than 2^32 rays"); return GVT_HIP_ERR_INVALID; }
  RaySrc S{};
  S.cam = make_cam_args(cam->eye, cam->focus, cam->up, cam->fov, cam->width, cam->height, cam->samples, cam->depth, cam->jitter_window_size, tile);
  S.from_cam = 1;
  if (rect && !first && !count && tile == 8 && camera_keep_rect(T, S.cam, keep_mask))
    count = (size_t)S.cam.rtpr * (size_t)((S.cam.ry1 - S.cam.ry0 + 7) / 8) * 64u * (size_t)(cam->samples * cam->samples); // (0: nothing in view, nothing launched)
  else {
    if (!count) count = n_all - first;
    S.cam.first = (unsigned)first;
  }
  return shuffle_async_src(T, S, count, nullptr, nullptr, -1, queues, keep_mask, nullptr, d_overflow, nullptr);
}
// Upper bound of the camera rays that can have instance `inst` as their first domain: the positions of the film rectangle its box
// projects onto (the whole list where there is no bounded projection, camera_keep_rect)
size_t camera_instance_bound(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, size_t inst) {
  const size_t n_all = (size_t)cam->width * cam->height * cam->samples * cam->samples;
  if (tile != 8 || inst >= T->n) return n_all;
  CamArgs A = make_cam_args(cam->eye, cam->focus, cam->up, cam->fov, cam->width, cam->height, cam->samples, cam->depth, cam->jitter_window_size, tile);
  std::vector<uint8_t> only(T->n, (uint8_t)0);
  only[inst] = 1;
  if (!camera_keep_rect(T, A, only.data())) return n_all;
  return std::min(n_all, (size_t)A.rtpr * (size_t)((A.ry1 - A.ry0 + 7) / 8) * 64u * (size_t)(cam->samples * cam->samples));
}
// clearBuffer + generateRays + FilterRaysLocally for a ONE-instance scene on one rank, with the launch chain's pass-0 resets folded
// in (k_cam1_count / k_cam1_scatter).  q must have room for all W*H*samples^2 rays; its count lives on the device only.
int camera_one_instance_async(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, gvt_hip_queue *q, gvt_hip_fb *fb, unsigned *d_overflow, unsigned *d_moved_count) {
  Ctx &C = gctx();
  const size_t n = (size_t)cam->width * cam->height * cam->samples * cam->samples;
  if (!n || n > 0xffffffffull || T->n != 1) { set_error("camera_one_instance: bad arguments"); return GVT_HIP_ERR_INVALID; }
  CamArgs A = make_cam_args(cam->eye, cam->focus, cam->up, cam->fov, cam->width, cam->height, cam->samples, cam->depth, cam->jitter_window_size, tile);
  // only the film rectangle the instance's box projects onto is enumerated (no other camera ray can enter it); the kernels then no longer
  // touch every pixel, so the framebuffer is cleared by a fill in front of them
  size_t n_list = n;
  if (tile == 8 && camera_keep_rect(T, A, nullptr)) {
    n_list = (size_t)A.rtpr * (size_t)((A.ry1 - A.ry0 + 7) / 8) * 64u * (size_t)(cam->samples * cam->samples);
    if (fb) HIPCHK(hipMemsetAsync(fb->d_rgba, 0, sizeof(float) * 4 * (size_t)fb->w * fb->h, C.stream));
    fb = nullptr;
  }
  if (!n_list) n_list = 1; // (nothing in view: one empty position, so that the kernels still publish the counts and do the resets)
  const unsigned n_blk = blocks_for(n_list, TOP_BLOCK);
  unsigned *d_blk = (unsigned *)scratch_get(14, sizeof(unsigned) * n_blk);
  if (!d_blk) return GVT_HIP_ERR_DEVICE;
  QueueDesc Q{ q->d_planes, q->cap, q->d_count, 1u };
  {
    ProfScope ps(KC_SHUFFLE);
    k_cam1_count<<<n_blk, TOP_BLOCK, 0, C.stream>>>(A, (unsigned)n_list, T->dev(), d_blk, fb ? (float4 *)fb->d_rgba : nullptr, fb ? (unsigned)(fb->w * fb->h) : 0u, C.d_counters, d_overflow);
    k_cam1_scatter<<<n_blk, TOP_BLOCK, 0, C.stream>>>(A, (unsigned)n_list, T->dev(), d_blk, Q, d_overflow, C.d_counters, d_moved_count);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int shuffle_async_src(gvt_hip_top *T, const RaySrc &S, size_t n_ub, const unsigned *n_dev, const int *from_arr, int from, gvt_hip_queue *const *queues,
                             const uint8_t *keep_mask, gvt_hip_fb *fb, unsigned *d_overflow, const void *d_qdesc) {
  Ctx &C = gctx();
  if (!n_ub) return 0;
  hipStream_t st = C.stream;
  int *d_next = (int *)scratch_get(6, sizeof(int) * n_ub);
  float *d_t = (float *)scratch_get(7, sizeof(float) * n_ub);
  if (!d_next || !d_t) return GVT_HIP_ERR_DEVICE;
  const size_t nI = T->n;
  const int use_lds = (C.top_lds && nI > 0 && nI <= 4096) ? 1 : 0;
  const unsigned n_blk = blocks_for(n_ub, TOP_BLOCK);
  unsigned *d_blk = nullptr;
  if (C.top_ordered && use_lds && nI <= GVT_TOP_ORDERED_MAX) {
    d_blk = (unsigned *)scratch_get(14, sizeof(unsigned) * nI * n_blk);
    if (!d_blk) return GVT_HIP_ERR_DEVICE;
  }
  const QueueDesc *qd = (const QueueDesc *)d_qdesc;
  if (!qd) {
    QueueDesc *desc = (QueueDesc *)T->h_qdesc;
    for (size_t i = 0; i < nI; i++) {
      gvt_hip_queue *Q = queues[i];
      desc[i].planes = Q->d_planes; desc[i].cap = Q->cap; desc[i].count = Q->d_count; desc[i].keep = (!keep_mask || keep_mask[i]) ? 1u : 0u;
    }
    // (the descriptors rarely change between frames: copy only when they differ from what the device holds)
    if (nI && (T->qdesc_uploaded.size() != sizeof(QueueDesc) * nI || std::memcmp(T->qdesc_uploaded.data(), desc, sizeof(QueueDesc) * nI) != 0)) {
      HIPCHK(hipMemcpyAsync(T->d_qdesc, desc, sizeof(QueueDesc) * nI, hipMemcpyHostToDevice, st));
      T->qdesc_uploaded.assign((const unsigned char *)desc, (const unsigned char *)desc + sizeof(QueueDesc) * nI);
    }
    qd = (const QueueDesc *)T->d_qdesc;
  }
  {
    ProfScope ps(KC_SHUFFLE);
    k_top_classify<<<n_blk, TOP_BLOCK, use_lds ? sizeof(unsigned) * nI : 0, st>>>(S, (unsigned)n_ub, T->dev(), (int)nI, from, d_next, d_t, nullptr, use_lds, d_blk,
                                                                                n_dev, from_arr, C.skip_known);
    if (d_blk) k_top_scan<<<(unsigned)nI, TOP_BLOCK, 0, st>>>(d_blk, n_blk, qd, nullptr);
    const size_t lds = d_blk ? sizeof(unsigned) * nI * (TOP_BLOCK / 64) : (use_lds ? 2 * sizeof(unsigned) * nI : 0);
    k_top_scatter<<<n_blk, TOP_BLOCK, lds, st>>>(S, (unsigned)n_ub, d_next, d_t, qd, (int)nI, fb ? fb->d_rgba : nullptr,
                                               fb ? (unsigned)(fb->w * fb->h) : 0u, use_lds, d_blk, n_dev, d_overflow);
  }
  HIPCHK(hipGetLastError());
  return 0; // q_in's count word is left as it is: the caller's next launch chain resets it (k_wave_pass_begin)
}

extern "C" int gvt_hip_shuffle(gvt_hip_top *T, gvt_hip_queue *q_in, int from, gvt_hip_queue *const *queues, const uint8_t *keep_mask,
                               gvt_hip_fb *fb) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!T || !q_in || (T->n && !queues)) { set_error("shuffle: null argument"); return GVT_HIP_ERR_INVALID; }
  for (size_t i = 0; i < T->n; i++)
    if (!queues[i] || queues[i] == q_in) { set_error("shuffle: queue %zu is null or aliases q_in", i); return GVT_HIP_ERR_INVALID; }
  if (!q_in->size) return 0;
  RaySrc S{};
  S.q = make_planes(q_in->d_planes, q_in->cap);
  S.from_cam = 0;
  int rc = shuffle_impl(T, S, q_in->size, from, queues, keep_mask, fb);
  if (rc) { // (the classify step has already advanced origins in place: a failed shuffle CONSUMES q_in too -- it is cleared, never left half-walked for a second attempt)
    const std::string why = gvt_hip_last_error();
    q_in->size = 0;
    hipMemsetAsync(q_in->d_count, 0, sizeof(unsigned), gctx().stream);
    set_error("%s; q_in has been cleared (its rays are lost)", why.c_str());
    return rc;
  }
  q_in->size = 0; // rays.clear(), TracerBase.h:411
  HIPCHK(hipMemsetAsync(q_in->d_count, 0, sizeof(unsigned), gctx().stream));
  return 0;
}

// shuffleRays of a device-resident list with EXACT growth of the destinations (two host synchronisations: the list's length, then the
// per-destination counts): the path for rounds whose worst-case reservation -- every destination sized for every moved ray -- would
// not fit the round's memory budget (hundreds of domains).  from_arr: source instance per ray, else `from` for all.
int shuffle_exact(gvt_hip_top *T, gvt_hip_queue *q_in, const int *from_arr, int from, gvt_hip_queue *const *queues, gvt_hip_fb *fb) {
  Ctx &C = gctx();
  HIPCHK(hipMemcpyAsync(C.h_pinned + 12, q_in->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, C.stream));
  HIPCHK(hipStreamSynchronize(C.stream));
  const size_t n = C.h_pinned[12];
  if (!n) return 0;
  RaySrc S{};
  S.q = make_planes(q_in->d_planes, q_in->cap);
  S.from_cam = 0;
  return shuffle_impl(T, S, n, from, queues, nullptr, fb, from_arr);
}

// The pixels the kept instances' boxes (keep_mask == NULL: all instances) project onto, as a tile-aligned rectangle of the film (two pixels of margin): a camera ray of any
// other pixel cannot enter one of those boxes at all, so it cannot have one of them as its first domain (shuffleDropRays keeps nothing of
// it).  False -- the whole film -- when a box reaches behind the eye plane, when sub-samples are spread by a jitter window, or when the
// rectangle is nearly the whole film anyway.  An empty rectangle (no kept instance in view) sets rx1 <= rx0.
static bool camera_keep_rect(const gvt_hip_top *T, CamArgs &A, const uint8_t *keep_mask) {
  if (A.offset != 0.f || T->h_lo.size() != 3 * T->n) return false;
  double x0 = 1e30, y0 = 1e30, x1 = -1e30, y1 = -1e30;
  bool any = false;
  for (size_t i = 0; i < T->n; i++) {
    if (keep_mask && !keep_mask[i]) continue;
    any = true;
    for (int c = 0; c < 8; c++) {
      const double p[3] = { (c & 1) ? T->h_hi[3 * i] : T->h_lo[3 * i], (c & 2) ? T->h_hi[3 * i + 1] : T->h_lo[3 * i + 1], (c & 4) ? T->h_hi[3 * i + 2] : T->h_lo[3 * i + 2] };
      const double q[3] = { p[0] - A.eye.x, p[1] - A.eye.y, p[2] - A.eye.z };
      const double depth = q[0] * A.w.x + q[1] * A.w.y + q[2] * A.w.z;
      if (!(depth > 1e-4)) return false; // at or behind the eye plane: no bounded projection
      const double x = (q[0] * A.u.x + q[1] * A.u.y + q[2] * A.u.z) / depth, y = (q[0] * A.v.x + q[1] * A.v.y + q[2] * A.v.z) / depth;
      const double px = (x / A.horz + 1.0) / A.wmult, py = (y / A.vert + 1.0) / A.hmult; // the inverse of camera_ray's x0 / y0
      x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
    }
  }
  int ix0 = 0, iy0 = 0, ix1 = 0, iy1 = 0;
  if (any) {
    ix0 = (int)std::max(0.0, std::floor(x0) - 2.0); iy0 = (int)std::max(0.0, std::floor(y0) - 2.0);
    ix1 = (int)std::min((double)A.W, std::ceil(x1) + 3.0); iy1 = (int)std::min((double)A.H, std::ceil(y1) + 3.0);
    if (ix1 < ix0) ix1 = ix0;
    if (iy1 < iy0) iy1 = iy0;
  }
  ix0 &= ~7; iy0 &= ~7;
  if ((double)(ix1 - ix0) * (double)(iy1 - iy0) > 0.9 * (double)A.W * (double)A.H) return false;
  A.rect_on = 1; A.rx0 = ix0; A.ry0 = iy0; A.rx1 = ix1; A.ry1 = iy1; A.rtpr = std::max(1, (ix1 - ix0 + 7) / 8);
  return true;
}

// generateRays + FilterRaysLocally in one step: the camera's rays go straight to the queues of the instances they enter first
extern "C" int gvt_hip_camera_filter(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, gvt_hip_queue *const *queues, const uint8_t *keep_mask) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!T || !cam || (T->n && !queues)) { set_error("camera_filter: null argument"); return GVT_HIP_ERR_INVALID; }
  if (cam->width < 2 || cam->height < 2 || cam->samples < 1 || (tile != 0 && tile != 8)) { set_error("camera_filter: bad arguments"); return GVT_HIP_ERR_INVALID; }
  for (size_t i = 0; i < T->n; i++)
    if (!queues[i]) { set_error("camera_filter: queue %zu is null", i); return GVT_HIP_ERR_INVALID; }
  const size_t n = (size_t)cam->width * cam->height * cam->samples * cam->samples;
  if (n > 0xffffffffull) { set_error("camera_filter: more than 2^32 rays"); return GVT_HIP_ERR_INVALID; }
  RaySrc S{};
  S.cam = make_cam_args(cam->eye, cam->focus, cam->up, cam->fov, cam->width, cam->height, cam->samples, cam->depth, cam->jitter_window_size, tile);
  S.from_cam = 1;
  size_t n_list = n;
  if (tile == 8 && camera_keep_rect(T, S.cam, keep_mask)) // only the pixels the (kept) instances project onto: no other camera ray can enter one of them
    n_list = (size_t)S.cam.rtpr * (size_t)((S.cam.ry1 - S.cam.ry0 + 7) / 8) * 64u * (size_t)(cam->samples * cam->samples);
  if (!n_list) return 0;
  return shuffle_impl(T, S, n_list, -1, queues, keep_mask, nullptr);
}

// ---- framebuffer ----
extern "C" gvt_hip_fb *gvt_hip_fb_create(int w, int h) {
  if (ensure_init()) return nullptr;
  if (w <= 0 || h <= 0) { set_error("fb_create: bad size"); return nullptr; }
  gvt_hip_fb *F = new gvt_hip_fb();
  F->w = w; F->h = h;
  if (hipMalloc((void **)&F->d_rgba, sizeof(float) * 4 * (size_t)w * h) != hipSuccess) { set_error("fb_create: hipMalloc failed"); delete F; return nullptr; }
  hipMemsetAsync(F->d_rgba, 0, sizeof(float) * 4 * (size_t)w * h, gctx().stream);
  return F;
}
extern "C" void gvt_hip_fb_destroy(gvt_hip_fb *F) {
  if (!F) return;
  hipFree(F->d_rgba);
  delete F;
}
extern "C" int gvt_hip_fb_clear(gvt_hip_fb *F) {
  if (!F) { set_error("fb_clear: null"); return GVT_HIP_ERR_INVALID; }
  HIPCHK(hipMemsetAsync(F->d_rgba, 0, sizeof(float) * 4 * (size_t)F->w * F->h, gctx().stream));
  return 0;
}
extern "C" void *gvt_hip_fb_device_ptr(gvt_hip_fb *F) { return F ? (void *)F->d_rgba : nullptr; }
extern "C" int gvt_hip_fb_download(gvt_hip_fb *F, float *rgba, int clamp) {
  if (!F || !rgba) { set_error("fb_download: null"); return GVT_HIP_ERR_INVALID; }
  Ctx &C = gctx();
  const size_t n4 = (size_t)F->w * F->h;
  float *tmp = (float *)scratch_get(6, sizeof(float) * 4 * n4);
  if (!tmp) return GVT_HIP_ERR_DEVICE;
  k_fb_clamp<<<blocks_for(n4), 256, 0, C.stream>>>(F->d_rgba, tmp, n4, clamp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(rgba, tmp, sizeof(float) * 4 * n4, hipMemcpyDeviceToHost, C.stream));
  HIPCHK(hipStreamSynchronize(C.stream));
  return 0;
}
extern "C" int gvt_hip_fb_write_ppm_bytes(gvt_hip_fb *F, unsigned char *rgb) {
  if (!F || !rgb) { set_error("fb_write_ppm_bytes: null"); return GVT_HIP_ERR_INVALID; }
  std::vector<float> host((size_t)F->w * F->h * 4);
  int rc = gvt_hip_fb_download(F, host.data(), 1);
  if (rc) return rc;
  size_t o = 0; // IceTComposite::write :119-157 -- rows bottom-up, truncating cast
  for (int j = F->h - 1; j >= 0; j--)
    for (int i = 0; i < F->w; ++i) {
      size_t index = 4 * ((size_t)j * F->w + i);
      rgb[o++] = (unsigned char)(host[index + 0] * 255);
      rgb[o++] = (unsigned char)(host[index + 1] * 255);
      rgb[o++] = (unsigned char)(host[index + 2] * 255);
    }
  return 0;
}

// ---- Tracer<ImageScheduler>::operator() (algorithm/ImageTracer.h:127-269), one rank, native loop ----
extern "C" int gvt_hip_image_frame(gvt_hip_top *T, gvt_hip_mesh *const *meshes, const float *m, const float *minv, const float *normi, size_t n_inst,
                                   const gvt_hip_light *lights, size_t n_lights, int normal_mode, const gvt_hip_camera *cam,
                                   gvt_hip_queue *const *queues, gvt_hip_queue *q_cam, gvt_hip_queue *q_moved, gvt_hip_fb *fb, uint64_t *adapter_calls) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  (void)q_cam; // the camera list is no longer materialised
  if (!T || !cam || !q_moved || !fb || (n_inst && (!meshes || !m || !minv || !normi || !queues)) || T->n != n_inst) {
    set_error("image_frame: null or inconsistent argument");
    return GVT_HIP_ERR_INVALID;
  }
  int rc;
  if ((rc = gvt_hip_fb_clear(fb))) return rc;                                              // clearBuffer :142
  for (size_t i = 0; i < n_inst; i++) if ((rc = gvt_hip_queue_clear(queues[i]))) return rc;
  if ((rc = gvt_hip_queue_clear(q_moved))) return rc;
  if ((rc = gvt_hip_camera_filter(T, cam, gctx().camera_tile, queues, nullptr))) return rc; // generateRays :137 + FilterRaysLocally :146
  uint64_t calls = 0;
  for (;;) {                                                                               // do { ... } while (instTarget != -1) :159-259
    int target = -1;
    size_t cnt = 0;
    for (size_t i = 0; i < n_inst; i++)
      if (queues[i]->size > cnt) { cnt = queues[i]->size; target = (int)i; }
    if (target < 0) break;
    if ((rc = gvt_hip_trace_queue_sink(meshes[target], queues[target], q_moved, m + 16 * (size_t)target, minv + 16 * (size_t)target,
                                       normi + 9 * (size_t)target, lights, n_lights, normal_mode, (uint32_t)calls, T, target, fb))) return rc;
    calls++;
    if ((rc = gvt_hip_shuffle(T, q_moved, target, queues, nullptr, fb))) return rc;        // shuffleRays(moved_rays, instTarget) :252
  }
  if (adapter_calls) *adapter_calls = calls;
  return 0;
}
