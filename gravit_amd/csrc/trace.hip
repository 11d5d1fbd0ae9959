// trace.hip -- the adapter hot path on gfx950: Adapter::trace
// (src/gvt/render/Adapter.h:82-84, adapter/embree/EmbreeMeshAdapter.cpp:436-660).
//
//   k_trace<false,..>  rtcIntersect (EmbreeMeshAdapter.cpp:474): ray -> object space, nearest hit in (1e-6, inf)
//   k_long_closest     the few rays k_trace parks after `long_steps` node steps, finished a wave per ray
//   k_shade            per-lane block (:483-609): miss forward / shadow drop / normal / material / Shade /
//                      shadow-ray generation (:320-358) / Russian-roulette bounce (:584-602)
//   k_trace<true,..>   rtcOccluded  (:364-385): un-occluded shadow rays are appended to moved_rays -- or, with a sink, end
//                      right there by the terminal rule of the shuffleRays that would follow (TracerBase.h:396-400)
//   trace_core         the host side of one Adapter::trace call: closest -> shade -> any, one read-back per pass
//   wave_trace_chain   the same chain for a whole scheduler ROUND (all non-empty local queues at once, MULTI kernel variants),
//                      ray counts in device memory, no host round trip; k_long_seed / k_wave_any: a wave per ray for small rounds
//
// One lane per ray.  k_trace runs persistent waves: a wave pulls index ranges from a device counter (an exit every wave
// reaches), refills lanes whose ray has finished while the others go on, alternates a tight loop over the compressed 4-wide
// nodes with a leaf phase, keeps its traversal stacks in LDS (stack[level][lane]: lane-contiguous, bank conflict free) and
// compacts survivors per wave (__ballot + mbcnt, one atomic per >= 64 rays).  Not in this file: diag_kernels.inc (the visit-count
// diagnostic) and, compiled only with -DGVT_EXPERIMENTS (libgvt_hip_exp.so, the library the knob sweeps and probes run against),
// the variants that were measured and lost -- experiments/: the first-version kernels (trav_kernel=0), the binary-node and
// quad-cooperative-fetch arms of k_trace (wide4=0, coop_fetch=1), k_fused, k_packet, k_traceq (four lanes per ray).  DESIGN.md 4.1
// lists what was measured on the way.
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

// ------------------------------------------------------------------------------------------------
// Persistent-wave traversal with lane refill ("wavefront compaction of active rays").
//
// A 64-ray batch finishes when its slowest ray does (measured on the 10 M-triangle soup: mean 54 steps per
// ray, but 125 per batch -- 43 % of the lanes do useful work).  Here a wave keeps a private range of ray
// indices [c_next, c_end) taken CHUNK at a time from the device counter; whenever refill_min lanes have
// retired their ray (__ballot), the idle lanes are handed the next indices of that range (rank by mbcnt)
// and start a fresh ray with an empty stack while their neighbours continue.  Traversal is while-while:
// all lanes that still have an inner node descend together, then the lanes that reached a leaf intersect
// their triangles together.  Every wave exits once the counter has passed n and its lanes are empty.
// ------------------------------------------------------------------------------------------------
#ifndef GVT_TRI_NG_FROM_SLOT
#define GVT_TRI_NG_FROM_SLOT 1 // measured: reading 48 of the 64 slot bytes and rebuilding Ng is not faster (same cache line either way)
#endif
#ifndef GVT_STAMP
#define GVT_STAMP 0
#endif
#if GVT_STAMP
__device__ unsigned long long g_stamp[24];
#endif
#define TRAV_OVF_WORD 8 // d_counters[8] of the launching context: set by k_trace / k_long_closest when a traversal stack would have exceeded
                        // LDS levels + spill entries (k_trace's `counter` is d_counters + 0, k_long_closest's d_counters + 4)
#ifndef TRAV_CHUNK
#define TRAV_CHUNK 256
#endif
#define TRAV_DONE ((int)0x80000000) // ~code with count 7: never a valid leaf reference

// Long rays.  Per-ray cost is heavy-tailed (10 M soup, binary visits: mean 53, p99 130, p99.9 206, max 711) and one traversal step is
// a dependent memory access of ~1.5 us, so a handful of rays used to set the duration of a whole closest-hit launch (capping
// rays at 100 steps -- 0.1 % of them -- shortened it by 18 %).  k_trace therefore parks a ray that exceeds `steps` inner steps,
// with its best hit so far, in `recs`; k_long_closest then traverses each parked ray with a whole wave: 64 pending nodes per step
// instead of one.  The result is the same minimum over (t, primID) -- it does not depend on the order in which boxes are opened.
// LongRec.ns > 0: the record carries the ray's pending stack (LongQ.stk[slot * LONG_SAVE + k], bottom first) and k_long_closest goes
// on from there; 0: it starts again at the root with the best hit so far as its bound.
#define LONG_SAVE 48
#define LONG_STK_CAP 65536u // records that may carry a stack (12 MiB); later ones start again at the root
struct LongRec { unsigned j, i; float bt; int bp; float bu, bv, bden; unsigned ns; };
struct LongQ {
  LongRec *recs;
  unsigned *count;
  int *stk;          // null: no stacks are saved
  unsigned stk_cap;
  int steps; // 0: off
  int steps_drain; // ... once the wave's work counter is exhausted (no refills left to hide a long ray behind): the launch then ends
                   // ~1 us per remaining step of its longest ray, so the few long rays still in flight are parked earlier
};

// One compressed 4-wide node against one ray: entry distances of the children the ray enters within [0, lim], GVT_FLT_MAX for the
// others (misses and unused slots, whose boxes are stored inverted), and their references.
//   plane t = (origin + q*scale - O) / d = q * (scale*inv_d) + (origin*inv_d - O*inv_d); scale is a power of two, so scale*inv_d is
//   exact.  The rounding of the two fused steps is covered by widening every slab by 2^-21 |O*inv_d| (about 5e-7 |O| in space, on
//   top of the padded boxes): RaySlab keeps O*inv_d moved out by that margin for the near planes (.x) and in for the far planes (.y).
// CDNA packed FP32: the four children are decoded two at a time (v_pk_fma_f32: 12 instead of 24 fused multiply-adds per node).
typedef float f2 __attribute__((ext_vector_type(2)));
struct RaySlab {
  float ix, iy, iz; // 1 / direction
  f2 ox, oy, oz;    // (O*inv_d + e, O*inv_d - e), e = 2^-21 |O*inv_d|
};
__device__ __forceinline__ RaySlab make_slab(float ix, float iy, float iz, float ox, float oy, float oz) {
  RaySlab S;
  S.ix = ix; S.iy = iy; S.iz = iz;
  const float ex = fabsf(ox) * 4.76837158e-7f, ey = fabsf(oy) * 4.76837158e-7f, ez = fabsf(oz) * 4.76837158e-7f;
  S.ox = (f2){ ox + ex, ox - ex }; S.oy = (f2){ oy + ey, oy - ey }; S.oz = (f2){ oz + ez, oz - ez };
  return S;
}
__device__ __forceinline__ f2 splat2(float v) { return (f2){ v, v }; }
// hit[k] (optional): child k is entered -- the same predicate as tn[k] < GVT_FLT_MAX, for callers that need no distances (any hit: saves
// the select and the second compare per child)
__device__ __forceinline__ void node4_test(const uint4 *__restrict__ nd, const RaySlab &S, float lim, float tn[4], int rr[4], bool *hit = nullptr) {
  const uint4 w0 = nd[0], w1 = nd[1], w2 = nd[2], w3 = nd[3];
  const float sx = __uint_as_float(w0.w) * S.ix, sy = __uint_as_float(w3.z) * S.iy, sz = __uint_as_float(w3.w) * S.iz;
  const f2 bx = __builtin_elementwise_fma(splat2(__uint_as_float(w0.x)), splat2(S.ix), -S.ox); // (near offset, far offset)
  const f2 by = __builtin_elementwise_fma(splat2(__uint_as_float(w0.y)), splat2(S.iy), -S.oy);
  const f2 bz = __builtin_elementwise_fma(splat2(__uint_as_float(w0.z)), splat2(S.iz), -S.oz);
  const unsigned qnx = S.ix >= 0.f ? w1.x : w1.y, qfx = S.ix >= 0.f ? w1.y : w1.x; // near / far planes by ray direction
  const unsigned qny = S.iy >= 0.f ? w1.z : w1.w, qfy = S.iy >= 0.f ? w1.w : w1.z;
  const unsigned qnz = S.iz >= 0.f ? w2.x : w2.y, qfz = S.iz >= 0.f ? w2.y : w2.x;
  rr[0] = (int)w2.z; rr[1] = (int)w2.w; rr[2] = (int)w3.x; rr[3] = (int)w3.y;
#define GVT_Q2(Q, SH) ((f2){ (float)(((Q) >> (SH)) & 0xffu), (float)(((Q) >> ((SH) + 8)) & 0xffu) })
#define GVT_SLAB2(SH, A, B)                                                                                          \
  {                                                                                                                  \
    const f2 nx_ = __builtin_elementwise_fma(GVT_Q2(qnx, SH), splat2(sx), splat2(bx.x));                              \
    const f2 ny_ = __builtin_elementwise_fma(GVT_Q2(qny, SH), splat2(sy), splat2(by.x));                              \
    const f2 nz_ = __builtin_elementwise_fma(GVT_Q2(qnz, SH), splat2(sz), splat2(bz.x));                              \
    const f2 fx_ = __builtin_elementwise_fma(GVT_Q2(qfx, SH), splat2(sx), splat2(bx.y));                              \
    const f2 fy_ = __builtin_elementwise_fma(GVT_Q2(qfy, SH), splat2(sy), splat2(by.y));                              \
    const f2 fz_ = __builtin_elementwise_fma(GVT_Q2(qfz, SH), splat2(sz), splat2(bz.y));                              \
    const float na_ = fmaxf(fmaxf(nx_.x, ny_.x), fmaxf(nz_.x, 0.f)), nb_ = fmaxf(fmaxf(nx_.y, ny_.y), fmaxf(nz_.y, 0.f)); \
    const float fa_ = fminf(fminf(fx_.x, fy_.x), fz_.x) * 1.0000004f, fb_ = fminf(fminf(fx_.y, fy_.y), fz_.y) * 1.0000004f; \
    const bool ha_ = na_ <= fminf(fa_, lim), hb_ = nb_ <= fminf(fb_, lim);                                            \
    tn[A] = ha_ ? na_ : GVT_FLT_MAX; /* a miss sorts last */                                                         \
    tn[B] = hb_ ? nb_ : GVT_FLT_MAX;                                                                                 \
    if (hit) { hit[A] = ha_; hit[B] = hb_; }                                                                         \
  }
  GVT_SLAB2(0, 0, 1) GVT_SLAB2(16, 2, 3)
#undef GVT_SLAB2
#undef GVT_Q2
}

// merged launches: segment of virtual ray index g (segments sorted by `begin`, n_seg small)
__device__ inline int wave_find_seg(const WaveSet &W, unsigned g) {
  int lo = 0, hi = W.n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (W.segs[mid].begin <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// copies the rays whose indices are parked in the wave's LDS list to consecutive slots of `out`; with a sink, rays that meet no
// other instance end here (shuffleRays' terminal rule) and only the others are copied
// ray_inst / out_from (merged launches): instance a shadow ray was generated in, per ray of q / per ray of out (shuffleRays' `from`)
__device__ inline void flush_pending(volatile unsigned *pend, int n_pend, const RayPlanes &q, const RayPlanes &out, unsigned *out_count, const TermSink &K,
                                     const int *__restrict__ ray_inst = nullptr, int *__restrict__ out_from = nullptr) {
  if (!K.fb) {
    unsigned base = 0;
    if (lane_id() == 0) base = atomicAdd(out_count, (unsigned)n_pend);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (int k = (int)lane_id(); k < n_pend; k += 64) {
      const unsigned src = pend[k];
      out.p0[base + k] = q.p0[src]; out.p1[base + k] = q.p1[src]; out.p2[base + k] = q.p2[src]; out.p3[base + k] = q.p3[src];
      if (out.p4) out.p4[base + k] = 0u; // shadow rays never draw: their stream word is 0
      store_no_known(out, base + k);     // ... and have crossed nothing yet
      if (out_from) out_from[base + k] = ray_inst[src];
    }
    return;
  }
  // a scene of ONE instance whose rays these are: there is nothing ahead of any of them -- no origin / direction fetch, no box test
  const bool alone = K.top.n_inst == 1 && !ray_inst && K.from >= 0;
  for (int k0 = 0; k0 < n_pend; k0 += 64) {
    const int k = k0 + (int)lane_id();
    bool go_on = false;
    float4 a, b, c, d;
    int from = K.from;
    if (k < n_pend) {
      const unsigned src = pend[k];
      c = q.p2[src]; d = q.p3[src];
      if (!alone) {
        a = q.p0[src]; b = q.p1[src];
        if (ray_inst) from = ray_inst[src];
        float ret_t;
        go_on = top_nearest(a, b, K.top, from, ret_t) >= 0;
      }
      if (!go_on) {
        const V3 col = mk3(c.x, c.y, c.z);
        const unsigned id = (unsigned)__float_as_int(d.x);
        if (__float_as_int(d.w) == 1 && len3(col) > 0.f && id < K.n_pix) { // TracerBase.h:396-400 -> IceTComposite::localAdd
          const V3 cw = scl3(col, d.z);
          float *px = K.fb + (size_t)4 * id;
          atomicAdd(px + 0, cw.x); atomicAdd(px + 1, cw.y); atomicAdd(px + 2, cw.z); atomicAdd(px + 3, 1.f);
        }
      }
    }
    const unsigned long long m = ballot64(go_on);
    if (m) {
      unsigned base = 0;
      if (lane_id() == 0) base = atomicAdd(out_count, (unsigned)__popcll(m));
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
      if (go_on) { const unsigned slot = base + lanes_below(m); out.p0[slot] = a; out.p1[slot] = b; out.p2[slot] = c; out.p3[slot] = d; if (out.p4) out.p4[slot] = 0u; store_no_known(out, slot); if (out_from) out_from[slot] = from; }
    }
  }
}

// MULTI (merged launch, gvt_internal.h): closest hit -- ray j is virtual index (idx ? idx[j] : j) of the segment table MS.W; any hit --
// ray j of q was generated in instance MS.ray_inst[j]; either way the lane takes transform and acceleration structure from
// MS.W.insts[instance].  out_from receives the source instance of every survivor appended to `out`.
// per-block LDS copies of the round's tables (<= KT_TAB segments and instances): a refill then costs one memory round trip -- the
// ray itself -- instead of a binary search over the segment table plus the segment's and the instance's rows, each a dependent load
#define KT_TAB 32
struct KtInst { Mat4 minv; const uint4 *nodes4; const float4 *tris; };
struct KtSeg { float4 *planes; unsigned long long cap; unsigned begin; int inst; };
struct MultiSrc {
  WaveSet W;
  const int *ray_inst;          // any hit: instance per ray of q; < 0 = an empty slot of a direct-mapped shadow list (skipped)
  int *out_from;
  unsigned long long *tot_any;  // any hit over a direct-mapped list: the rays actually traced are added here (one atomic per wave)
};
template <bool ANY, bool XFORM, int MODE, bool COOP, bool W4, bool MULTI = false>
__global__ __launch_bounds__(TRAV_BLOCK, (ANY ? KT_BLOCKS_ANY : KT_BLOCKS_CLOSEST)) void k_trace(RayPlanes q, const unsigned *__restrict__ idx, unsigned n, Mat4 minv, Trav T, float tnear,
                                                       gvt_hip_hit *__restrict__ hits, int *__restrict__ flags, RayPlanes out, unsigned *out_count,
                                                       unsigned *counter, int *spill_base, int refill_min, int inner_min, const unsigned *__restrict__ n_dev, int share, unsigned share_min, TermSink sink, LongQ LQ,
                                                       MultiSrc MS = MultiSrc{}) {
  // Work distribution: wave w first takes the static range [w*chunk, (w+1)*chunk) -- no atomic, see the refill below -- and after that
  // dynamic ranges of `dyn` rays from the counter.  chunk is a fraction of a wave's fair share (3/8 for closest-hit launches, whose
  // per-ray cost varies most, 5/8 for any-hit; measured at 1 M rays: 96/128/160 rays -> 0.542/0.549/0.579 ms closest, 0.400/0.400/0.384
  // any), never less than one wave's width; dyn is 1/16 of the share, at least 64 (32: the counter word saturates).
#ifdef GVT_EXPERIMENTS
  if (!ANY && (share & 2)) LQ.steps = 0; // closest-hit drain sharing (a rejected variant) and parking both go after a launch's last rays: one at a time
#endif
  // (refill_min / inner_min / share stay kernel arguments in the shipped library too, although only the experiments build can move them: as
  // compile-time constants they cost 1.5 % of the any-hit launch -- 0.3555 against 0.3500 ms, the register allocation changes)
  const unsigned n_waves_total = gridDim.x * (unsigned)(TRAV_BLOCK / 64);
  if (n_dev) n = *n_dev; // ray count produced by the previous kernel on this stream (no host round trip)
  const unsigned share_w = n / n_waves_total;
  const unsigned chunk = max(64u, ((share_w * (ANY ? 5u : 3u) / 8u) + 16u) & ~31u);
  const unsigned dyn = max(64u, (share_w / 16u) & ~63u);
  __shared__ int stack[TRAV_STACK * TRAV_BLOCK];
  int *lds = &stack[threadIdx.x];
  int *spill = spill_base + (size_t)(blockIdx.x * TRAV_BLOCK + threadIdx.x) * TRAV_SPILL;
  __shared__ KtInst s_inst[MULTI ? KT_TAB : 1];
  __shared__ KtSeg s_seg[(MULTI && !ANY) ? KT_TAB : 1];
  bool tab = false; // block-uniform
  if (MULTI) {
    tab = MS.W.n_inst > 0 && MS.W.n_inst <= KT_TAB && MS.W.n_seg <= KT_TAB;
    if (tab) {
      for (int t = (int)threadIdx.x; t < MS.W.n_inst; t += TRAV_BLOCK) { const WaveInst *wi = MS.W.insts + t; s_inst[t].minv = wi->minv; s_inst[t].nodes4 = wi->nodes4; s_inst[t].tris = wi->tris; }
      if (!ANY) // (strided from the far end of the block, so that with >= 64 threads other lanes than the instance rows' load these)
        for (int t = TRAV_BLOCK - 1 - (int)threadIdx.x; t < MS.W.n_seg; t += TRAV_BLOCK) { const WaveSeg sg = MS.W.segs[t]; KtSeg k; k.planes = sg.planes; k.cap = sg.cap; k.begin = sg.begin; k.inst = sg.inst; s_seg[t] = k; }
      __syncthreads();
    }
  }
#define KT_PUSH(REF)                                                                  \
  {                                                                                   \
    if (sp < TRAV_STACK) { lds[sp * TRAV_BLOCK] = (REF); sp++; }                      \
    else if (sp - TRAV_STACK < TRAV_SPILL) { spill[sp - TRAV_STACK] = (REF); sp++; }  \
    else atomicOr(counter + TRAV_OVF_WORD, 1u); /* reported, never silent */          \
  }
#define KT_POP()                                                                      \
  {                                                                                   \
    if (sp == sb) cur = TRAV_DONE;                                                    \
    else { sp--; if (sp < TRAV_STACK) cur = lds[sp * TRAV_BLOCK]; else cur = spill[sp - TRAV_STACK]; } \
  }
  __shared__ unsigned pend_all[(TRAV_BLOCK / 64) * 128];
  volatile unsigned *pend = &pend_all[(threadIdx.x >> 6) * 128];
  int n_pend = 0;                 // wave-uniform
  unsigned c_next = 0, c_end = 0; // wave-uniform
  bool exhausted = false;         // wave-uniform
  bool first_chunk = true;        // wave-uniform
  bool active = false;
  unsigned j = 0;
  V3 O = mk3(0, 0, 0), D = mk3(0, 0, 1);
  RaySlab S = make_slab(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
  float ox = 0, oy = 0, oz = 0; // O * inv_d as such: only the binary-tree variants (W4 false) read it
  float bt = GVT_FLT_MAX, bu = 0.f, bv = 0.f, bden = 1.f; // bu, bv: un-divided U, V of the best hit; bden its |den|
  int bp = -1, sp = 0, cur = TRAV_DONE;
  int sb = 0;           // bottom of this lane's stack window [sb, sp): entries below sb were given away to helper lanes
  bool sharing = false; // wave-uniform: some ray of this wave is being traversed by more than one lane
  int donor_lane = -1;  // helper: the lane it took its subtree from (to follow that lane's best hit)
  int nsteps = 0;       // closest hit: inner steps of this lane's ray; beyond LQ.steps the ray is parked for k_long_closest (nsteps = -1 from then on)
#define KT_PARKED (nsteps < 0)
  const uint4 *nodes4_l = T.nodes4; // MULTI: this lane's ray's instance
  const float4 *tris_l = T.tris;
  int inst_l = 0;
  unsigned gidx = 0;                // MULTI closest: virtual index of the lane's ray
  unsigned n_started = 0;           // MULTI any hit: rays this wave really traced (wave-uniform)
#if GVT_STAMP
  unsigned long long st_refill = 0, st_inner = 0, st_leaf = 0, st_retire = 0, n_inner_it = 0, n_outer_it = 0, t_mark = 0, t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_exh = 0, it_exh = 0, out_exh = 0, act_exh = 0, n_inner_lanes = 0, n_leaf_lanes = 0;
#endif
  for (;;) {
    // ---- refill idle lanes from the wave's private index range
#if GVT_STAMP == 1
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (n_outer_it) st_retire += t_ - t_mark; t_mark = t_; }
#endif
    unsigned long long idle = ballot64(!active);
    int nidle = __popcll(idle);
    if (!exhausted && (nidle >= refill_min || nidle == 64)) {
      while (nidle > 0) {
        if (c_next == c_end) {
          // the first chunk of every wave is assigned statically (chunk number = wave number): 4096 waves hitting one counter word at
          // launch would queue for ~45 us (a single word sustains ~90 atomics/us); the dynamic chunks start behind those
          unsigned base = 0;
          unsigned this_chunk = chunk;
          if (first_chunk) {
            first_chunk = false;
            base = (blockIdx.x * (unsigned)(TRAV_BLOCK / 64) + (threadIdx.x >> 6)) * chunk;
          } else {
            if (lane_id() == 0) base = atomicAdd(counter, dyn);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + n_waves_total * chunk;
            this_chunk = dyn;
          }
          if (base >= n) {
            exhausted = true;
#if GVT_STAMP
            if (!t_exh) { t_exh = __builtin_amdgcn_s_memtime(); it_exh = n_inner_it; out_exh = n_outer_it; act_exh = 64 - nidle; }
#endif
            break;
          }
          c_next = base;
          c_end = min(base + this_chunk, n);
        }
        const unsigned take = min(c_end - c_next, (unsigned)nidle);
        const unsigned rank = lanes_below(idle);
        bool start = false;
        if (!active && rank < take) {
          j = c_next + rank;
          const unsigned i = idx ? idx[j] : j;
          float4 a, b;
          start = true;
          if (MULTI) {
            if (ANY) {
              inst_l = MS.ray_inst[i];
              start = inst_l >= 0; // direct-mapped shadow list: a slot whose primary emitted nothing
              if (start) { a = q.p0[i]; b = q.p1[i]; }
            } else if (tab) {
              int lo = 0, hi = MS.W.n_seg - 1;
              while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_seg[mid].begin <= i) lo = mid; else hi = mid - 1; }
              const unsigned local = i - s_seg[lo].begin;
              const float4 *pl = s_seg[lo].planes;
              a = pl[local]; b = pl[s_seg[lo].cap + local];
              inst_l = s_seg[lo].inst; gidx = i;
            } else {
              const WaveSeg sg = MS.W.segs[wave_find_seg(MS.W, i)];
              const unsigned local = i - sg.begin;
              a = sg.planes[local]; b = sg.planes[sg.cap + local];
              inst_l = sg.inst; gidx = i;
            }
            if (start && tab) {
              nodes4_l = s_inst[inst_l].nodes4; tris_l = s_inst[inst_l].tris;
              O = xfm_point(s_inst[inst_l].minv, mk3(a.x, a.y, a.z)); D = xfm_vector(s_inst[inst_l].minv, mk3(b.x, b.y, b.z));
            } else if (start) {
              const WaveInst *wi = MS.W.insts + inst_l;
              nodes4_l = wi->nodes4; tris_l = wi->tris;
              O = xfm_point(wi->minv, mk3(a.x, a.y, a.z)); D = xfm_vector(wi->minv, mk3(b.x, b.y, b.z));
            }
          } else {
            a = q.p0[i]; b = q.p1[i];
            O = mk3(a.x, a.y, a.z); D = mk3(b.x, b.y, b.z);
            if (XFORM) { O = xfm_point(minv, O); D = xfm_vector(minv, D); }
          }
          if (start) {
            const float dx = fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x;
            const float dy = fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y;
            const float dz = fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z;
            const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
            ox = O.x * ix; oy = O.y * iy; oz = O.z * iz;
            S = make_slab(ix, iy, iz, ox, oy, oz);
            bt = GVT_FLT_MAX; bu = 0.f; bv = 0.f; bden = 1.f; bp = -1;
            sp = 0; sb = 0; donor_lane = -1; nsteps = 0;
            cur = (MULTI ? (nodes4_l != nullptr) : (W4 ? (T.nodes4 != nullptr) : (T.nodes != nullptr))) ? 0 : TRAV_DONE;
            active = true;
          }
        }
        if (MULTI && ANY) n_started += (unsigned)__popcll(ballot64(start));
        c_next += take;
        idle = ballot64(!active);
        nidle = __popcll(idle);
      }
    }
    if (nidle == 64) break; // nothing left in flight and nothing left to fetch
    // ---- drain phase: work sharing inside the wave.  Once the work counter is exhausted, a wave used to finish at the pace of
    //      its slowest ray (measured: up to 536 more inner steps at ~2.5 K cycles each while 63 lanes idle -- the fixed ~0.5 ms of
    //      every launch).  Now an idle lane takes the BOTTOM entry (the largest pending subtree) of a busy lane's stack together with a
    //      copy of its ray and best hit, and traverses that subtree as a helper; results are merged when lanes of a ray retire.
#ifdef GVT_EXPERIMENTS
    const bool share_on = ANY ? (share & 1) != 0 : (share & 2) != 0;
#else
    const bool share_on = ANY && (share & 1) != 0; // sharing for closest hit lost: compiled into the experiments build only
#endif
    if (share_on && exhausted && nidle > 0 && n >= share_min) { // small launches: the hand-off costs more than the tail it trims (15 K shadow rays: 83 -> 66 us)
      unsigned long long idle_m = idle;
      unsigned long long don_m = ballot64(active && cur != TRAV_DONE && (sp - sb) >= 2);
      const unsigned wave_tid0 = threadIdx.x & ~63u;
      for (int pairs = 0; idle_m && don_m && pairs < 16; pairs++) {
        const int h = __ffsll((long long)idle_m) - 1, d = __ffsll((long long)don_m) - 1;
        // the bottom entry = the largest pending subtree.  For the closest hit those are the far siblings near the root, which the
        // owner's eventual hit usually prunes: helpers therefore keep pulling the donor's current best distance (below).
        const int sbd = __shfl(sb, d);
        const unsigned gj = (unsigned)__shfl((int)j, d);
        const float gOx = __shfl(O.x, d), gOy = __shfl(O.y, d), gOz = __shfl(O.z, d);
        const float gDx = __shfl(D.x, d), gDy = __shfl(D.y, d), gDz = __shfl(D.z, d);
        const float gox = __shfl(ox, d), goy = __shfl(oy, d), goz = __shfl(oz, d);
        RaySlab gS;
        gS.ix = __shfl(S.ix, d); gS.iy = __shfl(S.iy, d); gS.iz = __shfl(S.iz, d);
        gS.ox = (f2){ __shfl(S.ox.x, d), __shfl(S.ox.y, d) }; gS.oy = (f2){ __shfl(S.oy.x, d), __shfl(S.oy.y, d) }; gS.oz = (f2){ __shfl(S.oz.x, d), __shfl(S.oz.y, d) };
        const float gbt = __shfl(bt, d), gbu = __shfl(bu, d), gbv = __shfl(bv, d), gbden = __shfl(bden, d);
        const int gbp = __shfl(bp, d);
        const int ginst = __shfl(inst_l, d);
        const unsigned ggidx = (unsigned)__shfl((int)gidx, d);
        if ((int)lane_id() == h) {
          if (MULTI) { inst_l = ginst; gidx = ggidx; nodes4_l = MS.W.insts[ginst].nodes4; tris_l = MS.W.insts[ginst].tris; }
          const unsigned tid_d = wave_tid0 + (unsigned)d;
          cur = (sbd < TRAV_STACK) ? stack[sbd * TRAV_BLOCK + tid_d]
                                   : spill_base[((size_t)blockIdx.x * TRAV_BLOCK + tid_d) * TRAV_SPILL + (sbd - TRAV_STACK)];
          j = gj; O = mk3(gOx, gOy, gOz); D = mk3(gDx, gDy, gDz);
          S = gS; ox = gox; oy = goy; oz = goz;
          bt = gbt; bu = gbu; bv = gbv; bden = gbden; bp = gbp; // the donor's best so far: a pruning bound, merged idempotently later
          sp = 0; sb = 0;
          donor_lane = d;
          nsteps = 0; // a fresh share of the ray: not the step count / parked state of the lane's previous ray
          active = true;
        }
        if ((int)lane_id() == d) sb++;
        idle_m &= idle_m - 1;
        don_m &= don_m - 1; // one entry per donor and round: spread the helpers over the busy lanes
        sharing = true;
      }
    }
#if GVT_STAMP == 1
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_refill += t_ - t_mark; t_mark = t_; n_outer_it++; }
#endif
    if (!ANY && sharing) { // helpers follow their donor's best hit: a closer hit found by the owner prunes the helper's subtree too
      const int dl = donor_lane >= 0 ? donor_lane : (int)lane_id();
      const unsigned dj = (unsigned)__shfl((int)j, dl);
      const float dt = __shfl(bt, dl), du = __shfl(bu, dl), dv = __shfl(bv, dl), dd = __shfl(bden, dl);
      const int dp = __shfl(bp, dl);
      if (active && donor_lane >= 0 && dj == j && dp >= 0 && (bp < 0 || dt < bt || (dt == bt && dp < bp))) { bt = dt; bp = dp; bu = du; bv = dv; bden = dd; }
    }
    // ---- inner nodes: the lanes holding one descend level by level in a tight loop; the loop is left as soon as
    //      fewer than inner_min lanes still descend (the others wait at a leaf, have finished, or are idle), so that
    //      both this loop and the dearer leaf phase below run at high lane utilisation.
    unsigned long long im = ballot64(active && cur >= 0);
    while (im) {
      const bool at_inner = active && cur >= 0;
      if (W4) {
        if (at_inner) { // one 64-byte fetch decides four children (8-bit boxes on the node's own grid)
          float tn[4];
          int rr[4];
          bool entered[4];
          node4_test((MULTI ? nodes4_l : T.nodes4) + (size_t)GVT_NODE4_F4 * cur, S, ANY ? GVT_FLT_MAX : bt, tn, rr, ANY ? entered : nullptr);
#define KT_ENTERED(K) (ANY ? entered[K] : (tn[K] < GVT_FLT_MAX)) // (closest hit: tn / rr are sorted below, the flags are not)
          if (!ANY) { // nearest first; for any-hit the order does not matter
#define GVT_CE(A, B) { const bool sw_ = tn[B] < tn[A]; const float ta_ = sw_ ? tn[B] : tn[A], tb_ = sw_ ? tn[A] : tn[B]; \
                       const int ra_ = sw_ ? rr[B] : rr[A], rb_ = sw_ ? rr[A] : rr[B]; tn[A] = ta_; tn[B] = tb_; rr[A] = ra_; rr[B] = rb_; }
            GVT_CE(0, 1) GVT_CE(2, 3) GVT_CE(0, 2) GVT_CE(1, 3) GVT_CE(1, 2)
#undef GVT_CE
          }
          // push the hit children farthest first (closest hit: sorted), continue with the first hit one
          int nxt = TRAV_DONE;
          bool have = false;
          if (sp + 3 <= TRAV_STACK) { // room for three entries in the LDS part: no bounds checks, no spill path
#pragma unroll
            for (int c4 = 3; c4 >= 0; c4--) {
              if (KT_ENTERED(c4)) {
                if (have) { lds[sp * TRAV_BLOCK] = nxt; sp++; }
                nxt = rr[c4]; have = true;
              }
            }
          } else {
#pragma unroll
            for (int c4 = 3; c4 >= 0; c4--) {
              if (KT_ENTERED(c4)) {
                if (have) KT_PUSH(nxt)
                nxt = rr[c4]; have = true;
              }
            }
          }
          if (have) cur = nxt;
          else KT_POP()
#undef KT_ENTERED
        }
      }
#ifdef GVT_EXPERIMENTS
      else {
#include "experiments/binary_node_arm.inc"
      }
#endif
#if GVT_STAMP == 1
      n_inner_it++; n_inner_lanes += (unsigned long long)__popcll(ballot64(at_inner));
#endif
      if (!ANY && at_inner && LQ.steps && ++nsteps > (exhausted ? LQ.steps_drain : LQ.steps) && cur != TRAV_DONE) { KT_PUSH(cur) cur = TRAV_DONE; nsteps = -1; } // the pending stack, `cur` on top, goes into the record
      im = ballot64(active && cur >= 0);
      if (__popcll(im) < inner_min) break;
    }
#if GVT_STAMP == 1
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_inner += t_ - t_mark; t_mark = t_; }
#endif
    // ---- leaves: every lane waiting at one intersects its triangles (64-byte slots: v0|prim, e1, e2, Ng).
    //      Measured with s_memtime stamps: a leaf phase that fetches and tests one triangle after the other costs ~6 K
    //      cycles (four dependent round trips at the loaded memory latency), an inner step ~1.9 K.  So the triangles are
    //      fetched two per round trip (48 of the 64 slot bytes each; fetching all four at once needs > 128 VGPRs and costs
    //      a wave per SIMD).
    {
      const bool at_leaf = active && cur < 0 && cur != TRAV_DONE;
#if GVT_STAMP == 1
      n_leaf_lanes += (unsigned long long)__popcll(ballot64(at_leaf));
#endif
#ifdef GVT_EXPERIMENTS
      if (COOP) {
#include "experiments/coop_leaf_arm.inc"
      } else
#endif
      if (at_leaf) {
        const unsigned code = (unsigned)~cur;
        const unsigned first = code >> 3, ntri = code & 7u;
        const float4 *ts = (MULTI ? tris_l : T.tris) + 4 * (size_t)first;
        bool occluded = false;
        for (unsigned kb = 0; kb < ntri && !(ANY && occluded); kb += 2) { // two triangles per round trip
          float4 s0[2], s1[2], s2[2];
#pragma unroll
          for (unsigned k = 0; k < 2; k++)
            if (kb + k < ntri) { s0[k] = ts[4 * (kb + k)]; s1[k] = ts[4 * (kb + k) + 1]; s2[k] = ts[4 * (kb + k) + 2]; }
#pragma unroll
          for (unsigned k = 0; k < 2; k++) {
            if (kb + k < ntri && !(ANY && occluded)) {
              const V3 e1 = mk3(s1[k].x, s1[k].y, s1[k].z), e2 = mk3(s2[k].x, s2[k].y, s2[k].z);
              float TT, U, V, aden;
              if (tri_test_raw(O, D, mk3(s0[k].x, s0[k].y, s0[k].z), e1, e2, cross3(e1, e2), tnear, TT, U, V, aden)) {
                const float t = TT / aden;
                if (t <= GVT_FLT_MAX) {
                  if (ANY) occluded = true;
                  else {
                    const int prim = __float_as_int(s0[k].w);
                    if (bp < 0 || t < bt || (t == bt && prim < bp)) { bt = t; bp = prim; bu = U; bv = V; bden = aden; }
                  }
                }
              }
            }
          }
        }
        if (ANY && occluded) bp = 0;
      }
      if (at_leaf) {
        if (ANY && bp == 0) cur = TRAV_DONE;
        else KT_POP()
      }
    }
#if GVT_STAMP == 1
    { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_leaf += t_ - t_mark; t_mark = t_; }
#endif
    // ---- retire finished rays.  Retirement is batched: a finished lane costs nothing while it waits, so the block below (result
    //      stores with their two divisions, the parked-ray and survivor lists) runs only when a refill is due, when nothing is left
    //      in flight, or -- while lanes of a drained wave share rays -- at once, because a finished helper's result ends its group
    {
      const int nfin_w = __popcll(ballot64(active && cur == TRAV_DONE)), nidle_w = __popcll(ballot64(!active));
      const bool retire_now = RETIRE_BATCH == 0 || (share_on && exhausted) || nfin_w + nidle_w >= (exhausted ? 64 : refill_min);
      if (!retire_now) continue;
    }
    if (sharing) { // lanes of one ray: the last one to finish carries the merged result, the others fold theirs into a partner
      const unsigned long long FM = ballot64(active && cur == TRAV_DONE);
      unsigned long long fm = FM;
      while (fm) {
        const int f = __ffsll((long long)fm) - 1;
        const unsigned jf = (unsigned)__shfl((int)j, f);
        const unsigned long long G = ballot64(active && j == jf);
        const unsigned long long Gf = G & FM, A = G & ~FM;
        const int tgt = A ? __ffsll((long long)A) - 1 : __ffsll((long long)Gf) - 1;
        unsigned long long src = Gf & ~(1ull << tgt);
        bool merged_occluded = false;
        while (src) {
          const int sidx = __ffsll((long long)src) - 1;
          const float st = __shfl(bt, sidx), su = __shfl(bu, sidx), sv = __shfl(bv, sidx), sd = __shfl(bden, sidx);
          const int spr = __shfl(bp, sidx);
          const bool spk = __shfl(nsteps, sidx) < 0;
          if (!ANY && (int)lane_id() == tgt && spk) nsteps = -1; // a lane that gave up on its share: the whole ray goes to k_long_closest
          if (ANY) { if (spr >= 0) merged_occluded = true; }
          else if ((int)lane_id() == tgt && spr >= 0 && (bp < 0 || st < bt || (st == bt && spr < bp))) { bt = st; bp = spr; bu = su; bv = sv; bden = sd; }
          if ((int)lane_id() == sidx) active = false; // folded into tgt: retires without writing
          src &= src - 1;
        }
        if (ANY && merged_occluded && ((G >> lane_id()) & 1ull) && active) { bp = 0; cur = TRAV_DONE; } // one occluder ends the whole group
        fm &= ~Gf;
      }
    }
    const bool fin = active && cur == TRAV_DONE;
    if (ANY && MODE == 1) {
      // un-occluded shadow rays go to moved_rays.  Their indices are parked in a per-wave LDS list and flushed 64+
      // at a time: one atomic on the queue counter per flush instead of one per retirement (a single counter
      // word sustains only ~90 atomics/us chip-wide).
      const bool survive = fin && bp < 0;
      const unsigned long long sm = ballot64(survive);
      if (sm) {
        if (survive) pend[n_pend + lanes_below(sm)] = j;
        n_pend += __popcll(sm);
      }
      if (n_pend >= 64) { flush_pending(pend, n_pend, q, out, out_count, sink, MULTI ? MS.ray_inst : nullptr, MULTI ? MS.out_from : nullptr); n_pend = 0; }
    }
    if (!ANY && LQ.steps) {
      const unsigned long long pm = ballot64(fin && KT_PARKED);
      if (pm) {
        unsigned base = 0;
        if ((int)lane_id() == __ffsll((long long)pm) - 1) base = atomicAdd(LQ.count, (unsigned)__popcll(pm));
        base = __shfl(base, __ffsll((long long)pm) - 1);
        if (fin && KT_PARKED) {
          const unsigned slot = base + lanes_below(pm);
          LongRec R; R.j = j; R.i = MULTI ? gidx : (idx ? idx[j] : j); R.bt = bt; R.bp = bp; R.bu = bu; R.bv = bv; R.bden = bden; R.ns = 0u;
          const int depth = sp - sb;
          if (LQ.stk && !sharing && depth > 0 && depth <= LONG_SAVE && slot < LQ.stk_cap) { // a shared ray's windows are not one stack: it starts again
            int *dst = LQ.stk + (size_t)slot * LONG_SAVE;
            for (int k = 0; k < depth; k++) dst[k] = (sb + k < TRAV_STACK) ? lds[(sb + k) * TRAV_BLOCK] : spill[sb + k - TRAV_STACK];
            R.ns = (unsigned)depth;
          }
          LQ.recs[slot] = R;
        }
      }
    }
    if (fin) {
      if (ANY) { if (MODE == 0) flags[j] = (bp >= 0) ? 1 : 0; }
      else if (!KT_PARKED) { gvt_hip_hit h; h.t = bt; h.prim = bp; h.u = (bp >= 0) ? bu / bden : 0.f; h.v = (bp >= 0) ? bv / bden : 0.f; hits[j] = h; }
      active = false;
    }
  }
#if GVT_STAMP
  if (lane_id() == 0) { atomicAdd(&g_stamp[0], st_refill); atomicAdd(&g_stamp[1], st_inner); atomicAdd(&g_stamp[2], st_leaf); atomicAdd(&g_stamp[3], st_retire); atomicAdd(&g_stamp[4], n_inner_it); atomicAdd(&g_stamp[5], n_outer_it); atomicAdd(&g_stamp[6], 1ull); atomicAdd(&g_stamp[7], (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin); atomicAdd(&g_stamp[14], n_inner_lanes); atomicAdd(&g_stamp[15], n_leaf_lanes); atomicMax(&g_stamp[16], ~t_begin); atomicMax(&g_stamp[17], (unsigned long long)__builtin_amdgcn_s_memtime()); atomicMax(&g_stamp[18], t_exh); atomicAdd(&g_stamp[19], t_exh ? t_exh - t_begin : 0ull); atomicMax(&g_stamp[20], (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin); atomicMax(&g_stamp[21], t_exh ? t_exh - t_begin : 0ull); atomicMax(&g_stamp[22], ~(t_exh ? t_exh - t_begin : ~0ull));
    if (t_exh) { atomicAdd(&g_stamp[8], (unsigned long long)__builtin_amdgcn_s_memtime() - t_exh); atomicAdd(&g_stamp[9], n_inner_it - it_exh); atomicAdd(&g_stamp[10], n_outer_it - out_exh); atomicAdd(&g_stamp[11], act_exh); atomicMax(&g_stamp[12], (unsigned long long)__builtin_amdgcn_s_memtime() - t_exh); atomicMax(&g_stamp[13], n_inner_it - it_exh); } }
#endif
  if (ANY && MODE == 1) { if (n_pend) flush_pending(pend, n_pend, q, out, out_count, sink, MULTI ? MS.ray_inst : nullptr, MULTI ? MS.out_from : nullptr); }
  if (MULTI && ANY && MS.tot_any && n_started && lane_id() == 0) atomicAdd(MS.tot_any, (unsigned long long)n_started);
#undef KT_PUSH
#undef KT_POP
#undef KT_PARKED
}

#ifdef GVT_EXPERIMENTS
#include "experiments/quad_kernel.inc" // k_traceq: four lanes per ray (measured slower: VALU bound)
#endif

// A whole wave per parked ray.  The pending nodes live in a per-wave LDS list; each step the 64 lanes open up to 64 of them (newest
// first), append the children the ray enters to the node list or the leaf list, and when enough leaves have gathered (or no node is
// left) every lane intersects one leaf and the wave reduces to the best (t, primID).  Entries farther than the best hit are dropped
// when they are taken.  LONG_CAP throttles the number of nodes opened per step so that the lists cannot outgrow LONG_PHYS.
#define LONG_CAP 512
#define LONG_PHYS (LONG_CAP + 256)

// The traversal of ONE ray by a whole wave (every lane holds the same O, D): shared by k_long_closest, k_wave_any and k_finish.
// The caller has put the pending nodes / leaves into the wave's LDS lists (ns / nl entries).  A node step opens up to 64 nodes and
// appends the children the ray enters (nearest last: the lists are taken from their end); `take` is throttled so that the lists stay
// within CAP, and PHYS = CAP + 256 leaves room for the one step that may exceed it -- the bound is checked ONCE per step for the
// wave (a list that would outgrow PHYS: flag word set, the ray's traversal ends; reported, never silent), not per store.
template <int CAP, int PHYS>
__device__ __forceinline__ void wave_closest_run(const uint4 *__restrict__ nodes4, const float4 *__restrict__ tris, V3 O, V3 D, const RaySlab &S, float tnear,
                                                  volatile int *s_ref, volatile float *s_tn, volatile int *l_ref, volatile float *l_tn, int ns, int nl,
                                                  float &bt, int &bp, float &bu, float &bv, float &bden, unsigned *ovf_word) {
  const int lane = (int)lane_id();
  while (ns > 0 || nl > 0) {
    const bool do_leaf = nl > 0 && (ns == 0 || nl >= 64 || nl > CAP - 256);
    if (!do_leaf) {
      int take = min(min(ns, 64), min((CAP - ns) / 3, (CAP - nl) / 4));
      take = max(take, 1);
      if (ns + 3 * take > PHYS || nl + 4 * take > PHYS) { if (lane == 0) atomicOr(ovf_word, 1u); break; }
      const bool mine = lane < take;
      int ref = 0;
      float etn = 0.f;
      if (mine) { ref = s_ref[ns - 1 - lane]; etn = s_tn[ns - 1 - lane]; }
      __builtin_amdgcn_wave_barrier();
      ns -= take;
      float tn[4];
      int rr[4];
      const bool open = mine && etn <= bt;
      if (open) {
        node4_test(nodes4 + (size_t)GVT_NODE4_F4 * ref, S, bt, tn, rr);
#define GVT_CE(A, B) { const bool sw_ = tn[B] < tn[A]; const float ta_ = sw_ ? tn[B] : tn[A], tb_ = sw_ ? tn[A] : tn[B]; \
                       const int ra_ = sw_ ? rr[B] : rr[A], rb_ = sw_ ? rr[A] : rr[B]; tn[A] = ta_; tn[B] = tb_; rr[A] = ra_; rr[B] = rb_; }
        GVT_CE(0, 1) GVT_CE(2, 3) GVT_CE(0, 2) GVT_CE(1, 3) GVT_CE(1, 2)
#undef GVT_CE
      }
#pragma unroll
      for (int c = 3; c >= 0; c--) {
        const bool hit = open && tn[c] < GVT_FLT_MAX;
        const bool inner = hit && rr[c] >= 0, leaf = hit && rr[c] < 0;
        const unsigned long long mi = ballot64(inner), ml = ballot64(leaf);
        if (inner) { const int pos = ns + (int)lanes_below(mi); s_ref[pos] = rr[c]; s_tn[pos] = tn[c]; }
        if (leaf) { const int pos = nl + (int)lanes_below(ml); l_ref[pos] = rr[c]; l_tn[pos] = tn[c]; }
        ns += __popcll(mi);
        nl += __popcll(ml);
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      const int take = min(nl, 64);
      const bool mine = lane < take;
      int ref = -1;
      float etn = 0.f;
      if (mine) { ref = l_ref[nl - 1 - lane]; etn = l_tn[nl - 1 - lane]; }
      __builtin_amdgcn_wave_barrier();
      nl -= take;
      float lt = GVT_FLT_MAX, lu = 0.f, lv = 0.f, ld = 1.f;
      int lp = -1;
      if (mine && etn <= bt) {
        const unsigned code = (unsigned)~ref;
        const unsigned first = code >> 3, ntri = code & 7u;
        const float4 *ts = tris + 4 * (size_t)first;
        for (unsigned k = 0; k < ntri; k++) {
          const float4 s0 = ts[4 * k], s1 = ts[4 * k + 1], s2 = ts[4 * k + 2];
          const V3 e1 = mk3(s1.x, s1.y, s1.z), e2 = mk3(s2.x, s2.y, s2.z);
          float TT, U, V, aden;
          if (tri_test_raw(O, D, mk3(s0.x, s0.y, s0.z), e1, e2, cross3(e1, e2), tnear, TT, U, V, aden)) {
            const float t = TT / aden;
            if (t <= GVT_FLT_MAX) {
              const int prim = __float_as_int(s0.w);
              if (lp < 0 || t < lt || (t == lt && prim < lp)) { lt = t; lp = prim; lu = U; lv = V; ld = aden; }
            }
          }
        }
      }
      if (ballot64(lp >= 0)) { // the wave's best candidate, then against the ray's best so far
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const float ot = __shfl_xor(lt, off), ou = __shfl_xor(lu, off), ov = __shfl_xor(lv, off), od = __shfl_xor(ld, off);
          const int op = __shfl_xor(lp, off);
          if (op >= 0 && (lp < 0 || ot < lt || (ot == lt && op < lp))) { lt = ot; lp = op; lu = ou; lv = ov; ld = od; }
        }
        if (lp >= 0 && (bp < 0 || lt < bt || (lt == bt && lp < bp))) { bt = lt; bp = lp; bu = lu; bv = lv; bden = ld; }
      }
    }
  }
}

// any hit of one ray by a whole wave, from the root: true = occluded
template <int CAP, int PHYS>
__device__ __forceinline__ bool wave_any_run(const uint4 *__restrict__ nodes4, const float4 *__restrict__ tris, V3 O, V3 D, const RaySlab &S, float tnear,
                                              volatile int *s_ref, volatile int *l_ref, unsigned *ovf_word) {
  const int lane = (int)lane_id();
  int ns = nodes4 ? 1 : 0, nl = 0; // wave-uniform
  bool occluded = false;           // wave-uniform
  if (lane == 0) s_ref[0] = 0;
  __builtin_amdgcn_wave_barrier();
  while (!occluded && (ns > 0 || nl > 0)) {
    const bool do_leaf = nl > 0 && (ns == 0 || nl >= 64 || nl > CAP - 256);
    if (!do_leaf) {
      int take = min(min(ns, 64), min((CAP - ns) / 3, (CAP - nl) / 4));
      take = max(take, 1);
      if (ns + 3 * take > PHYS || nl + 4 * take > PHYS) { if (lane == 0) atomicOr(ovf_word, 1u); break; }
      const bool mine = lane < take;
      int ref = 0;
      if (mine) ref = s_ref[ns - 1 - lane];
      __builtin_amdgcn_wave_barrier();
      ns -= take;
      float tn[4];
      int rr[4];
      bool entered[4] = { false, false, false, false };
      if (mine) node4_test(nodes4 + (size_t)GVT_NODE4_F4 * ref, S, GVT_FLT_MAX, tn, rr, entered);
#pragma unroll
      for (int c = 3; c >= 0; c--) {
        const bool hit = mine && entered[c];
        const bool inner = hit && rr[c] >= 0, leaf = hit && rr[c] < 0;
        const unsigned long long mi = ballot64(inner), ml = ballot64(leaf);
        if (inner) s_ref[ns + (int)lanes_below(mi)] = rr[c];
        if (leaf) l_ref[nl + (int)lanes_below(ml)] = rr[c];
        ns += __popcll(mi);
        nl += __popcll(ml);
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      const int take = min(nl, 64);
      const bool mine = lane < take;
      int ref = -1;
      if (mine) ref = l_ref[nl - 1 - lane];
      __builtin_amdgcn_wave_barrier();
      nl -= take;
      bool hit_any = false;
      if (mine) {
        const unsigned code = (unsigned)~ref;
        const unsigned first_slot = code >> 3, ntri = code & 7u;
        const float4 *ts = tris + 4 * (size_t)first_slot;
        for (unsigned k = 0; k < ntri && !hit_any; k++) {
          const float4 s0 = ts[4 * k], s1 = ts[4 * k + 1], s2 = ts[4 * k + 2];
          const V3 e1 = mk3(s1.x, s1.y, s1.z), e2 = mk3(s2.x, s2.y, s2.z);
          float TT, U, V, aden;
          if (tri_test_raw(O, D, mk3(s0.x, s0.y, s0.z), e1, e2, cross3(e1, e2), tnear, TT, U, V, aden)) {
            const float t = TT / aden;
            if (t <= GVT_FLT_MAX) hit_any = true;
          }
        }
      }
      occluded = ballot64(hit_any) != 0ull;
    }
  }
  return occluded;
}
__device__ __forceinline__ RaySlab slab_of(V3 O, V3 D) {
  const float dx = fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x;
  const float dy = fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y;
  const float dz = fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z;
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
  return make_slab(ix, iy, iz, O.x * ix, O.y * iy, O.z * iz);
}

template <bool XFORM, bool MULTI = false>
__global__ __launch_bounds__(256) void k_long_closest(RayPlanes q, const LongRec *__restrict__ recs, const unsigned *__restrict__ n_recs, Mat4 minv,
                                                       Trav T, float tnear, gvt_hip_hit *__restrict__ hits, unsigned *counter, WaveSet W = WaveSet{},
                                                       const int *__restrict__ stk = nullptr) {
  __shared__ int s_ref_all[4][LONG_PHYS];
  __shared__ float s_tn_all[4][LONG_PHYS];
  __shared__ int l_ref_all[4][LONG_PHYS];
  __shared__ float l_tn_all[4][LONG_PHYS];
  const int wv = threadIdx.x >> 6;
  const int lane = (int)lane_id();
  volatile int *s_ref = s_ref_all[wv];
  volatile float *s_tn = s_tn_all[wv];
  volatile int *l_ref = l_ref_all[wv];
  volatile float *l_tn = l_tn_all[wv];
  const unsigned n = *n_recs;
  bool first = true;
  for (;;) {
    unsigned r = 0;
    if (first) { // first ray of a wave: its own number; later ones through the counter, behind those
      first = false;
      r = blockIdx.x * 4u + (unsigned)wv;
    } else {
      if (lane == 0) r = atomicAdd(counter, 1u);
      r = (unsigned)__builtin_amdgcn_readfirstlane((int)r) + gridDim.x * 4u;
    }
    if (r >= n) break;
    const LongRec R = recs[r];
    float4 a, b;
    V3 O, D;
    if (MULTI) {
      const WaveSeg sg = W.segs[wave_find_seg(W, R.i)];
      const unsigned local = R.i - sg.begin;
      a = sg.planes[local]; b = sg.planes[sg.cap + local];
      const WaveInst *wi = W.insts + sg.inst;
      T.nodes4 = wi->nodes4; T.tris = wi->tris;
      O = xfm_point(wi->minv, mk3(a.x, a.y, a.z)); D = xfm_vector(wi->minv, mk3(b.x, b.y, b.z));
    } else {
      a = q.p0[R.i]; b = q.p1[R.i];
      O = mk3(a.x, a.y, a.z); D = mk3(b.x, b.y, b.z);
      if (XFORM) { O = xfm_point(minv, O); D = xfm_vector(minv, D); }
    }
    const RaySlab S = slab_of(O, D);
    float bt = R.bt, bu = R.bu, bv = R.bv, bden = R.bden; // the same in every lane
    int bp = R.bp;
    int ns = T.nodes4 ? 1 : 0, nl = 0;                    // wave-uniform (an instance whose mesh has no nodes: the ray retires as a miss)
    if (stk && R.ns && T.nodes4) { // go on from the parked ray's pending stack (bottom first, so the nearest entries are taken first); entry distances unknown: 0
      const int e = lane < (int)R.ns ? stk[(size_t)r * LONG_SAVE + lane] : TRAV_DONE;
      const bool is_node = lane < (int)R.ns && e >= 0, is_leaf = lane < (int)R.ns && e < 0 && e != TRAV_DONE;
      const unsigned long long mi = ballot64(is_node), ml = ballot64(is_leaf);
      if (is_node) { s_ref[lanes_below(mi)] = e; s_tn[lanes_below(mi)] = 0.f; }
      if (is_leaf) { l_ref[lanes_below(ml)] = e; l_tn[lanes_below(ml)] = 0.f; }
      ns = __popcll(mi); nl = __popcll(ml);
    } else if (lane == 0) { s_ref[0] = 0; s_tn[0] = 0.f; }
    __builtin_amdgcn_wave_barrier();
    wave_closest_run<LONG_CAP, LONG_PHYS>(T.nodes4, T.tris, O, D, S, tnear, s_ref, s_tn, l_ref, l_tn, ns, nl, bt, bp, bu, bv, bden, counter + (TRAV_OVF_WORD - 4));
    // Every lane stores the (same) result.  A lane-0-only block as the LAST statement of a loop whose header takes the next ray with
    // readfirstlane lets hipcc send lane 0 and the other 63 lanes round the loop separately (seen in an experiment that finished the
    // parked rays inside k_trace: the 63 lanes then read a ticket no lane had taken and traced the same record for ever).
    { gvt_hip_hit h; h.t = bt; h.prim = bp; h.u = (bp >= 0) ? bu / bden : 0.f; h.v = (bp >= 0) ? bv / bden : 0.f; hits[R.j] = h; }
  }
}

// Small launches.  A persistent one-lane-per-ray launch cannot be faster than its slowest ray's chain of dependent fetches
// (100-350 steps of ~1 us): a round that holds a few hundred rays -- every later round of a multi-domain frame -- cost ~150 us
// per traversal launch whatever its size.  Below `small_rays` the rounds therefore give EVERY ray a whole wave (the k_long_closest
// scheme: 64 pending nodes opened per step, ~15-25 steps per ray): k_long_seed turns the ray list into LongRecs, k_long_closest
// finds the closest hits, k_wave_any below is the same traversal for shadow rays (stops at the first occluder).
__global__ __launch_bounds__(256) void k_long_seed(LongRec *__restrict__ recs, unsigned *__restrict__ count, const unsigned *__restrict__ idx, unsigned n,
                                                    const unsigned *__restrict__ n_dev) {
  if (n_dev) n = *n_dev;
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) *count = n;
  if (j >= n) return;
  LongRec R; R.j = j; R.i = idx ? idx[j] : j; R.bt = GVT_FLT_MAX; R.bp = -1; R.bu = 0.f; R.bv = 0.f; R.bden = 1.f; R.ns = 0u;
  recs[j] = R;
}

template <bool MULTI>
__global__ __launch_bounds__(256) void k_wave_any(RayPlanes q, const unsigned *__restrict__ n_dev, Mat4 minv, Trav T, float tnear, RayPlanes out,
                                                   unsigned *out_count, unsigned *counter, TermSink sink, MultiSrc MS) {
  __shared__ int s_ref_all[4][LONG_PHYS];
  __shared__ int l_ref_all[4][LONG_PHYS];
  const int wv = threadIdx.x >> 6;
  const int lane = (int)lane_id();
  volatile int *s_ref = s_ref_all[wv];
  volatile int *l_ref = l_ref_all[wv];
  const unsigned n = *n_dev;
  bool first = true;
  for (;;) {
    unsigned r = 0;
    if (first) { first = false; r = blockIdx.x * 4u + (unsigned)wv; }
    else {
      if (lane == 0) r = atomicAdd(counter, 1u);
      r = (unsigned)__builtin_amdgcn_readfirstlane((int)r) + gridDim.x * 4u;
    }
    if (r >= n) break;
    const float4 a = q.p0[r], b = q.p1[r];
    int inst = sink.from;
    V3 O, D;
    if (MULTI) {
      inst = MS.ray_inst[r];
      const WaveInst *wi = MS.W.insts + inst;
      T.nodes4 = wi->nodes4; T.tris = wi->tris;
      O = xfm_point(wi->minv, mk3(a.x, a.y, a.z)); D = xfm_vector(wi->minv, mk3(b.x, b.y, b.z));
    } else {
      O = xfm_point(minv, mk3(a.x, a.y, a.z)); D = xfm_vector(minv, mk3(b.x, b.y, b.z));
    }
    const bool occluded = wave_any_run<LONG_CAP, LONG_PHYS>(T.nodes4, T.tris, O, D, slab_of(O, D), tnear, s_ref, l_ref, counter + TRAV_OVF_WORD);
    if (!occluded && lane == 0) { // un-occluded: moved on, or ended here by shuffleRays' terminal rule (TracerBase.h:396-400)
      const float4 c = q.p2[r], d = q.p3[r];
      bool go_on = true;
      if (sink.fb) {
        float ret_t;
        go_on = top_nearest(a, b, sink.top, inst, ret_t) >= 0;
        if (!go_on) {
          const V3 col = mk3(c.x, c.y, c.z);
          const unsigned id = (unsigned)__float_as_int(d.x);
          if (__float_as_int(d.w) == 1 && len3(col) > 0.f && id < sink.n_pix) {
            const V3 cw = scl3(col, d.z);
            float *px = sink.fb + (size_t)4 * id;
            atomicAdd(px + 0, cw.x); atomicAdd(px + 1, cw.y); atomicAdd(px + 2, cw.z); atomicAdd(px + 3, 1.f);
          }
        }
      }
      if (go_on) {
        const unsigned slot = atomicAdd(out_count, 1u);
        out.p0[slot] = a; out.p1[slot] = b; out.p2[slot] = c; out.p3[slot] = d;
        if (out.p4) out.p4[slot] = 0u;
        store_no_known(out, slot);
        if (MULTI && MS.out_from) MS.out_from[slot] = inst;
      }
    }
    __builtin_amdgcn_wave_barrier(); // (convergent: the lanes meet again here, not at the loop header's readfirstlane -- see k_long_closest)
  }
}

#ifdef GVT_EXPERIMENTS
#include "experiments/packet_kernel.inc" // k_packet: a wave walks the BVH for a packet of 64 coherent rays
#endif

// ------------------------------------------------------------------------------------------------
// Shading (Material.cpp:50-139, Light.cpp:58-133), in the oracle's evaluation order
// ------------------------------------------------------------------------------------------------
__device__ inline V3 light_contribution(const gvt_hip_light &L, V3 hit, V3 samplePos) {
  V3 c = ld3(L.color);
  if (L.type == GVT_HIP_LIGHT_AMBIENT) return c; // Light.cpp:70
  V3 p = (L.type == GVT_HIP_LIGHT_AREA) ? samplePos : ld3(L.position);
  float distance = 1.f / len3(sub3(p, hit));
  distance = (distance > 1.f) ? 1.f : distance;
  return scl3(c, distance);
}

__device__ inline V3 area_light_position(const gvt_hip_light &L, uint32_t &seed) { // Light.cpp:72-99,115-127
  V3 v = ld3(L.normal), u, w;
  if (v.x == 0.f && v.y == 1.f && v.z == 0.f) {
    u = mk3(1, 0, 0); w = mk3(0, 0, 1);
  } else {
    const V3 up = mk3(0, 1, 0);
    u.x = up.y * v.z - v.y * up.z; u.y = up.z * v.x - v.z * up.x; u.z = up.x * v.y - v.x * up.y;
    w.x = v.y * u.z - u.y * v.z; w.y = v.z * u.x - u.z * v.x; w.z = v.x * u.y - u.x * v.y;
  }
  float xLocation = (float)(((double)gvt_fastrand_lcg(seed, 0, 1) - 0.5) * (double)L.width);
  float zLocation = (float)(((double)gvt_fastrand_lcg(seed, 0, 1) - 0.5) * (double)L.height);
  float xCoord = xLocation * u.x + zLocation * w.x;
  float yCoord = xLocation * u.y + zLocation * w.y;
  float zCoord = xLocation * u.z + zLocation * w.z;
  return mk3(L.position[0] + xCoord, L.position[1] + yCoord, L.position[2] + zCoord);
}

// primitives::Shade (Material.cpp:90-139)
// the material as Shade() sees it (per-face material, Mesh::mat, or the vertex-colour Lambert of EmbreeMeshAdapter.cpp:534-569)
struct MatEval {
  int type;
  V3 kd, ks;
  float alpha;
  V3 eta, k;
  float roughness;
  V3 hsc;
  float back, falloff;
};
__device__ inline MatEval mat_eval(const gvt_hip_material &m) {
  MatEval e;
  e.type = m.type; e.kd = ld3(m.kd); e.ks = ld3(m.ks); e.alpha = m.alpha; e.eta = ld3(m.eta); e.k = ld3(m.k); e.roughness = m.roughness;
  e.hsc = ld3(m.horizonScatteringColor); e.back = m.backScattering; e.falloff = m.horizonScatteringFallOff;
  return e;
}
__device__ inline float clamp01(float x) { const float a = (x < 0.f) ? 0.f : x; return (1.f < a) ? 1.f : a; } // embree clamp(x)

__device__ inline bool shade(const MatEval &m, const RayRec &ray, V3 N, const gvt_hip_light &L, V3 lightPos, V3 &out) {
  const int mtype = m.type;
  const V3 kd = m.kd, ks = m.ks;
  const float alpha = m.alpha;
  V3 hitPoint = add3(ray.o, scl3(ray.d, ray.t));
  V3 wi = norm3(sub3(lightPos, hitPoint));
  float dNw = dot3(N, wi);
  float NdotL = (0.f < dNw) ? dNw : 0.f;
  V3 Li = light_contribution(L, hitPoint, lightPos);
  if (NdotL == 0.f || (Li.x == 0.f && Li.y == 0.f && Li.z == 0.f)) return false;
  V3 color;
  if (mtype == 0) { // lambertShade :50-57
    color = scl3(kd, NdotL * ray.w);
  } else if (mtype == 1) { // phongShade :59-70
    V3 R = sub3(scl3(scl3(N, 2.f), NdotL), wi);
    float vr = dot3(R, neg3(ray.d));
    float VdotR = (0.f < vr) ? vr : 0.f;
    float power = VdotR * powf(VdotR, alpha);
    color = scl3(kd, NdotL * ray.w);
    color = add3(color, scl3(ks, power * ray.w));
  } else if (mtype == 2) { // blinnPhongShade :72-87
    V3 H = norm3(sub3(wi, ray.d));
    float hn = dot3(H, N);
    float NdotH = (0.f < hn) ? hn : 0.f;
    float power = NdotH * powf(NdotH, alpha);
    V3 diffuse = scl3(kd, NdotL * ray.w);
    V3 specular = scl3(ks, power * ray.w);
    color = add3(diffuse, specular);
  } else if (mtype >= 3 && mtype <= 5) {
    // EMBREE_MATERIAL_METAL / VELVET / MATTE: Material.cpp:106-122 -> Material__eval (adapter/embree/EmbreeMaterial.h:289-314), the
    // Embree tutorials' BRDFs; dg.Ns = N, wo = -ray.direction.  Embree's rcp/rsqrt (SSE estimate + Newton step) are 1/x, 1/sqrt(x)
    // here: parity with the reference is ~1e-6 relative for these types (checked against its own build through the oracle).
    const V3 wo = neg3(ray.d);
    const float one_over_pi = 0.31830988618379069122f;
    V3 r = mk3(0, 0, 0);
    if (mtype == 5) { // MatteMaterial__eval :168-172
      r = scl3(kd, clamp01(dot3(wi, N)));
    } else if (mtype == 4) { // VelvetMaterial__eval :246-253 = Minneart :188-192 + Velvety :222-228
      const float cosThetaI = clamp01(dot3(wi, N));
      const float backScatter = powf(clamp01(dot3(wo, wi)), m.back);
      const V3 a = scl3(ks, backScatter * cosThetaI * one_over_pi);
      const float cosThetaO = clamp01(dot3(wo, N));
      const float sinThetaO = sqrtf(1.0f - cosThetaO * cosThetaO);
      const float horizonScatter = powf(sinThetaO, m.falloff);
      const V3 b = scl3(m.hsc, horizonScatter * cosThetaI * one_over_pi);
      r = add3(a, b);
    } else { // MetalMaterial__eval :259-277; optics.h:75-83 fresnelConductor, :131-137 PowerCosineDistribution
      const float expo = 1.0f / m.roughness;
      const float cosThetaO = dot3(wo, N), cosThetaI = dot3(wi, N);
      if (!(cosThetaI <= 0.0f || cosThetaO <= 0.0f)) {
        const V3 s_ = add3(wi, wo);
        const V3 wh = scl3(s_, 1.0f / sqrtf(dot3(s_, s_)));
        const float cosThetaH = dot3(wh, N);
        const float cosTheta = dot3(wi, wh);
        const float cosi = cosTheta, c2 = cosi * cosi;
        const V3 tmp = add3(mul3(m.eta, m.eta), mul3(m.k, m.k));
        const V3 two_eta_c = scl3(scl3(m.eta, 2.0f), cosi);
        const V3 one = mk3(1.0f, 1.0f, 1.0f), vc2 = mk3(c2, c2, c2);
        const V3 num1 = add3(sub3(scl3(tmp, c2), two_eta_c), one), den1 = add3(add3(scl3(tmp, c2), two_eta_c), one);
        const V3 num2 = add3(sub3(tmp, two_eta_c), vc2), den2 = add3(add3(tmp, two_eta_c), vc2);
        const V3 Rpar = mk3(num1.x / den1.x, num1.y / den1.y, num1.z / den1.z);
        const V3 Rper = mk3(num2.x / den2.x, num2.y / den2.y, num2.z / den2.z);
        const V3 F = scl3(add3(Rpar, Rper), 0.5f);
        const float D = (expo + 2) * (1.0f / (2.0f * 3.14159265358979323846f)) * powf(fabsf(cosThetaH), expo);
        const float g1 = 2.0f * cosThetaH * cosThetaO / cosTheta, g2 = 2.0f * cosThetaH * cosThetaI / cosTheta;
        const float gm = (g1 < g2) ? g1 : g2;
        const float G = (1.0f < gm) ? 1.0f : gm;
        r = scl3(scl3(scl3(mul3(ks, F), D), G), 1.0f / (4.0f * cosThetaO));
      }
    }
    color = scl3(scl3(r, 2.f), ray.w); // 2.f * glm::vec3(r) * ray.w :120
  } else {
    color = mk3(0, 0, 0);
  }
  color = mul3(color, Li);
  float c[3] = { color.x, color.y, color.z };
  for (int i = 0; i < 3; i++) {
    float a = (c[i] < 0.f) ? 0.f : c[i];
    c[i] = (1.f < a) ? 1.f : a;
  }
  out = mk3(c[0], c[1], c[2]);
  return true;
}

#ifdef GVT_EXPERIMENTS
// out-of-line copies for k_fused: the shading code runs once per ray, the traversal loop thousands of times -- keeping it a call
// keeps its registers out of the loop's allocation
__device__ __attribute__((noinline)) bool shade_call(const MatEval &m, const RayRec &ray, V3 N, const gvt_hip_light &L, V3 lightPos, V3 &out) {
  return shade(m, ray, N, L, lightPos, out);
}
#endif

// CosWeightedRandomHemisphereDirection2 (EmbreeMeshAdapter.cpp:289-318)
__device__ inline V3 cos_weighted_dir(V3 n, uint32_t &seed) {
  float Xi1 = gvt_fastrand01(seed);
  float Xi2 = gvt_fastrand01(seed);
  // acos / sinf / cosf: include/gvt_math.h, the definitions shared with the checker (no ocml call: bit-identical on both sides)
  float theta = (float)gvt_acos(__builtin_sqrt(1.0 - (double)Xi1));
  float phi = (float)(2.0 * 3.1415926535897932384626433832795 * (double)Xi2);
  float xs = gvt_sinf(theta) * gvt_cosf(phi);
  float ys = gvt_cosf(theta);
  float zs = gvt_sinf(theta) * gvt_sinf(phi);
  V3 y = n, h = y;
  if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z)) h.x = 1.0f;
  else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z)) h.y = 1.0f;
  else h.z = 1.0f;
  V3 x = cross3(h, y);
  V3 z = cross3(x, y);
  V3 d = add3(add3(scl3(x, xs), scl3(y, ys)), scl3(z, zs));
  return norm3(d);
}

struct ShadeArgs {
  RayPlanes in;            // rayList (updated in place)
  const unsigned *idx;     // active list or null (identity)
  unsigned n;
  unsigned long long index_base;
  const gvt_hip_hit *hits;
  int first_pass;
  int carried_rng;         // 1: a ray's stream is the word it carries (plane 4); 0: first pass keyed on (seed, index in rayList)
  RayPlanes out; unsigned *out_count;       // moved_rays
  RayPlanes shadow; unsigned *shadow_count; // shadowRays of this pass
  unsigned *next_idx; unsigned *next_count; // rays that bounce (valid[pi] stays set)
  const gvt_hip_light *lights;
  Mat3 normi;
  int normal_mode, n_lights;
  uint32_t seed;
  unsigned *zero_word;     // reset for the launch that follows (the any-hit kernel's work counter)
  TermSink sink;           // terminal rule of the shuffle for rays that leave this instance without a hit (fb == nullptr: off)
  int update_in_place;     // 1: every shaded ray is written back (gvt_hip_trace: the caller reads rayList); 0: only rays that bounce
  const unsigned *n_dev;   // ray count of this pass in device memory (a pass launched without a host round trip); null: n
  // merged launch (k_shade<true>): rays are virtual indices into W's segments; per-ray mesh / normi from W.insts
  WaveSet W;
  int *out_from;           // source instance of every ray appended to `out`
  int *shadow_inst;        // instance of every ray appended to `shadow`
  unsigned shadow_stride;  // != 0: the shadow list is direct-mapped -- light li's ray of thread j at slot li * stride + j, in the order of the
                           // traced list (its tile coherence kept for the any-hit launch, no slot atomics); shadow_inst < 0 marks empty slots
};

#define SHADE_BLOCK 512
template <bool MULTI>
__global__ __launch_bounds__(SHADE_BLOCK, 4) void k_shade(ShadeArgs A, MeshView M1) { // 128 VGPRs: 4 waves per SIMD (unbounded: 134-148, 3)
  // one LDS (count, base) pair per output list use: moved_rays, next list, one per light
  __shared__ unsigned sh_alloc[2 * (2 + 64)];
  for (int k = threadIdx.x; k < 2 * (2 + 64); k += SHADE_BLOCK) sh_alloc[k] = 0u;
  __syncthreads();
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0 && A.zero_word) *A.zero_word = 0u;
  const unsigned n = A.n_dev ? *A.n_dev : A.n;
  const bool in_range = j < n;
  unsigned i = in_range ? (A.idx ? A.idx[j] : j) : 0u; // index in rayList (MULTI: virtual index, then index in the ray's queue)
  const unsigned gi = i;
  RayPlanes in = A.in;
  const MeshView *M = &M1;
  Mat3 normi = A.normi;
  int inst = A.sink.from;
  if (MULTI && in_range) {
    const WaveSeg sg = A.W.segs[wave_find_seg(A.W, i)];
    in = make_planes(sg.planes, sg.cap);
    i -= sg.begin;
    inst = sg.inst;
    M = &A.W.insts[inst].mv;
    normi = A.W.insts[inst].normi;
  }
  RayRec r;
  gvt_hip_hit h;
  h.prim = -1; h.t = 0.f; h.u = 0.f; h.v = 0.f;
  bool miss = false, shaded = false, bounce = false;
  uint32_t g_seed = 0;
  V3 N = mk3(0, 0, 0);
  MatEval me = mat_eval(M->mat);
  if (in_range) {
    r = load_ray(in, i);
    h = A.hits[j];
    g_seed = (A.first_pass && (!A.carried_rng || r.rng == 0u)) ? ray_stream_seed(A.seed, A.index_base + gi) : r.rng;
    if (h.prim < 0) {
      miss = true; // :605-609
    } else if (r.type != 1) { // a SHADOW ray that hits is dropped :486-488
      shaded = true;
      float t = h.t;
      r.t = t; // :491
      // the triangle's vertices from the 64-byte slot the traversal has just intersected (one line, usually still in L2) instead of
      // three index words and three vertices gathered from the mesh's own arrays; the indices only where per-vertex attributes are used
      const float4 *sl = M->slots + 4 * (size_t)M->slot_of[h.prim];
      const float4 s0 = sl[0], s1 = sl[1], s2 = sl[2], s3 = sl[3];
      const V3 v0 = mk3(s0.x, s0.y, s0.z), v1 = mk3(s1.w, s2.w, s3.x), v2 = mk3(s3.y, s3.z, s3.w);
      int ia = 0, ib = 0, ic = 0;
      if (A.normal_mode == GVT_HIP_NORMALS_SMOOTH || M->vcolors) { ia = M->tris[3 * h.prim]; ib = M->tris[3 * h.prim + 1]; ic = M->tris[3 * h.prim + 2]; }
      const V3 negNg = cross3(sub3(v1, v0), sub3(v2, v0)); // -Ng of Embree (cf. OptixMeshAdapter.cu:280-287)
      const V3 normalflat = norm3(mat3_mul(normi, negNg)); // :504
      if (A.normal_mode == GVT_HIP_NORMALS_SMOOTH) { // :505-518
        const V3 a = ld3(M->normals + 3 * ib), b = ld3(M->normals + 3 * ic), c = ld3(M->normals + 3 * ia);
        const V3 mn = add3(add3(scl3(a, h.u), scl3(b, h.v)), scl3(c, 1.0f - h.u - h.v));
        N = norm3(mat3_mul(normi, mn));
      } else {
        N = normalflat; // :520-522
      }
      if (dot3(neg3(r.d), normalflat) <= 0.f) N = neg3(N); // :527-529
      // material pick :534-569
      if (M->vcolors) {
        const V3 c0 = ld3(M->vcolors + 3 * ia), c1 = ld3(M->vcolors + 3 * ib), c2 = ld3(M->vcolors + 3 * ic);
        me.kd = add3(add3(scl3(c0, 1.f - h.u - h.v), scl3(c1, h.u)), scl3(c2, h.v));
        me.type = 0; me.ks = mk3(.5f, .5f, .5f); me.alpha = 1.f;
      } else if (M->face_mat && M->face_mat[h.prim] >= 0 && (unsigned)M->face_mat[h.prim] < M->n_mat) {
        me = mat_eval(M->materials[M->face_mat[h.prim]]);
      }
      if (r.type == 2) { // SECONDARY :572-575
        t = (t > 1) ? 1.f / t : t;
        r.w = r.w * t;
      }
    }
  }
  // moved_rays: misses are forwarded as they are -- unless the sink is on and no other instance lies ahead: then shuffleRays would
  // only drop the ray (or, for a SHADOW ray that carries colour, deposit it: TracerBase.h:396-400), which is done here at once
  {
    bool forward = miss;
    if (miss && A.sink.fb) {
      float ret_t;
      const float4 a = make_float4(r.o.x, r.o.y, r.o.z, r.t_min), b = make_float4(r.d.x, r.d.y, r.d.z, r.t_max);
      if (top_nearest(a, b, A.sink.top, inst, ret_t) < 0) {
        forward = false;
        if (r.type == 1 && len3(r.c) > 0.f && (unsigned)r.id < A.sink.n_pix) {
          const V3 cw = scl3(r.c, r.w);
          float *px = A.sink.fb + (size_t)4 * (unsigned)r.id;
          atomicAdd(px + 0, cw.x); atomicAdd(px + 1, cw.y); atomicAdd(px + 2, cw.z); atomicAdd(px + 3, 1.f);
        }
      }
    }
    const unsigned slot = block_alloc(A.out_count, forward, &sh_alloc[0]);
    if (forward) { store_ray(A.out, slot, r); if (MULTI) A.out_from[slot] = inst; }
  }
  // generateShadowRays :320-358 -- one pass per light so that the wave allocates slots together
  for (int li = 0; li < A.n_lights; li++) {
    bool emit = false;
    RayRec s;
    if (shaded) {
      const gvt_hip_light L = A.lights[li];
      const V3 lightPos = (L.type == GVT_HIP_LIGHT_AREA) ? area_light_position(L, g_seed) : ld3(L.position);
      V3 c;
      if (shade(me, r, N, L, lightPos, c)) {
        emit = true;
        const float multiplier = 1.0f - GVT_RAY_EPSILON * 16;
        const float t_shadow = multiplier * r.t;
        const V3 origin = add3(r.o, scl3(r.d, t_shadow));
        const V3 dir = sub3(lightPos, origin);
        s.o = origin; s.t_min = GVT_RAY_EPSILON;
        s.d = norm3(dir); // Ray ctor normalizes, Ray.h:109
        s.t_max = 3.0f;   // dir.length() == glm component count (:347,355)
        s.c = c; s.t = r.t;
        s.id = r.id; s.depth = r.depth; s.w = r.w; s.type = 1;
        s.rng = 0u;
        s.km[0] = 0u; s.km[1] = 0u; s.km[2] = 0u;
      }
    }
    if (A.shadow_stride) {
      if (j < A.shadow_stride) {
        const unsigned slot = (unsigned)li * A.shadow_stride + j;
        A.shadow_inst[slot] = emit ? (MULTI ? inst : 0) : -1;
        if (emit) store_ray(A.shadow, slot, s);
      }
    } else {
      const unsigned slot = block_alloc(A.shadow_count, emit, &sh_alloc[2 * (2 + li)]);
      if (emit) { store_ray(A.shadow, slot, s); if (MULTI) A.shadow_inst[slot] = inst; }
    }
  }
  if (shaded) { // :584-602
    const int ndepth = r.depth - 1;
    const float p = 1.f - gvt_fastrand01(g_seed);
    if (ndepth > 0 && r.w > p) {
      r.type = 2;
      const float multiplier = 1.0f - 16.0f * GVT_FLT_EPSILON;
      const float t_secondary = multiplier * r.t;
      r.o = add3(r.o, scl3(r.d, t_secondary));
      const V3 nd = cos_weighted_dir(N, g_seed);
      r.d = nd;
      r.w = r.w * dot3(nd, N);
      r.depth = ndepth;
      r.km[0] = 0u; r.km[1] = 0u; r.km[2] = 0u; // a new straight segment: nothing is known about it
      bounce = true;
    }
    if (bounce || A.update_in_place) { // rayList is updated in place; a list the caller discards anyway (device queues) only needs it for the next pass
      r.rng = g_seed; // the stream goes on with the ray
      store_ray(in, i, r);
    }
  }
  {
    const unsigned slot = block_alloc(A.next_count, bounce, &sh_alloc[2]);
    if (bounce) A.next_idx[slot] = gi;
  }
}

#include "finish_kernel.inc" // k_finish: a small round in one launch, every ray followed to its end on this rank

#ifdef GVT_EXPERIMENTS
#include "experiments/fused_kernel.inc" // k_fused: the one-kernel closest + shade + shadow variant (measured slower; knob `fused`)
#endif

// ------------------------------------------------------------------------------------------------
// layout conversions at the ABI boundary: 80-byte Ray AoS <-> planes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_aos_to_planes(const float4 *__restrict__ src /* 5 float4 per ray */, unsigned n, RayPlanes dst,
                                                        unsigned long long off) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 *s = src + (size_t)5 * i;
  dst.p0[off + i] = s[0]; dst.p1[off + i] = s[1]; dst.p2[off + i] = s[2]; dst.p3[off + i] = s[3];
  if (dst.p4) dst.p4[off + i] = __float_as_uint(s[4].x); // bytes 64..67: the stream word
  if (dst.p5) { dst.p5[3 * (off + i)] = __float_as_uint(s[4].y); dst.p5[3 * (off + i) + 1] = __float_as_uint(s[4].z); dst.p5[3 * (off + i) + 2] = __float_as_uint(s[4].w); } // 68..79: known misses
}
__global__ __launch_bounds__(256) void k_planes_to_aos(RayPlanes src, unsigned long long off, unsigned n, float4 *__restrict__ dst) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 *d = dst + (size_t)5 * i;
  d[0] = src.p0[off + i]; d[1] = src.p1[off + i]; d[2] = src.p2[off + i]; d[3] = src.p3[off + i];
  d[4] = make_float4(__uint_as_float(src.p4 ? src.p4[off + i] : 0u), __uint_as_float(src.p5 ? src.p5[3 * (off + i)] : 0u), __uint_as_float(src.p5 ? src.p5[3 * (off + i) + 1] : 0u),
                     __uint_as_float(src.p5 ? src.p5[3 * (off + i) + 2] : 0u));
}
__global__ __launch_bounds__(256) void k_od_to_planes(const float *__restrict__ org, const float *__restrict__ dir, unsigned n, RayPlanes dst) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst.p0[i] = make_float4(org[3 * i], org[3 * i + 1], org[3 * i + 2], GVT_RAY_EPSILON);
  dst.p1[i] = make_float4(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], GVT_FLT_MAX);
}

// ------------------------------------------------------------------------------------------------
// Ray sorting: 30-bit Morton code of the object-space origin inside the mesh box + direction octant.
// Neighbouring lanes then walk neighbouring nodes (fewer divergent iterations, more L1/L2 hits).
// ------------------------------------------------------------------------------------------------
__device__ inline unsigned expand10(unsigned v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(256) void k_ray_keys(RayPlanes q, const unsigned *__restrict__ idx, unsigned n, Mat4 minv, float3 blo, float3 inv_ext,
                                                   unsigned *__restrict__ keys, unsigned *__restrict__ vals, int key_bits) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned i = idx ? idx[j] : j;
  const float4 a = q.p0[i], b = q.p1[i];
  const V3 O = xfm_point(minv, mk3(a.x, a.y, a.z)), D = xfm_vector(minv, mk3(b.x, b.y, b.z));
  // the point where the ray enters the mesh's box (the origin itself for rays that start inside): camera rays, which all
  // leave one eye point, are thereby keyed by where they hit the box face -- 2-D tiles of the image instead of scanlines
  const float bhx = blo.x + (inv_ext.x > 0.f ? 1.f / inv_ext.x : 0.f), bhy = blo.y + (inv_ext.y > 0.f ? 1.f / inv_ext.y : 0.f),
              bhz = blo.z + (inv_ext.z > 0.f ? 1.f / inv_ext.z : 0.f);
  const float rx = 1.f / (fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x), ry = 1.f / (fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y),
              rz = 1.f / (fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z);
  const float t_in = fmaxf(fmaxf(fminf((blo.x - O.x) * rx, (bhx - O.x) * rx), fminf((blo.y - O.y) * ry, (bhy - O.y) * ry)),
                           fmaxf(fminf((blo.z - O.z) * rz, (bhz - O.z) * rz), 0.f));
  const float te = t_in < 1e30f ? t_in : 0.f;
  const float px = (O.x + D.x * te - blo.x) * inv_ext.x, py = (O.y + D.y * te - blo.y) * inv_ext.y, pz = (O.z + D.z * te - blo.z) * inv_ext.z;
  const unsigned qx = (unsigned)fminf(fmaxf(px * 1024.f, 0.f), 1023.f);
  const unsigned qy = (unsigned)fminf(fmaxf(py * 1024.f, 0.f), 1023.f);
  const unsigned qz = (unsigned)fminf(fmaxf(pz * 1024.f, 0.f), 1023.f);
  const unsigned oct = (D.x < 0.f ? 4u : 0u) | (D.y < 0.f ? 2u : 0u) | (D.z < 0.f ? 1u : 0u);
  const unsigned morton = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
  keys[j] = ((oct << 29) | (morton >> 1)) >> (32 - key_bits); // top key_bits bits: octant, then the coarsest Morton levels
  vals[j] = i;
}

__global__ __launch_bounds__(256) void k_math_probe(int kind, const float *__restrict__ in, unsigned n, float *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  out[i] = kind == 0 ? gvt_sinf(x) : kind == 1 ? gvt_cosf(x) : (float)gvt_acos(__builtin_sqrt(1.0 - (double)x));
}

__global__ void k_set_u32(unsigned *p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }
// start of a trace call: moved_rays count := its current size, work counter / shadow count / next count := 0 (one launch, not three memsets)
__global__ void k_trace_begin(unsigned *out_count, unsigned out_size, unsigned *counters) {
  if (blockIdx.x == 0 && threadIdx.x < 5) counters[threadIdx.x] = 0u; // work counter, shadow count, next count, parked rays, their work counter
  if (blockIdx.x == 0 && threadIdx.x == 5) *out_count = out_size;
}

// after the sort: object-space origin/direction of the rays in sorted order, as two contiguous planes, so that the
// traversal kernel's lane refills read consecutive memory (the transform is the one k_trace<XFORM> would apply)
__global__ __launch_bounds__(256) void k_gather_od(RayPlanes q, const unsigned *__restrict__ idx, unsigned n, Mat4 minv, float4 *__restrict__ o_out,
                                                    float4 *__restrict__ d_out) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned i = idx[j];
  const float4 a = q.p0[i], b = q.p1[i];
  const V3 O = xfm_point(minv, mk3(a.x, a.y, a.z)), D = xfm_vector(minv, mk3(b.x, b.y, b.z));
  o_out[j] = make_float4(O.x, O.y, O.z, a.w);
  d_out[j] = make_float4(D.x, D.y, D.z, b.w);
}

inline unsigned blocks_for(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }

int trav_grid(size_t n) {
  Ctx &C = gctx();
  size_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
  return (int)(need < (size_t)C.trav_blocks ? (need ? need : 1) : (size_t)C.trav_blocks);
}

template <bool ANY, bool XFORM, int MODE, typename... Args> void launch_trace(bool have_nodes4, int grid, hipStream_t st, Args... args) {
#ifdef GVT_EXPERIMENTS
  if (!(gctx().wide4 && have_nodes4)) { // the uncompressed binary nodes
    if (gctx().coop_fetch) k_trace<ANY, XFORM, MODE, true, false><<<grid, TRAV_BLOCK, 0, st>>>(args...);
    else k_trace<ANY, XFORM, MODE, false, false><<<grid, TRAV_BLOCK, 0, st>>>(args...);
    return;
  }
#endif
  k_trace<ANY, XFORM, MODE, false, true><<<grid, TRAV_BLOCK, 0, st>>>(args...); // a mesh without 4-wide nodes has no triangles: every ray retires as a miss
}

// scratch of a closest-hit launch's parked rays: n records, then LONG_SAVE stack entries for each of the first LONG_STK_CAP of them
static size_t long_scratch_bytes(size_t n) { return sizeof(LongRec) * n + sizeof(int) * LONG_SAVE * (size_t)LONG_STK_CAP; }
static void long_limits(LongQ &LQ, size_t n) {
  Ctx &C = gctx();
  LQ.steps = C.long_steps;
  LQ.steps_drain = C.long_steps_drain > 0 && C.long_steps_drain < C.long_steps ? C.long_steps_drain : C.long_steps;
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

int convert_aos_to_planes(const gvt_hip_ray *d_src, size_t n, RayPlanes dst, size_t dst_off) {
  if (!n) return 0;
  ProfScope ps(KC_CONVERT);
  k_aos_to_planes<<<blocks_for(n), 256, 0, gctx().stream>>>((const float4 *)d_src, (unsigned)n, dst, dst_off);
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
    ShadeArgs A;
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
__global__ void k_wave_pass_begin(unsigned *c, int pass, unsigned n_host, unsigned *out_count, const unsigned *n_dev0 = nullptr) {
  if (blockIdx.x || threadIdx.x) return;
  unsigned long long *tot = (unsigned long long *)(c + 16);
  const int cur = (pass & 1) ? 5 : 2, prev = (pass & 1) ? 2 : 5;
  if (pass == 0) { *out_count = 0u; c[2] = 0u; c[5] = 0u; tot[0] += n_dev0 ? *n_dev0 : n_host; }
  else { tot[1] += c[1]; tot[0] += c[prev]; c[cur] = 0u; }
  c[0] = 0u; c[1] = 0u; c[3] = 0u; c[4] = 0u; c[6] = 0u;
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
                     bool defer_end) {
  Ctx &C = gctx();
  if (!n_total) return 0;
  hipStream_t st = C.stream;
  const int nL = P.n_lights;
  const size_t n = n_total;
  gvt_hip_hit *d_hits = (gvt_hip_hit *)scratch_get(0, sizeof(gvt_hip_hit) * n);
  const size_t shadow_cap = n * (size_t)(nL > 0 ? nL : 1);
  float4 *d_shadow = (float4 *)scratch_get(2, sizeof(float4) * 4 * shadow_cap);
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
  RayPlanes shadow = make_planes(d_shadow, shadow_cap);
  shadow.p4 = nullptr; shadow.p5 = nullptr;
  RayPlanes outp = make_planes(out->d_planes, out->cap);
  RayPlanes none{};
  unsigned *c = C.d_counters;
  Trav T{};
  Mat4 id{};
  const bool use_long = C.long_steps > 0 && n >= (size_t)C.long_min_rays;
  const bool small = n <= (size_t)C.small_rays; // a wave per ray (see k_long_seed)
  const int small_grid = (int)std::min<size_t>((n + 3) / 4, (size_t)C.n_cu * 3);
  for (int pass = 0; pass < passes; pass++) {
    const unsigned *n_dev = pass ? c + ((pass & 1) ? 2 : 5) : ((single && single->n_dev) ? single->n_dev : nullptr); // count written by the previous pass's k_shade
    const unsigned *idx = pass ? ((pass & 1) ? d_idx_a : d_idx_b) : nullptr;    // its bounce list
    unsigned *next = (pass & 1) ? d_idx_b : d_idx_a;
    unsigned *c_next = c + ((pass & 1) ? 5 : 2);
    if (!(pass == 0 && single && single->pass0_begun))
      k_wave_pass_begin<<<1, 64, 0, st>>>(c, pass, (unsigned)n, out->d_count, (single && single->n_dev) ? single->n_dev : nullptr);
    if (single) {
      // one segment (a single non-empty local queue, e.g. the one-domain benchmark): the single-mesh kernels -- no segment lookup and
      // no per-ray table loads at a refill -- with the same device-side counts
      gvt_hip_mesh *M = single->mesh;
      const bool have4 = M->d_nodes4 != nullptr;
      Trav TS{ M->d_nodes, M->d_tri, M->d_nodes4 };
      LongQ LQ{};
      if (use_long && have4) { LQ.recs = d_long; LQ.count = c + 3; long_limits(LQ, n); }
      const bool small1 = small && have4;
#ifdef GVT_EXPERIMENTS
      // a coherent list (camera rays in 8x8 tiles, straight from the filter): a wave walks the tree for 64 rays at once (k_packet)
      const bool pkt = C.packet && single->coherent && pass == 0 && have4 && !small1;
      if (pkt) {
        ProfScope ps(KC_CLOSEST);
        LongQ LP{ d_long, c + 3, nullptr, 0u, 0, 0 };
        k_packet<false><<<blocks_for(n), 256, 0, st>>>(single->planes, (unsigned)n, n_dev, single->minv, TS, GVT_RAY_EPSILON, d_hits, nullptr, none, nullptr,
                                                     TermSink{}, LP, nullptr, nullptr, nullptr, c + 9);
        k_long_closest<true><<<C.n_cu * 3, 256, 0, st>>>(single->planes, d_long, c + 3, single->minv, TS, GVT_RAY_EPSILON, d_hits, c + 4); // packets that bailed out
      } else
#else
      const bool pkt = false;
#endif
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
      ShadeArgs A;
      A.in = single->planes; A.idx = idx; A.n = (unsigned)n; A.index_base = 0; A.hits = d_hits;
      A.first_pass = (pass == 0); A.carried_rng = 1; A.out = outp; A.out_count = out->d_count; A.shadow = shadow; A.shadow_count = c + 1;
      A.next_idx = next; A.next_count = c_next; A.lights = d_lights; A.normi = single->normi; A.normal_mode = P.normal_mode;
      A.n_lights = nL; A.seed = P.seed; A.zero_word = c + 0;
      A.sink = P.sink; A.sink.from = single->inst; A.update_in_place = 0;
      A.n_dev = n_dev; A.W = WaveSet{}; A.out_from = nullptr; A.shadow_inst = nullptr; A.shadow_stride = 0;
      if (pkt) { A.shadow_inst = d_shadow_inst; A.shadow_stride = (unsigned)n; } // shadow rays in the primaries' order: packets again
      MeshView mv;
      mv.slots = M->d_tri; mv.slot_of = M->d_slot_of; mv.verts = M->d_verts; mv.tris = M->d_tris; mv.normals = M->d_normals; mv.vcolors = M->d_vcolors;
      mv.materials = M->d_materials; mv.n_mat = (unsigned)M->nMat; mv.face_mat = M->d_face_mat; mv.mat = M->mesh_mat;
      {
        ProfScope ps(KC_SHADE);
        k_shade<false><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, mv);
      }
      if (nL) {
        ProfScope ps(KC_ANY);
        TermSink sk = P.sink;
        sk.from = single->inst;
#ifdef GVT_EXPERIMENTS
        if (pkt) {
          unsigned *d_retry = (unsigned *)scratch_get(18, sizeof(unsigned) * shadow_cap);
          if (!d_retry) return GVT_HIP_ERR_DEVICE;
          k_packet<true><<<blocks_for(shadow_cap), 256, 0, st>>>(shadow, (unsigned)shadow_cap, nullptr, single->minv, TS, GVT_RAY_EPSILON, nullptr, d_shadow_inst, outp, out->d_count,
                                                               sk, LongQ{}, d_retry, c + 6, (unsigned long long *)(c + 18), c + 9);
          // rays of packets that bailed out: one lane per ray (an empty list costs a few microseconds)
          launch_trace<true, true, 1>(have4, trav_grid2(4096), st, shadow, d_retry, 0u, single->minv, TS, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                      c + 0, C.d_spill, C.refill_min, C.inner_min, c + 6, C.share, (unsigned)C.share_min_rays, sk, LongQ{});
        } else if (quad_usable(M) && !small1)
          k_traceq<true, true, 1><<<quad_grid(shadow_cap), 256, 0, st>>>(shadow, nullptr, 0u, single->minv, TravQ{ M->d_nodes4q, M->d_triq }, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                       c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, c + 1, sk, LongQ{});
        else
#endif
        if (small1) k_wave_any<false><<<(int)std::min<size_t>((shadow_cap + 3) / 4, (size_t)C.n_cu * 3), 256, 0, st>>>(shadow, c + 1, single->minv, TS, GVT_RAY_EPSILON, outp, out->d_count, c + 0, sk, MultiSrc{});
        else launch_trace<true, true, 1>(have4, trav_grid2(shadow_cap), st, shadow, nullptr, 0u, single->minv, TS, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                         c + 0, C.d_spill, C.refill_min, C.inner_min, c + 1, C.share, (unsigned)C.share_min_rays, sk, LongQ{});
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
    if (small) {
      ProfScope ps(KC_CLOSEST);
      k_long_seed<<<blocks_for(n), 256, 0, st>>>(d_long, c + 3, idx, (unsigned)n, n_dev);
      k_long_closest<true, true><<<small_grid, 256, 0, st>>>(none, d_long, c + 3, id, T, GVT_RAY_EPSILON, d_hits, c + 4, W);
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
                                                                                      TermSink{}, LQ, MS);
    }
    if (use_long && !small) {
      ProfScope ps(KC_LONG);
      k_long_closest<true, true><<<C.n_cu * 3, 256, 0, st>>>(none, d_long, c + 3, id, T, GVT_RAY_EPSILON, d_hits, c + 4, W, LQ.stk);
    }
    ShadeArgs A;
    A.in = none; A.idx = idx; A.n = (unsigned)n; A.index_base = 0; A.hits = d_hits;
    A.first_pass = (pass == 0); A.carried_rng = 1; A.out = outp; A.out_count = out->d_count; A.shadow = shadow; A.shadow_count = c + 1;
    A.next_idx = next; A.next_count = c_next; A.lights = d_lights; A.normi = P.normi; A.normal_mode = P.normal_mode;
    A.n_lights = nL; A.seed = P.seed; A.zero_word = c + 0;
    A.sink = P.sink; A.update_in_place = 0;
    A.n_dev = n_dev; A.W = W; A.out_from = d_out_from; A.shadow_inst = d_shadow_inst;
    const bool direct = C.shadow_direct && pass == 0 && !small; // later passes and small rounds hold few rays: compacted slots
    A.shadow_stride = direct ? (unsigned)n : 0u;
    {
      ProfScope ps(KC_SHADE);
      k_shade<true><<<blocks_for(n, SHADE_BLOCK), SHADE_BLOCK, 0, st>>>(A, MeshView{});
    }
    if (nL) {
      ProfScope ps(KC_ANY);
      MultiSrc MA{ W, d_shadow_inst, d_out_from, direct ? (unsigned long long *)(c + 18) : nullptr };
      if (small) k_wave_any<true><<<(int)std::min<size_t>((shadow_cap + 3) / 4, (size_t)C.n_cu * 3), 256, 0, st>>>(shadow, c + 1, id, T, GVT_RAY_EPSILON, outp, out->d_count, c + 0, P.sink, MA);
#ifdef GVT_EXPERIMENTS
      else if (C.quad && W.quad_ok)
        k_traceq<true, true, 1, true><<<quad_grid(shadow_cap), 256, 0, st>>>(shadow, nullptr, direct ? (unsigned)shadow_cap : 0u, id, TravQ{}, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                           c + 0, C.d_spill, C.quad_refill_min, C.quad_inner_min, direct ? nullptr : c + 1, P.sink, LongQ{}, MA);
#endif
      else
      k_trace<true, true, 1, false, true, true><<<trav_grid2(shadow_cap), TRAV_BLOCK, 0, st>>>(shadow, nullptr, direct ? (unsigned)shadow_cap : 0u, id, T, GVT_RAY_EPSILON, nullptr, nullptr, outp, out->d_count,
                                                                                              c + 0, C.d_spill, C.refill_min, C.inner_min, direct ? nullptr : c + 1, C.share, (unsigned)C.share_min_rays,
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
int finish_round(const WaveSet &W, size_t n_total, const TraceParams &P, const gvt_hip_light *lights_host, const void *d_qdesc, const int *d_owner, int rank,
                 unsigned *d_queue_overflow, unsigned *const *d_count_ptr, const unsigned char *d_mask) {
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
      cached.assign((const unsigned char *)lights_host, (const unsigned char *)lights_host + bytes);
      cached_dst = d_lights;
      HIPCHK(hipMemcpyAsync(d_lights, cached.data(), bytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  FinishArgs A;
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
    k_finish<<<(int)std::min<size_t>((n_total + 3) / 4, (size_t)C.n_cu * 3), 256, 0, st>>>(A); // 48 KiB of LDS per block
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
