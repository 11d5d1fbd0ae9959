// gvt_device.h -- shared device-side types and arithmetic of the gfx950 adapter.
//
// Parity rule: every function here that feeds a value the reference would produce (ray transform,
// triangle test, normals, Shade, light contribution, camera) is written in the evaluation order of
// the reference source it cites (glm 0.9.8.1 order for vector ops) and this library is compiled with
// -ffp-contract=off, so the results are bit-identical to a strict-IEEE CPU evaluation.  Only the BVH
// slab test (our own structure, results do not depend on it as long as it is conservative) uses
// explicit fused multiply-adds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gvt_hip.h"
#define GVT_MATH_FN __host__ __device__ static inline
#include "../../include/gvt_math.h"

#define GVT_RAY_EPSILON 1.e-6f   // actor/Ray.cpp:33
#define GVT_FLT_MAX 3.402823466e+38f
#define GVT_FLT_EPSILON 1.192092896e-07f

struct V3 {
  float x, y, z;
};
__host__ __device__ inline V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ inline V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
__host__ __device__ inline V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline V3 mul3(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ inline V3 scl3(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ inline V3 neg3(V3 a) { return mk3(-a.x, -a.y, -a.z); }
// glm compute_dot<tvec3>: tmp = x*y; tmp.x + tmp.y + tmp.z  (func_geometric.inl:54-61)
__host__ __device__ inline float dot3(V3 a, V3 b) { V3 t = mul3(a, b); return t.x + t.y + t.z; }
// glm compute_cross (func_geometric.inl:74-85)
__host__ __device__ inline V3 cross3(V3 x, V3 y) {
  return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
__host__ __device__ inline float len3(V3 a) { return sqrtf(dot3(a, a)); }
// glm compute_normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
__host__ __device__ inline V3 norm3(V3 a) { return scl3(a, 1.0f / sqrtf(dot3(a, a))); }

// glm mat4 * vec4 (column-major m[c*4+r]): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)  (type_mat4x4.inl)
struct Mat4 {
  float m[16];
};
struct Mat3 {
  float n[9];
};
__host__ __device__ inline V3 xfm_point(const Mat4 &M, V3 p) {
  const float *m = M.m;
  return mk3((m[0] * p.x + m[4] * p.y) + (m[8] * p.z + m[12] * 1.0f), (m[1] * p.x + m[5] * p.y) + (m[9] * p.z + m[13] * 1.0f),
             (m[2] * p.x + m[6] * p.y) + (m[10] * p.z + m[14] * 1.0f));
}
__host__ __device__ inline V3 xfm_vector(const Mat4 &M, V3 d) {
  const float *m = M.m;
  return mk3((m[0] * d.x + m[4] * d.y) + (m[8] * d.z + m[12] * 0.0f), (m[1] * d.x + m[5] * d.y) + (m[9] * d.z + m[13] * 0.0f),
             (m[2] * d.x + m[6] * d.y) + (m[10] * d.z + m[14] * 0.0f));
}
// glm mat3 * vec3 (n[c*3+r]): m00*x + m10*y + m20*z left to right (type_mat3x3.inl)
__host__ __device__ inline V3 mat3_mul(const Mat3 &N, V3 v) {
  const float *n = N.n;
  return mk3(n[0] * v.x + n[3] * v.y + n[6] * v.z, n[1] * v.x + n[4] * v.y + n[7] * v.z, n[2] * v.x + n[5] * v.y + n[8] * v.z);
}

// ---------------------------------------------------------------------------------------------
// Device ray queue: four planes of float4 ("SoA of 16-byte columns"): every lane of a wave reads one
// float4 per plane, 1 KiB contiguous per wave-instruction.
//   plane0 = (origin.xyz, t_min)   plane1 = (direction.xyz, t_max)
//   plane2 = (color.rgb, t)        plane3 = (id, depth, w, type) bit-cast
//   plane4 = one uint32 per ray: the ray's RNG stream word = bytes 64..67 of the 80-byte Ray image (inside Ray::data[68], copied by
//            every Ray copy / pack of the reference and never read by it, actor/Ray.h:95,128-151).  The reference seeds one RandEngine
//            per TBB chunk (EmbreeMeshAdapter.cpp:446-447: schedule dependent); here the stream belongs to the ray and travels with
//            it -- through queues, shuffles and the wire -- so a frame does not depend on list order, rank count or scheduling.
//            p4 may be null (lists that never shade: object-space query rays, shadow-ray scratch): loads read 0, stores skip it.
//   plane5 = three uint32 per ray: the ray's KNOWN MISSES = bytes 68..79 of the Ray image (same unused part of Ray::data): six 16-bit
//            entries, instance + 1, of the instances the ray has crossed without a hit on its current straight segment (the schedulers'
//            known-miss shortcut, below).  p5 may be null (one-instance scenes, scratch lists): loads read 0, stores skip it.
// The traversal kernels touch planes 0 and 1 only (32 B/ray).
// ---------------------------------------------------------------------------------------------
struct RayPlanes {
  float4 *p0, *p1, *p2, *p3;
  uint32_t *p4;
  uint32_t *p5;
};
#define GVT_QUEUE_BYTES_PER_RAY (4 * sizeof(float4) + 4 * sizeof(uint32_t)) // = the 80 bytes of the Ray image
__host__ __device__ inline RayPlanes make_planes(float4 *base, size_t cap) { // a queue allocation: GVT_QUEUE_BYTES_PER_RAY * cap
  RayPlanes r;
  r.p0 = base; r.p1 = base + cap; r.p2 = base + 2 * cap; r.p3 = base + 3 * cap; r.p4 = (uint32_t *)(base + 4 * cap); r.p5 = r.p4 + cap;
  return r;
}

struct RayRec { // unpacked ray in registers
  V3 o; float t_min;
  V3 d; float t_max;
  V3 c; float t;
  int id, depth; float w; int type;
  uint32_t rng; // stream word (plane 4)
  uint32_t km[3]; // known misses (plane 5)
};
__device__ inline RayRec load_ray(const RayPlanes &q, size_t i) {
  float4 a = q.p0[i], b = q.p1[i], c = q.p2[i], d = q.p3[i];
  RayRec r;
  r.o = mk3(a.x, a.y, a.z); r.t_min = a.w;
  r.d = mk3(b.x, b.y, b.z); r.t_max = b.w;
  r.c = mk3(c.x, c.y, c.z); r.t = c.w;
  r.id = __float_as_int(d.x); r.depth = __float_as_int(d.y); r.w = d.z; r.type = __float_as_int(d.w);
  r.rng = q.p4 ? q.p4[i] : 0u;
  if (q.p5) { r.km[0] = q.p5[3 * i]; r.km[1] = q.p5[3 * i + 1]; r.km[2] = q.p5[3 * i + 2]; }
  else { r.km[0] = 0u; r.km[1] = 0u; r.km[2] = 0u; }
  return r;
}
__device__ inline void store_ray(const RayPlanes &q, size_t i, const RayRec &r) {
  q.p0[i] = make_float4(r.o.x, r.o.y, r.o.z, r.t_min);
  q.p1[i] = make_float4(r.d.x, r.d.y, r.d.z, r.t_max);
  q.p2[i] = make_float4(r.c.x, r.c.y, r.c.z, r.t);
  q.p3[i] = make_float4(__int_as_float(r.id), __int_as_float(r.depth), r.w, __int_as_float(r.type));
  if (q.p4) q.p4[i] = r.rng;
  if (q.p5) { q.p5[3 * i] = r.km[0]; q.p5[3 * i + 1] = r.km[1]; q.p5[3 * i + 2] = r.km[2]; }
}
__device__ inline void store_no_known(const RayPlanes &q, size_t i) { if (q.p5) { q.p5[3 * i] = 0u; q.p5[3 * i + 1] = 0u; q.p5[3 * i + 2] = 0u; } }

// ---------------------------------------------------------------------------------------------
// BVH node: 64 B, both children's boxes in the parent (one fetch decides both).
//   n0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)   n1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   n2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)   n3 = (child0, child1, -, -) bit-cast ints
// child >= 0: inner node index.  child < 0: leaf, ~child = (first_tri_slot << 3) | count (count <= 4).
// Triangle slot (64 B = one line, leaf order): t0 = (v0.xyz, primID), t1 = (e1 = v0-v1, 0), t2 = (e2 = v2-v0, 0),
// t3 = (Ng = e1 x e2, 0) -- all precomputed with the very float operations the test would perform.
// ---------------------------------------------------------------------------------------------
struct BvhNode {
  float4 n0, n1, n2, n3;
};
#ifndef GVT_LEAF_MAX
#define GVT_LEAF_MAX 2 // measured 1..6 on the 10 M soup: 1-3 within noise (0.90-0.92 ms closest), 4: 0.96, 6: 1.05
#endif
// Compressed 4-wide node (64 B = one line, 4 float4), built by collapsing the binary tree (largest inner child first) until
// four slots are filled.  Child boxes are quantised to 8 bits per plane on a per-node grid:
//   plane = origin[a] + q * 2^(exp[a]-127),  q in 0..255, rounded outwards and verified in double at build time, so every
//   decoded box CONTAINS the (already padded) binary box -- results cannot depend on the layout.
//   w0 = (origin.x, origin.y, origin.z, step.x)      step = 2^(exp-127) as a float (no decode in the traversal loop)
//   w1 = (qlo.x[4], qhi.x[4], qlo.y[4], qhi.y[4])    one byte per child, child c in byte c
//   w2 = (qlo.z[4], qhi.z[4], ref0, ref1)   w3 = (ref2, ref3, step.y, step.z)
// refs as in the binary node (>= 0: 4-wide node index, < 0: leaf); an unused slot has ref = GVT_EMPTY_REF (an empty leaf) and an
// inverted box (qlo 255, qhi 0), so the slab test itself rejects it.
#define GVT_NODE4_F4 4
#define GVT_EMPTY_REF (-1) // == leaf_ref(0, 0)
__host__ __device__ inline int leaf_ref(uint32_t first, uint32_t count) { return ~(int)((first << 3) | count); }

// Embree 2.x Moeller-Trumbore (kernels/geometry/triangle_intersector_moeller.h), restated from its published
// form (e1 = v0-v1, e2 = v2-v0, Ng = e1 x e2, inclusive edge tests); reached through rtcIntersect/rtcOccluded at
// EmbreeMeshAdapter.cpp:474,375.  True division instead of rcp+Newton.
__device__ inline bool tri_test(V3 O, V3 D, V3 v0, V3 e1, V3 e2, float tnear, float &t, float &u, float &v) {
  V3 Ng = cross3(e1, e2);
  V3 C = sub3(v0, O);
  V3 R = cross3(D, C);
  float den = dot3(Ng, D);
  float absDen = fabsf(den);
  float sgn = (den < 0.f) ? -1.f : 1.f;
  float U = dot3(R, e2) * sgn;
  float V = dot3(R, e1) * sgn;
  if (!(den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen)) return false;
  float T = dot3(Ng, C) * sgn;
  if (!(absDen * tnear < T)) return false;
  float tt = T / absDen;
  if (!(tt <= GVT_FLT_MAX)) return false;
  t = tt; u = U / absDen; v = V / absDen;
  return true;
}

// The same test with the geometric normal taken from the slot and the divisions left to the caller: returns the
// un-divided (T, U, V, |den|); t = T/|den| decides the order, u = U/|den| and v = V/|den| are needed for the winner only.
__device__ inline bool tri_test_raw(V3 O, V3 D, V3 v0, V3 e1, V3 e2, V3 Ng, float tnear, float &T, float &U, float &V, float &absDen) {
  V3 C = sub3(v0, O);
  V3 R = cross3(D, C);
  float den = dot3(Ng, D);
  absDen = fabsf(den);
  float sgn = (den < 0.f) ? -1.f : 1.f;
  U = dot3(R, e2) * sgn;
  V = dot3(R, e1) * sgn;
  if (!(den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen)) return false;
  T = dot3(Ng, C) * sgn;
  return absDen * tnear < T;
}

// The distance beyond which a box (or a stored stack entry) is culled against the best hit so far: the best t plus a relative slack of 2^-10.
// The closest hit is DEFINED as the arg-min of the triangle test over all triangles (ties to the lower primID: what a brute-force loop returns,
// oracle/gvt_oracle.c use_bvh = 0).  For a ray that grazes a triangle almost in its plane the test's t is noise of relative size ~1e-4: it can come
// out EARLIER than the ray's entry into that triangle's own (padded) box.  Culled at exactly the best t, the box holding a duplicate with a lower
// primID -- or a triangle whose noisy t is smaller still -- was then skipped depending on the traversal order (round 5: 3 of 1,500 fuzz seeds against
// the brute-force loop; the checker's tree had the same flaw).  With the slack every kernel returns the definition's hit on those too.
__host__ __device__ inline float cull_bound(float best_t) { return __builtin_fmaf(__builtin_fabsf(best_t), 0x1p-10f, best_t); }

// RandEngine::rng (core/math/RandEngine.h:43-56)
__host__ __device__ inline uint32_t rotl32(uint32_t r, int n) { return (r << n) | (r >> (32 - n)); }
__host__ __device__ inline float gvt_rng(uint32_t &seed) {
  uint32_t x, y, z;
  x = (seed >> 16) + 4125832013u;
  y = (seed & 0xffff) + 814584116u;
  z = 542;
  x *= 255519323u;
  x = rotl32(x, 13);
  y *= 3166389663u;
  y = rotl32(y, 17);
  z -= rotl32(z, 11);
  z = rotl32(z, 27);
  seed = x ^ y ^ z;
  return ((float)(seed & 0x00FFFFFF) / (float)0x01000000);
}
__host__ __device__ inline float gvt_fastrand01(uint32_t &seed) { return 0.0f + gvt_rng(seed) * (1.0f - 0.0f); }
// RandEngine::fastrand(uint*,min,max) LCG (:78-81)
__host__ __device__ inline float gvt_fastrand_lcg(uint32_t &seed, float mn, float mx) {
  const float ff = (1.0f / 65535.0f);
  seed = 214013u * seed + 2531011u;
  return mn + (seed >> 16) * ff * (mx - mn);
}
// One RNG stream per ray.  The reference seeds one engine per TBB chunk (EmbreeMeshAdapter.cpp:446-447), which is schedule
// dependent; a per-ray stream is order independent.  Two ways a stream starts:
//  - Adapter::trace on a host RayVector (gvt_hip_trace): keyed on (call seed, index in rayList); bytes 64..67 of the incoming
//    rays are ignored (a GraviT host leaves them uninitialised);
//  - rays born on the device (camera) carry their stream word from birth: camera_stream_word(list index of the primary ray), and
//    every later draw advances the word the ray carries.  A word of 0 means "no stream yet" and falls back to the first rule.
__host__ __device__ inline uint32_t ray_stream_seed(uint32_t seed, uint64_t index) {
  uint32_t s = seed ^ (uint32_t)(index * 0x9E3779B9u) ^ (uint32_t)(index >> 32);
  s ^= s >> 16; s *= 0x85EBCA6Bu; s ^= s >> 13; s *= 0xC2B2AE35u; s ^= s >> 16;
  return s;
}
// ridx = pixel * samples^2 + sub-sample: the position of the ray in generateRays' own list (gvtCamera.cpp:262-305), whatever order
// the rays are enumerated in here (tiles).  Never 0.
__host__ __device__ inline uint32_t camera_stream_word(uint64_t ridx) {
  const uint32_t s = ray_stream_seed(0x243F6A88u, ridx);
  return s ? s : 0x9E3779B9u;
}

// wave64 helpers
__device__ inline unsigned lane_id() { return __lane_id(); }
// the wave's mask of a per-lane condition (bool in, no widening to int as with HIP's __ballot)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline unsigned lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// wave-aggregated slot allocation: one atomic per wave for all lanes with `want`
__device__ inline unsigned wave_alloc(unsigned *counter, bool want) {
  unsigned long long mask = ballot64(want);
  if (mask == 0ull) return 0u;
  unsigned base = 0;
  int leader = __ffsll((long long)mask) - 1;
  if ((int)lane_id() == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
  base = __shfl(base, leader);
  return base + lanes_below(mask);
}

// block-aggregated slot allocation: wave ballot -> LDS counter -> ONE global atomic per block and output list.
// `sh` points at two LDS words reserved for this call site (count, base), zeroed before the first __syncthreads of the
// kernel.  Must be reached by every thread of the block.
__device__ inline unsigned block_alloc(unsigned *counter, bool want, unsigned *sh) {
  const unsigned long long mask = ballot64(want);
  unsigned woff = 0;
  if (mask) {
    const int leader = __ffsll((long long)mask) - 1;
    if ((int)lane_id() == leader) woff = atomicAdd(&sh[0], (unsigned)__popcll(mask));
    woff = __shfl(woff, leader);
  }
  __syncthreads();
  if (threadIdx.x == 0 && sh[0]) sh[1] = atomicAdd(counter, sh[0]);
  __syncthreads();
  return sh[1] + woff + lanes_below(mask);
}

// ---------------------------------------------------------------------------------------------
// Quad-cooperative fetch of 64-byte records (BVH nodes, triangle slots).
// A wave whose 64 lanes each read their own 64-byte record with 4 x global_load_dwordx4 spends 4 x 64 L1 tag
// look-ups on 64 cache lines; the vector L1 is then the busiest unit of the traversal kernel (TCP busy 82 %).
// Here the 4 lanes of a quad fetch the quad's 4 records together -- instruction j reads record j, lane k its 16-byte
// piece k, so a quad touches ONE line per instruction -- and transpose the pieces with DPP quad permutes (two
// butterfly stages, VALU only).  Measured on MI355X (tools/gather_bench.hip, dependent node chase, L2-resident set):
// 110 -> 202 G records/s.  Every lane of the wave must call it (EXEC all ones); `want` = this lane needs its record.
// ---------------------------------------------------------------------------------------------
template <int CTRL> __device__ inline int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ inline float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }

template <int XOR_CTRL> __device__ inline void quad_xchg(float &a, float &b, bool upper) {
  const float send = upper ? a : b;      // the lower lane of the pair keeps a and gives b away, the upper one the reverse
  const float recv = dpp_f<XOR_CTRL>(send);
  a = upper ? recv : a;
  b = upper ? b : recv;
}
template <int XOR_CTRL> __device__ inline void quad_xchg4(float4 &a, float4 &b, bool upper) {
  quad_xchg<XOR_CTRL>(a.x, b.x, upper); quad_xchg<XOR_CTRL>(a.y, b.y, upper);
  quad_xchg<XOR_CTRL>(a.z, b.z, upper); quad_xchg<XOR_CTRL>(a.w, b.w, upper);
}

__device__ inline void quad_fetch64(const float4 *__restrict__ base, unsigned rec, bool want, float4 &r0, float4 &r1, float4 &r2, float4 &r3) {
  const unsigned k = lane_id() & 3u;
  const int w = want ? 1 : 0;
  float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
  { const unsigned r = (unsigned)dpp_i<0x00>((int)rec); if (dpp_i<0x00>(w)) p0 = base[(size_t)r * 4 + k]; }
  { const unsigned r = (unsigned)dpp_i<0x55>((int)rec); if (dpp_i<0x55>(w)) p1 = base[(size_t)r * 4 + k]; }
  { const unsigned r = (unsigned)dpp_i<0xAA>((int)rec); if (dpp_i<0xAA>(w)) p2 = base[(size_t)r * 4 + k]; }
  { const unsigned r = (unsigned)dpp_i<0xFF>((int)rec); if (dpp_i<0xFF>(w)) p3 = base[(size_t)r * 4 + k]; }
  // lane k holds piece k of records 0..3; lane j needs pieces 0..3 of record j: a 4x4 transpose inside the quad
  const bool odd = (k & 1u) != 0u, hi = (k & 2u) != 0u;
  quad_xchg4<0xB1>(p0, p1, odd); quad_xchg4<0xB1>(p2, p3, odd); // quad_perm [1,0,3,2]
  quad_xchg4<0x4E>(p0, p2, hi);  quad_xchg4<0x4E>(p1, p3, hi);  // quad_perm [2,3,0,1]
  r0 = p0; r1 = p1; r2 = p2; r3 = p3;
}

// ---- top-level instance test (BVH::intersect + RayPacketIntersection), shared by the shuffle kernels and k_trace's sink ----
// RayPacket.h fastmin/fastmax: (a<b)?a:b / (a>b)?a:b
__device__ inline float fmin_ref(float a, float b) { return (a < b) ? a : b; }
__device__ inline float fmax_ref(float a, float b) { return (a > b) ? a : b; }

// The top-level instance set on the device: instance boxes in the leaf order of the reference's BVH (blo.w = instance id) and,
// for larger sets, the BVH's own nodes (accel/BVH.cpp:77-171 restated on the host, sched.hip): node k covers the leaves
// [first, first + count) of that order; nlo[k].w = left child (inner node) or ~first (leaf), nhi[k].w = right child or count.
struct TopDev {
  const float4 *blo, *bhi;
  int n_inst;
  const float4 *nlo, *nhi; // null / n_nodes == 0: no tree (linear scan in leaf order)
  int n_nodes;
};
#define GVT_TOP_BVH_MIN 48 // instance sets smaller than this are scanned linearly (no stack, no dependent node fetches)

// RayPacketIntersection::intersect for one ray and one box (RayPacket.h:111-193): entry / exit distances in the reference's order
__device__ inline void top_slab(const float4 lo, const float4 hi, float ox, float oy, float oz, float dx, float dy, float dz, float &tnear, float &tfar) {
  const float lx = (lo.x - ox) * dx, ly = (lo.y - oy) * dy, lz = (lo.z - oz) * dz;
  const float ux = (hi.x - ox) * dx, uy = (hi.y - oy) * dy, uz = (hi.z - oz) * dz;
  const float minx = fmin_ref(lx, ux), maxx = fmax_ref(lx, ux);
  const float miny = fmin_ref(ly, uy), maxy = fmax_ref(ly, uy);
  const float minz = fmin_ref(lz, uz), maxz = fmax_ref(lz, uz);
  tnear = fmax_ref(fmax_ref(minx, miny), minz);
  tfar = fmin_ref(fmin_ref(maxx, maxy), maxz);
}

// per ray: nearest other instance box with tfar>tnear && tnear>eps && t>tnear (BVH::intersect, accel/BVH.h:61-135); instances are
// visited in the reference BVH's leaf order so that equal entry distances resolve identically.  With nodes: the reference's own
// traversal -- left child first, a node is entered when tfar > tnear && t > tnear with the ray's current t (RayPacket.h:190, update
// = false) -- which visits the same leaves in the same order minus those a node test prunes (they could not have passed).
__device__ inline int top_nearest(const float4 a, const float4 b, const TopDev &T, int from, float &ret_t) {
  const float ox = a.x, oy = a.y, oz = a.z;
  const float dx = 1.f / b.x, dy = 1.f / b.y, dz = 1.f / b.z;
  float t = b.w; // ray t_max
  int next = -1;
  ret_t = GVT_FLT_MAX;
#define GVT_TOP_LEAF(K)                                                                  \
  {                                                                                      \
    const float4 lo_ = T.blo[K], hi_ = T.bhi[K];                                          \
    const int inst_ = __float_as_int(lo_.w);                                             \
    if (from != inst_) {                                                                 \
      float tn_, tf_;                                                                    \
      top_slab(lo_, hi_, ox, oy, oz, dx, dy, dz, tn_, tf_);                               \
      if (tf_ > tn_ && tn_ > GVT_RAY_EPSILON && t > tn_) {                                \
        t = tn_;                                                                         \
        if (ret_t > t) { next = inst_; ret_t = t; }                                      \
      }                                                                                  \
    }                                                                                    \
  }
  if (T.n_nodes > 0 && T.n_inst >= GVT_TOP_BVH_MIN) {
    int stack[48];
    int sp = 0, cur = 0;
    bool fallback = false;
    for (;;) {
      const float4 nl = T.nlo[cur], nh = T.nhi[cur];
      float tn, tf;
      top_slab(nl, nh, ox, oy, oz, dx, dy, dz, tn, tf);
      bool pop = true;
      if (tf > tn && t > tn) {
        const int l = __float_as_int(nl.w);
        if (l < 0) { // leaf: its instances in order
          const int first = ~l, count = __float_as_int(nh.w);
          for (int k = first; k < first + count; k++) GVT_TOP_LEAF(k)
        } else {
          if (sp == 48) { fallback = true; break; }
          stack[sp++] = __float_as_int(nh.w); // right child later
          cur = l;
          pop = false;
        }
      }
      if (pop) {
        if (sp == 0) break;
        cur = stack[--sp];
      }
    }
    if (!fallback) return next;
    t = b.w; next = -1; ret_t = GVT_FLT_MAX; // a tree deeper than the stack: scan (same result)
  }
  for (int k = 0; k < T.n_inst; k++) GVT_TOP_LEAF(k)
#undef GVT_TOP_LEAF
  return next;
}

// ---------------------------------------------------------------------------------------------
// Known misses: an image-identical shortcut of shuffleRays (NOT in the reference; restated in the checker, oracle/gvt_oracle.c).
// shuffleRays (TracerBase.h:392-400) sends a ray that left instance A without a hit to the nearest other instance box ahead of it,
// its origin advanced by 95 % of the distance (:393).  Where the boxes of A and B overlap, a ray that has crossed both without a hit
// is handed back and forth -- A, B, A, B, ..., five or six hops until the remaining distance falls below the box test's 1e-6 -- and every
// hop is a full traversal (and, between ranks, an exchange) that cannot find anything: it is the same half-line from an origin further
// along, and the instance held nothing on the longer one.  With the shortcut on, a ray carries the instances it has crossed without a
// hit on its current straight segment (plane 5); when shuffleRays' choice is one of them, the trace there is taken as the miss it must
// be: the origin advance is replayed with the reference's arithmetic and the next choice is made as if the ray came from there.  A
// bounce starts a new segment (list cleared), a shadow ray is born with an empty list.  The ray that finally reaches a queue or the
// framebuffer is bit for bit the ray the reference's hops deliver; only the number of rays traced and sent drops.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline bool km_has(const uint32_t km[3], int inst) {
  if (inst < 0 || inst >= 65535) return false;
  const uint32_t v = (uint32_t)(inst + 1);
  return (km[0] & 0xffffu) == v || (km[0] >> 16) == v || (km[1] & 0xffffu) == v || (km[1] >> 16) == v || (km[2] & 0xffffu) == v || (km[2] >> 16) == v;
}
__host__ __device__ inline void km_add(uint32_t km[3], int inst) { // first free entry; a full list forgets its oldest entry (forgetting only costs a trace)
  if (inst < 0 || inst >= 65535 || km_has(km, inst)) return;
  const uint32_t v = (uint32_t)(inst + 1);
  // (written out: a loop with early returns keeps its index dynamic and the three words in scratch memory)
  if (!(km[0] & 0xffffu)) { km[0] |= v; return; }
  if (!(km[0] >> 16)) { km[0] |= v << 16; return; }
  if (!(km[1] & 0xffffu)) { km[1] |= v; return; }
  if (!(km[1] >> 16)) { km[1] |= v << 16; return; }
  if (!(km[2] & 0xffffu)) { km[2] |= v; return; }
  if (!(km[2] >> 16)) { km[2] |= v << 16; return; }
  km[0] = (km[0] >> 16) | (km[1] << 16);
  km[1] = (km[1] >> 16) | (km[2] << 16);
  km[2] = (km[2] >> 16) | (v << 16);
}
#define GVT_KM_MAX_HOPS 64
// shuffleRays' decision for one ray (a = origin | t_min, b = direction | t_max) leaving instance `from`, with the shortcut: the instance it
// goes on in -- its origin advanced, `walked` set when that advance has been applied to `a` here -- or -1.  `from` joins the list.
__device__ inline int shuffle_walk(float4 &a, const float4 b, const TopDev &T, int from, uint32_t km[3], float &ret_t, bool &walked) {
  km_add(km, from);
  walked = false;
  int next = top_nearest(a, b, T, from, ret_t);
  if (next < 0 || !km_has(km, next)) return next; // the ordinary case: the caller advances the origin by ret_t (TracerBase.h:393)
  walked = true;
  for (int hop = 0;; hop++) {
    const V3 o = add3(mk3(a.x, a.y, a.z), scl3(mk3(b.x, b.y, b.z), ret_t * 0.95f)); // :393, replayed
    a.x = o.x; a.y = o.y; a.z = o.z;
    if (!km_has(km, next) || hop >= GVT_KM_MAX_HOPS) return next;
    from = next; // crossed without a hit before: as if traced there again and forwarded as it is
    next = top_nearest(a, b, T, from, ret_t);
    if (next < 0) return -1;
  }
}
