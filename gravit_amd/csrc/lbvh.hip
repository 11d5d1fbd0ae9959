// lbvh.hip -- acceleration-structure build on the device (replaces rtcCommit of
// EmbreeMeshAdapter::EmbreeMeshAdapter, EmbreeMeshAdapter.cpp:125-162).
//
// Pipeline (all on the adapter stream, one pass over HBM each):
//   tri bounds + scene box  ->  63-bit Morton keys  ->  radix sort (rocPRIM)  ->  Karras 2012 topology
//   ->  node boxes from a base-32 range-union table over the sorted triangle boxes (no inter-workgroup hand-off)
//   ->  collapse subtrees of <= GVT_LEAF_MAX triangles into leaves, compact live nodes (prefix scan)
//   ->  emit 64-byte binary nodes (both child boxes in the parent) + 64-byte triangle slots in leaf order
//   ->  collapse to 4-wide nodes, breadth first, child boxes quantised to 8 bits on a per-node grid (build_nodes4): the layout
//       k_trace traverses.
// Results of the closest/any-hit queries do not depend on the tree (conservative, padded boxes); only
// speed does.
#include <string.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gvt_internal.h"
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>

namespace {

__device__ inline unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  union { unsigned u; float f; } c;
  c.u = u;
  return c.f;
}

__global__ __launch_bounds__(256) void k_tri_bounds(const float *__restrict__ verts, const int *__restrict__ tris, unsigned n,
                                                    float4 *__restrict__ plo, float4 *__restrict__ phi) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int a = tris[3 * i], b = tris[3 * i + 1], c = tris[3 * i + 2];
  float lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    float x = verts[3 * a + k], y = verts[3 * b + k], z = verts[3 * c + k];
    lo[k] = fminf(x, fminf(y, z));
    hi[k] = fmaxf(x, fmaxf(y, z));
  }
  plo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
  phi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// 32:1 box reduction: out[k] = union of in[32k .. 32k+31].  Used (a) to reduce the triangle boxes to the scene box without
// atomics and (b) to build the levels of the range-union table over the SORTED triangle boxes (k_node_boxes).
__global__ __launch_bounds__(256) void k_reduce32(const float4 *__restrict__ in_lo, const float4 *__restrict__ in_hi, unsigned n_in,
                                                  float4 *__restrict__ out_lo, float4 *__restrict__ out_hi) {
  const unsigned lane = threadIdx.x & 31u;
  const unsigned grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; // one 32-lane half-wave per output box
  const unsigned n_out = (n_in + 31u) / 32u;
  if (grp >= n_out) return;
  const unsigned i = grp * 32u + lane;
  float4 l = make_float4(GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX, 0.f), h = make_float4(-GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX, 0.f);
  if (i < n_in) { l = in_lo[i]; h = in_hi[i]; }
  for (int off = 16; off > 0; off >>= 1) {
    l.x = fminf(l.x, __shfl_xor(l.x, off)); l.y = fminf(l.y, __shfl_xor(l.y, off)); l.z = fminf(l.z, __shfl_xor(l.z, off));
    h.x = fmaxf(h.x, __shfl_xor(h.x, off)); h.y = fmaxf(h.y, __shfl_xor(h.y, off)); h.z = fmaxf(h.z, __shfl_xor(h.z, off));
  }
  if (lane == 0) { out_lo[grp] = l; out_hi[grp] = h; }
}

// The scene box's first reduction level straight from the vertices: out[k] = union of the boxes of triangles 32k .. 32k+31 (the
// per-triangle boxes themselves are never stored: the Morton codes and the sorted boxes are computed from the vertices again)
__device__ inline void tri_box(const float *__restrict__ verts, const int *__restrict__ tris, unsigned i, float lo[3], float hi[3]) {
  const int a = tris[3 * i], b = tris[3 * i + 1], c = tris[3 * i + 2];
  for (int k = 0; k < 3; k++) {
    const float x = verts[3 * a + k], y = verts[3 * b + k], z = verts[3 * c + k];
    lo[k] = fminf(x, fminf(y, z));
    hi[k] = fmaxf(x, fmaxf(y, z));
  }
}
__global__ __launch_bounds__(256) void k_tri_bounds32(const float *__restrict__ verts, const int *__restrict__ tris, unsigned n,
                                                      float4 *__restrict__ out_lo, float4 *__restrict__ out_hi) {
  const unsigned lane = threadIdx.x & 31u;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = i >> 5;
  if (grp >= (n + 31u) / 32u) return; // (whole 32-lane groups leave together)
  float lo[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, hi[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
  if (i < n) tri_box(verts, tris, i, lo, hi);
  for (int off = 16; off > 0; off >>= 1)
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
  if (lane == 0) { out_lo[grp] = make_float4(lo[0], lo[1], lo[2], 0.f); out_hi[grp] = make_float4(hi[0], hi[1], hi[2], 0.f); }
}

__device__ inline unsigned long long expand21(unsigned v) { // 21 bits -> every third bit
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// The last <= 32 boxes of the scene-box reduction folded into ONE record on the device: scene[0..2] = lo, [3..5] = hi, [6] = 1 / largest extent (the Morton
// scale), [7] = pad (what keeps the slab test conservative w.r.t. the triangle test's rounding).  The kernels behind it read the record; the host fetches it with
// the node count later -- a build used to stop here for a read-back (copy, synchronisation, launch latency: ~100 us of a 4.2 ms build)
__global__ void k_scene_box(const float4 *__restrict__ lo, const float4 *__restrict__ hi, unsigned cnt, float *__restrict__ scene) {
  float l[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, h[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
  for (unsigned i = threadIdx.x; i < cnt; i += 64) {
    const float4 a = lo[i], b = hi[i];
    l[0] = fminf(l[0], a.x); l[1] = fminf(l[1], a.y); l[2] = fminf(l[2], a.z);
    h[0] = fmaxf(h[0], b.x); h[1] = fmaxf(h[1], b.y); h[2] = fmaxf(h[2], b.z);
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int k = 0; k < 3; k++) { l[k] = fminf(l[k], __shfl_xor(l[k], off)); h[k] = fmaxf(h[k], __shfl_xor(h[k], off)); }
  if (threadIdx.x) return;
  float ext = 0.f;
  for (int k = 0; k < 3; k++) { scene[k] = l[k]; scene[3 + k] = h[k]; ext = fmaxf(ext, fabsf(l[k])); ext = fmaxf(ext, fabsf(h[k])); }
  // ONE scale for the three axes (the largest extent): Morton cells are cubes whatever the shape of the mesh's box.  Normalising
  // every axis by its own extent made the cells of a 1 : 2 : 4 box -- a tile of a domain decomposition -- as elongated as the
  // box, and the tree's nodes with them (same box, A/B: 8 soup tiles 1.70 -> 1.66 ms, bunny.conf 0.343 -> 0.327, the cube-shaped soup unchanged)
  const float em = fmaxf(h[0] - l[0], fmaxf(h[1] - l[1], h[2] - l[2]));
  scene[6] = em > 0 ? 1.f / em : 0.f;
  scene[7] = ext * 1e-5f;
}

// key = the top `bits` bits of the 63-bit Morton code of the triangle's centre (bits a multiple of 3, chosen by the host from the triangle count: cells 64 times
// finer per axis than the mean triangle spacing -- the radix sort then runs ceil(bits / 8) passes instead of 8; triangles of one cell keep their input order,
// Karras' tie-break)
__global__ __launch_bounds__(256) void k_morton(const float *__restrict__ verts, const int *__restrict__ tris, unsigned n, const float *__restrict__ scene, int bits,
                                                unsigned long long *__restrict__ keys, unsigned *__restrict__ vals) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lo[3], hi[3];
  tri_box(verts, tris, i, lo, hi);
  const float iso = scene[6];
  float cx = (0.5f * (lo[0] + hi[0]) - scene[0]) * iso;
  float cy = (0.5f * (lo[1] + hi[1]) - scene[1]) * iso;
  float cz = (0.5f * (lo[2] + hi[2]) - scene[2]) * iso;
  unsigned qx = (unsigned)fminf(fmaxf(cx * 2097152.f, 0.f), 2097151.f);
  unsigned qy = (unsigned)fminf(fmaxf(cy * 2097152.f, 0.f), 2097151.f);
  unsigned qz = (unsigned)fminf(fmaxf(cz * 2097152.f, 0.f), 2097151.f);
  keys[i] = ((expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz)) >> (63 - bits);
  vals[i] = i;
}

// common-prefix length of sorted keys i and j, ties broken by position (Karras 2012, sec. 4)
__device__ inline int delta(const unsigned long long *__restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz(i ^ j);
  return __clzll((long long)(a ^ b));
}

// child encoding in the temporary tree: >= 0 inner node, < 0 leaf at sorted position ~c
__global__ __launch_bounds__(256) void k_karras(const unsigned long long *__restrict__ keys, int n, int *__restrict__ child_l,
                                                int *__restrict__ child_r, int *__restrict__ rfirst, int *__restrict__ rlast) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + min(d, 0);
  int first = min(i, j), last = max(i, j);
  int cl, cr;
  cl = (first == gamma) ? ~gamma : gamma;
  cr = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  child_l[i] = cl; child_r[i] = cr;
  rfirst[i] = first; rlast[i] = last;
}

__device__ inline void child_box(int c, const float4 *__restrict__ slo, const float4 *__restrict__ shi, const float4 *__restrict__ ilo,
                                 const float4 *__restrict__ ihi, float4 &lo, float4 &hi) {
  if (c < 0) { lo = slo[~c]; hi = shi[~c]; }
  else { lo = ilo[c]; hi = ihi[c]; }
}

// Box of every inner node = union of the sorted triangle boxes of its Karras range [first, last], read from a base-32
// range-union table (level L holds the unions of aligned blocks of 32^L triangles): no inter-workgroup hand-off at all.
// (The first version walked up the tree with an arrival counter per node; on gfx950 that needs an agent-scope release and
// acquire per level -- L2 write-back + invalidate -- and took 71 of the 95 ms of a 10 M-triangle build.)
#define GVT_BOX_LEVELS 6
struct BoxLevels {
  const float4 *lo[GVT_BOX_LEVELS];
  const float4 *hi[GVT_BOX_LEVELS];
  int n_levels;
};
#define NB_CHUNK 1024 // k_node_boxes_chunk's chunk of sorted triangles (68 KB of LDS per block: two blocks per CU)
__device__ inline void node_box_from_table(unsigned a, const unsigned e, const BoxLevels &T, float4 &out_lo, float4 &out_hi) {
  float lx = GVT_FLT_MAX, ly = GVT_FLT_MAX, lz = GVT_FLT_MAX, hx = -GVT_FLT_MAX, hy = -GVT_FLT_MAX, hz = -GVT_FLT_MAX;
  int L = 0;
  unsigned S = 1u;
#define GVT_TAKE(LV, IDX) { const float4 l_ = T.lo[LV][IDX], h_ = T.hi[LV][IDX]; lx = fminf(lx, l_.x); ly = fminf(ly, l_.y); lz = fminf(lz, l_.z); \
                            hx = fmaxf(hx, h_.x); hy = fmaxf(hy, h_.y); hz = fmaxf(hz, h_.z); }
  for (;;) { // ascend: blocks of the current size until the position is aligned for the next size
    while (a < e && (a % (S * 32u)) != 0u && a + S <= e) { GVT_TAKE(L, a / S) a += S; }
    if (a + S * 32u <= e && L + 1 < T.n_levels) { L++; S *= 32u; } else break;
  }
  for (;;) { // descend: the remaining tail
    while (a + S <= e) { GVT_TAKE(L, a / S) a += S; }
    if (L == 0) break;
    L--; S /= 32u;
  }
#undef GVT_TAKE
  out_lo = make_float4(lx, ly, lz, 0.f);
  out_hi = make_float4(hx, hy, hz, 0.f);
}
// every inner node from the table (meshes too small for the chunked path; and the definition the chunked path is tested against)
__global__ __launch_bounds__(256) void k_node_boxes(int n_inner, const int *__restrict__ rfirst, const int *__restrict__ rlast, BoxLevels T,
                                                    float4 *__restrict__ ilo, float4 *__restrict__ ihi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  float4 lo, hi;
  node_box_from_table((unsigned)rfirst[i], (unsigned)rlast[i] + 1u, T, lo, hi);
  ilo[i] = lo; ihi[i] = hi;
}

// Boxes of the inner nodes whose Karras range lies inside ONE aligned chunk of NB_CHUNK sorted triangles, bottom-up in LDS (Karras' walk with arrival
// counters -- in LDS, where an atomic costs a few cycles and needs no cross-XCD visibility; the same walk over the whole tree in global memory was 71 of the
// 95 ms of round 1's build): the block loads its chunk's triangle boxes, every inner node inside the chunk tells its children who their parent is, then every
// leaf walks up: the first child to arrive at a node stops, the second one unions the two boxes and goes on, until the parent is a node that spans a chunk
// boundary (no parent recorded).  min / max are exact: the boxes are the ones the range-union table gives.  Nodes that DO span a boundary (the spines, ~3 % of
// the nodes) are listed per block for k_node_boxes_spine.
#define NB_THREADS 1024 // one sorted triangle / one node per thread: 16 waves per block keep enough loads in flight (at 256 threads x 4 the kernel ran at 1.5 TB/s)
__global__ __launch_bounds__(NB_THREADS) void k_node_boxes_chunk(int n, const int *__restrict__ child_l, const int *__restrict__ child_r, const int *__restrict__ rfirst,
                                                          const int *__restrict__ rlast, const float4 *__restrict__ slo, const float4 *__restrict__ shi,
                                                          float4 *__restrict__ ilo, float4 *__restrict__ ihi, int *__restrict__ spine, unsigned *__restrict__ spine_n) {
  __shared__ float lb[6][NB_CHUNK]; // the chunk's triangle boxes, one plane per component (conflict free)
  __shared__ float nb[6][NB_CHUNK]; // node boxes (node c0 + k at k)
  __shared__ int cl_s[NB_CHUNK], cr_s[NB_CHUNK];
  __shared__ short par_leaf[NB_CHUNK], par_node[NB_CHUNK]; // parent (local node number) of leaf / node k, -1: none inside the chunk
  __shared__ unsigned arrived[NB_CHUNK];
  __shared__ unsigned s_spine;
  constexpr int PER = NB_CHUNK / NB_THREADS;
  const int c0 = (int)blockIdx.x * NB_CHUNK, c1 = min(c0 + NB_CHUNK, n);
  if (threadIdx.x == 0) s_spine = 0u;
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int k = (int)threadIdx.x + NB_THREADS * m, i = c0 + k;
    par_leaf[k] = -1; par_node[k] = -1; arrived[k] = 0u; cl_s[k] = 0; cr_s[k] = 0;
    if (i < c1) {
      const float4 a = slo[i], b = shi[i];
      lb[0][k] = a.x; lb[1][k] = a.y; lb[2][k] = a.z; lb[3][k] = b.x; lb[4][k] = b.y; lb[5][k] = b.z;
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int k = (int)threadIdx.x + NB_THREADS * m, i = c0 + k;
    if (i >= c1 || i >= n - 1) continue;
    if (rfirst[i] >= c0 && rlast[i] < c1) { // inside: its children lie inside too
      const int l = child_l[i], r = child_r[i];
      cl_s[k] = l; cr_s[k] = r;
      if (l < 0) par_leaf[~l - c0] = (short)k; else par_node[l - c0] = (short)k;
      if (r < 0) par_leaf[~r - c0] = (short)k; else par_node[r - c0] = (short)k;
    } else spine[(size_t)blockIdx.x * NB_CHUNK + atomicAdd(&s_spine, 1u)] = i; // (LDS atomic; the order within the block's list does not matter)
  }
  __syncthreads();
#pragma unroll 1
  for (int m = 0; m < PER; m++) {
    const int k = (int)threadIdx.x + NB_THREADS * m;
    if (c0 + k >= c1) continue;
    int p = par_leaf[k];
    while (p >= 0) {
      // (release: this thread's box of the child it comes from is in LDS before the count; acquire: the sibling's is visible to the second arrival)
      if (__hip_atomic_fetch_add(&arrived[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;
      const int l = cl_s[p], r = cr_s[p];
      const float(*pl)[NB_CHUNK] = l < 0 ? lb : nb;
      const float(*pr)[NB_CHUNK] = r < 0 ? lb : nb;
      const int il = (l < 0 ? ~l : l) - c0, ir = (r < 0 ? ~r : r) - c0;
#pragma unroll
      for (int c = 0; c < 3; c++) { nb[c][p] = fminf(pl[c][il], pr[c][ir]); nb[3 + c][p] = fmaxf(pl[3 + c][il], pr[3 + c][ir]); }
      p = par_node[p];
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PER; m++) { // the finished nodes, coalesced
    const int k = (int)threadIdx.x + NB_THREADS * m;
    if (arrived[k] == 2u) {
      ilo[c0 + k] = make_float4(nb[0][k], nb[1][k], nb[2][k], 0.f);
      ihi[c0 + k] = make_float4(nb[3][k], nb[4][k], nb[5][k], 0.f);
    }
  }
  if (threadIdx.x == 0) spine_n[blockIdx.x] = s_spine;
}
// the nodes that span a chunk boundary, block by block from k_node_boxes_chunk's lists: the range-union table's walk (node_box_from_table), a wave per list
__global__ __launch_bounds__(64) void k_node_boxes_spine(const int *__restrict__ spine, const unsigned *__restrict__ spine_n, const int *__restrict__ rfirst,
                                                         const int *__restrict__ rlast, BoxLevels T, float4 *__restrict__ ilo, float4 *__restrict__ ihi) {
  const unsigned cnt = spine_n[blockIdx.x];
  for (unsigned k = threadIdx.x; k < cnt; k += 64) {
    const int i = spine[(size_t)blockIdx.x * NB_CHUNK + k];
    float4 lo, hi;
    node_box_from_table((unsigned)rfirst[i], (unsigned)rlast[i] + 1u, T, lo, hi);
    ilo[i] = lo; ihi[i] = hi;
  }
}

__global__ __launch_bounds__(256) void k_mark_live(int n_inner, const int *__restrict__ rfirst, const int *__restrict__ rlast,
                                                   unsigned *__restrict__ live, int leaf_max) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  live[i] = (rlast[i] - rfirst[i] + 1 > leaf_max) ? 1u : 0u;
}

__device__ inline int final_ref(int c, const int *__restrict__ rfirst, const int *__restrict__ rlast, const unsigned *__restrict__ newidx, int leaf_max) {
  if (c < 0) return leaf_ref((unsigned)(~c), 1u);
  int cnt = rlast[c] - rfirst[c] + 1;
  if (cnt <= leaf_max) return leaf_ref((unsigned)rfirst[c], (unsigned)cnt);
  return (int)newidx[c];
}

__global__ __launch_bounds__(256) void k_emit_nodes(int n_inner, const unsigned *__restrict__ live, const unsigned *__restrict__ newidx,
                                                    const float4 *__restrict__ slo, const float4 *__restrict__ shi,
                                                    const float4 *__restrict__ ilo, const float4 *__restrict__ ihi,
                                                    const int *__restrict__ child_l, const int *__restrict__ child_r, const int *__restrict__ rfirst,
                                                    const int *__restrict__ rlast, const float *__restrict__ scene, BvhNode *__restrict__ nodes, int leaf_max,
                                                    unsigned *__restrict__ leaf_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner || !live[i]) return;
  const float pad = scene[7];
  int cl = child_l[i], cr = child_r[i];
  float4 al, ah, bl, bh;
  child_box(cl, slo, shi, ilo, ihi, al, ah);
  child_box(cr, slo, shi, ilo, ihi, bl, bh);
  BvhNode nd;
  nd.n0 = make_float4(al.x - pad, ah.x + pad, al.y - pad, ah.y + pad);
  nd.n1 = make_float4(bl.x - pad, bh.x + pad, bl.y - pad, bh.y + pad);
  nd.n2 = make_float4(al.z - pad, ah.z + pad, bl.z - pad, bh.z + pad);
  int r0 = final_ref(cl, rfirst, rlast, newidx, leaf_max), r1 = final_ref(cr, rfirst, rlast, newidx, leaf_max);
  nd.n3 = make_float4(__int_as_float(r0), __int_as_float(r1), 0.f, 0.f);
  nodes[newidx[i]] = nd;
  // (no leaf counter: the emitted nodes form a binary tree, so it has one leaf more than nodes -- counting them here took one atomic per
  // wave on a single word, half of this kernel's 1.8 ms at 10 M triangles)
  if (!leaf_of) return;
  // every sorted triangle learns its leaf (first slot, count): the transposed leaf blocks of k_emit_trisq are laid out per leaf
  if (r0 < 0) { const unsigned code = (unsigned)~r0; for (unsigned k = 0; k < (code & 7u); k++) leaf_of[(code >> 3) + k] = code; }
  if (r1 < 0) { const unsigned code = (unsigned)~r1; for (unsigned k = 0; k < (code & 7u); k++) leaf_of[(code >> 3) + k] = code; }
}

__global__ __launch_bounds__(256) void k_emit_tris(const float *__restrict__ verts, const int *__restrict__ tris, const unsigned *__restrict__ sorted,
                                                   unsigned n, float4 *__restrict__ out, unsigned *__restrict__ slot_of, float4 *__restrict__ slo,
                                                   float4 *__restrict__ shi) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  unsigned p = sorted[s];
  int a = tris[3 * p], b = tris[3 * p + 1], c = tris[3 * p + 2];
  V3 v0 = ld3(verts + 3 * a), v1 = ld3(verts + 3 * b), v2 = ld3(verts + 3 * c);
  if (slo) { // the triangle's box at its sorted position (what the range-union table of k_node_boxes is built over)
    slo[s] = make_float4(fminf(v0.x, fminf(v1.x, v2.x)), fminf(v0.y, fminf(v1.y, v2.y)), fminf(v0.z, fminf(v1.z, v2.z)), 0.f);
    shi[s] = make_float4(fmaxf(v0.x, fmaxf(v1.x, v2.x)), fmaxf(v0.y, fmaxf(v1.y, v2.y)), fmaxf(v0.z, fmaxf(v1.z, v2.z)), 0.f);
  }
  V3 e1 = sub3(v0, v1), e2 = sub3(v2, v0); // same float ops as evaluating them per test (no contraction): bit-identical
  // the traversal reads (v0 | prim), e1, e2 (48 of the 64 bytes); the remaining six floats carry v1 and v2 as they are, so that the
  // shading kernel finds the whole triangle in the one line the traversal has just touched (EmbreeMeshAdapter.cpp:503-504 needs
  // cross(v1 - v0, v2 - v0) of the ORIGINAL vertices: -(e1 x e2) differs from it in the signs of zeros)
  out[4 * s + 0] = make_float4(v0.x, v0.y, v0.z, __int_as_float((int)p));
  out[4 * s + 1] = make_float4(e1.x, e1.y, e1.z, v1.x);
  out[4 * s + 2] = make_float4(e2.x, e2.y, e2.z, v1.y);
  out[4 * s + 3] = make_float4(v1.z, v2.x, v2.y, v2.z);
  slot_of[p] = s;
}

// Leaf blocks for the quad-per-ray traversal (quad_kernel.inc): the n <= 4 triangles of a leaf occupy n x 64 B starting at slot
// `first`, TRANSPOSED -- piece i (0: v0|prim, 1: e1, 2: e2, 3: Ng) of triangle k at float4 index 4*first + i*n + k -- so that the
// lanes of a quad, lane k testing triangle k, read n consecutive 16-byte pieces (one or two cache lines) per load instruction
// instead of n different lines.  Same float operations as k_emit_tris: bit-identical operands.
__global__ __launch_bounds__(256) void k_emit_trisq(const float *__restrict__ verts, const int *__restrict__ tris, const unsigned *__restrict__ sorted,
                                                    const unsigned *__restrict__ leaf_of, unsigned n, float4 *__restrict__ out) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const unsigned code = leaf_of[s], first = code >> 3, cnt = code & 7u, k = s - first;
  unsigned p = sorted[s];
  int a = tris[3 * p], b = tris[3 * p + 1], c = tris[3 * p + 2];
  V3 v0 = ld3(verts + 3 * a), v1 = ld3(verts + 3 * b), v2 = ld3(verts + 3 * c);
  V3 e1 = sub3(v0, v1), e2 = sub3(v2, v0);
  V3 Ng = cross3(e1, e2);
  float4 *blk = out + (size_t)4 * first;
  blk[0 * cnt + k] = make_float4(v0.x, v0.y, v0.z, __int_as_float((int)p));
  blk[1 * cnt + k] = make_float4(e1.x, e1.y, e1.z, 0.f);
  blk[2 * cnt + k] = make_float4(e2.x, e2.y, e2.z, 0.f);
  blk[3 * cnt + k] = make_float4(Ng.x, Ng.y, Ng.z, 0.f);
}

// n <= leaf_max: one node, child0 = the only leaf, child1 = empty leaf behind an inverted box
__global__ void k_single_node(const float4 *__restrict__ plo, const float4 *__restrict__ phi, unsigned n, float pad, BvhNode *nodes) {
  if (threadIdx.x || blockIdx.x) return;
  float lo[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, hi[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
  for (unsigned i = 0; i < n; i++) {
    lo[0] = fminf(lo[0], plo[i].x); lo[1] = fminf(lo[1], plo[i].y); lo[2] = fminf(lo[2], plo[i].z);
    hi[0] = fmaxf(hi[0], phi[i].x); hi[1] = fmaxf(hi[1], phi[i].y); hi[2] = fmaxf(hi[2], phi[i].z);
  }
  BvhNode nd;
  nd.n0 = make_float4(lo[0] - pad, hi[0] + pad, lo[1] - pad, hi[1] + pad);
  nd.n1 = make_float4(GVT_FLT_MAX, -GVT_FLT_MAX, GVT_FLT_MAX, -GVT_FLT_MAX);
  nd.n2 = make_float4(lo[2] - pad, hi[2] + pad, GVT_FLT_MAX, -GVT_FLT_MAX);
  nd.n3 = make_float4(__int_as_float(leaf_ref(0u, n)), __int_as_float(leaf_ref(0u, 0u)), 0.f, 0.f);
  nodes[0] = nd;
}

// Compressed 4-wide collapse, breadth first.  Every thread turns one binary node into one 4-wide node: it starts from the
// node's two children and replaces the inner child with the largest surface area by that child's two children until four
// slots are filled (or only leaves remain).  Inner children get consecutive indices in the next level (one atomic per
// wave), so the array ends up in breadth-first order.  Node index of frontier entry i = base_in + i.
struct Slot4 { float lo[3], hi[3]; int ref; };
__device__ inline void slot_from(const BvhNode &nd, int side, Slot4 &c) {
  const float4 xy = side ? nd.n1 : nd.n0;
  c.lo[0] = xy.x; c.hi[0] = xy.y; c.lo[1] = xy.z; c.hi[1] = xy.w;
  c.lo[2] = side ? nd.n2.z : nd.n2.x; c.hi[2] = side ? nd.n2.w : nd.n2.y;
  c.ref = __float_as_int(side ? nd.n3.y : nd.n3.x);
}
__device__ inline float slot_area(const Slot4 &c) {
  const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
// Level `level` of the collapse: its frontier size is levels[level], written by the launch before; the host launches several levels back to
// back with grids sized by a bound (4^level, at most every node) and reads the sizes once per batch -- a level used to cost a counter
// reset, a launch, a copy and a host synchronisation (14 levels at 10 M triangles).
#define GVT_COLLAPSE_BLOCK 512
#define GVT_COLLAPSE_LEVELS 96
__global__ __launch_bounds__(GVT_COLLAPSE_BLOCK, 4) void k_collapse4( // (4 waves per SIMD = two 512-thread blocks per CU: 128 registers; unbounded the compiler took 132 and ONE block was resident)
    const BvhNode *__restrict__ nodes, const int *__restrict__ fin, unsigned *__restrict__ levels, int level,
                                                                  int *__restrict__ fout, uint4 *__restrict__ nodes4, uint4 *__restrict__ nodes4q) {
  const unsigned n_in = levels[level];
  if (blockIdx.x * blockDim.x >= n_in) return; // (block-uniform)
  unsigned base_in = 0;
  for (int l = 0; l < level; l++) base_in += levels[l];
  unsigned *next_count = levels + level + 1;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n_in;
  Slot4 c[4];
  int n = 0;
  if (live) {
    const BvhNode nb = nodes[fin[i]];
    slot_from(nb, 0, c[0]); slot_from(nb, 1, c[1]);
    n = 2;
    if (c[1].ref == GVT_EMPTY_REF) n = 1; // single-leaf mesh (k_single_node)
    // (slots addressed through unrolled compares, not a run-time index: the four of them stay in registers)
#pragma unroll
    for (int it = 0; it < 2; it++) {
      if (n >= 4) break;
      int k = -1;
      float best = -1.f;
#pragma unroll
      for (int s = 0; s < 4; s++)
        if (s < n && c[s].ref >= 0) { const float a = slot_area(c[s]); if (a > best) { best = a; k = s; } }
      if (k < 0) break;
      int ref_k = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) if (s == k) ref_k = c[s].ref;
      const BvhNode nc = nodes[ref_k];
      Slot4 a0, a1;
      slot_from(nc, 0, a0); slot_from(nc, 1, a1);
#pragma unroll
      for (int s = 0; s < 4; s++) { if (s == k) c[s] = a0; if (s == n) c[s] = a1; }
      n++;
    }
  }
  // next-level indices for the inner children: ONE atomic per block (a single word takes ~90 atomics per microsecond: one per wave was
  // 47 K of them at 10 M triangles, half of the collapse's time)
  int want = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) want += (s < n && c[s].ref >= 0) ? 1 : 0;
  int incl = want;
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if ((int)lane_id() >= d) incl += v; }
  __shared__ unsigned s_wave[GVT_COLLAPSE_BLOCK / 64];
  __shared__ unsigned s_base;
  const unsigned wv = threadIdx.x >> 6;
  if (lane_id() == 63) s_wave[wv] = (unsigned)incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (unsigned k = 0; k < GVT_COLLAPSE_BLOCK / 64; k++) { const unsigned t = s_wave[k]; s_wave[k] = run; run += t; }
    s_base = run ? atomicAdd(next_count, run) : 0u;
  }
  __syncthreads();
  if (!live) return;
  const unsigned base = s_base;
  unsigned slot = base + s_wave[wv] + (unsigned)(incl - want);
#pragma unroll
  for (int s = 0; s < 4; s++)
    if (s < n && c[s].ref >= 0) { fout[slot] = c[s].ref; c[s].ref = (int)(base_in + n_in + slot); slot++; }
  // quantise (outwards, verified in double against the very expression the traversal decodes with)
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = 0u;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float o = c[0].lo[a], h = c[0].hi[a];
#pragma unroll
    for (int s = 1; s < 4; s++) if (s < n) { o = fminf(o, c[s].lo[a]); h = fmaxf(h, c[s].hi[a]); }
    int k2 = 0;
    (void)frexpf((h - o) * (1.0f / 255.0f), &k2);
    int e = k2 + 127;
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    uint32_t ql = 0, qh = 0;
    for (;;) {
      const float scale = __int_as_float(e << 23);
      bool ok = true;
      ql = 0; qh = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (s >= n || !ok) break;
        int l = (int)floorf(ldexpf(c[s].lo[a] - o, 127 - e)); // (/ scale, a power of two: the same correctly rounded quotient in one instruction)
        l = l < 0 ? 0 : (l > 255 ? 255 : l);
        while (l > 0 && (double)o + (double)l * (double)scale > (double)c[s].lo[a]) l--;
        int u = (int)ceilf(ldexpf(c[s].hi[a] - o, 127 - e));
        u = u < 0 ? 0 : u;
        while (u <= 255 && (double)o + (double)u * (double)scale < (double)c[s].hi[a]) u++;
        if (u > 255) { ok = false; break; }
        ql |= (uint32_t)l << (8 * s); qh |= (uint32_t)u << (8 * s);
      }
      if (ok || e >= 254) break;
      e++;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) if (s >= n) ql |= 255u << (8 * s); // unused slots: an inverted box (lo plane 255, hi plane 0) that no ray enters
    w[a] = __float_as_uint(o);
    w[a == 0 ? 3 : 13 + a] = (uint32_t)e << 23; // the grid step 2^(e-127) as a float: w0.w (x), w3.z (y), w3.w (z)
    w[4 + 2 * a] = ql; w[5 + 2 * a] = qh;
  }
#pragma unroll
  for (int s = 0; s < 4; s++) w[10 + s] = (uint32_t)(s < n ? c[s].ref : GVT_EMPTY_REF);
  uint4 *dst = nodes4 + (size_t)GVT_NODE4_F4 * (base_in + i);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]); dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  dst[2] = make_uint4(w[8], w[9], w[10], w[11]); dst[3] = make_uint4(w[12], w[13], w[14], w[15]);
  if (nodes4q) {
    // the same node for the quad-per-ray traversal: piece s (16 B) is all lane s of a quad needs of child s --
    //   x = ref_s, y = (lo.x, lo.y, lo.z, hi.x) bytes, z = (hi.y, hi.z) bytes | grid step of axis s as the upper half of its float
    //   (a power of two: the lower 16 bits are zero), w = grid origin of axis s (s < 3)
    const uint32_t step[3] = { w[3], w[14], w[15] };
    uint4 *dq = nodes4q + (size_t)GVT_NODE4_F4 * (base_in + i);
    for (int s = 0; s < 4; s++) {
      const uint32_t lx = (w[4] >> (8 * s)) & 0xffu, hx = (w[5] >> (8 * s)) & 0xffu, ly = (w[6] >> (8 * s)) & 0xffu, hy = (w[7] >> (8 * s)) & 0xffu,
                     lz = (w[8] >> (8 * s)) & 0xffu, hz = (w[9] >> (8 * s)) & 0xffu;
      dq[s] = make_uint4(w[10 + s], lx | (ly << 8) | (lz << 16) | (hx << 24), hy | (hz << 8) | (s < 3 ? (step[s] & 0xffff0000u) : 0u), s < 3 ? w[s] : 0u);
    }
  }
}

// Marks the binary nodes that become roots of W-wide nodes under k_collapse4's rule (start from the two children, replace the inner
// child of largest area by its two children until W slots are filled), level by level: the diagnostic behind "how many steps would
// an 8-wide layout take" (gvt_hip_wide_visit_stats).
__global__ __launch_bounds__(256) void k_collapse_mark(const BvhNode *__restrict__ nodes, const int *__restrict__ fin, unsigned n_in, int width,
                                                       int *__restrict__ fout, unsigned *__restrict__ next_count, unsigned char *__restrict__ marks) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  marks[fin[i]] = 1;
  Slot4 c[8];
  int n = 2;
  const BvhNode nb = nodes[fin[i]];
  slot_from(nb, 0, c[0]); slot_from(nb, 1, c[1]);
  if (c[1].ref == GVT_EMPTY_REF) n = 1;
  while (n < width) {
    int k = -1;
    float best = -1.f;
    for (int s = 0; s < n; s++)
      if (c[s].ref >= 0) { const float a = slot_area(c[s]); if (a > best) { best = a; k = s; } }
    if (k < 0) break;
    const BvhNode nc = nodes[c[k].ref];
    slot_from(nc, 0, c[k]); slot_from(nc, 1, c[n]);
    n++;
  }
  for (int s = 0; s < n; s++)
    if (c[s].ref >= 0) fout[atomicAdd(next_count, 1u)] = c[s].ref;
}

// The build's temporaries come out of ONE allocation (the context's scratch slot 21; given back when it is larger than 1 GiB) instead
// of thirty hipMalloc / hipFree pairs.  (GVT_HIP_BUILD_TRACE=1 prints where a build's time goes -- tools/build_probe.py: 10 M triangles
// take 4.1 ms, 2.4 Gtris/s; the first build of a process 13-18 ms, most of it the first kernel launches: code objects, rocPRIM.)
struct BuildArena {
  char *base = nullptr;
  size_t off = 0, cap = 0;
  template <typename T> int take(T **p, size_t n) {
    const size_t bytes = (sizeof(T) * (n ? n : 1) + 255) & ~(size_t)255;
    if (!base || off + bytes > cap) { set_error("BVH build: scratch arena too small (%zu + %zu > %zu)", off, bytes, cap); return GVT_HIP_ERR_DEVICE; }
    *p = (T *)(base + off);
    off += bytes;
    return 0;
  }
};
static double g_alloc_ms = 0.0; // (GVT_HIP_BUILD_TRACE: host time spent in hipMalloc during a build)
template <typename T> int dalloc(T **p, size_t n) {
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc((void **)p, sizeof(T) * (n ? n : 1));
  g_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (e != hipSuccess) { set_error("hipMalloc(%zu B) failed: %s", sizeof(T) * n, hipGetErrorString(e)); return GVT_HIP_ERR_DEVICE; }
  return 0;
}

} // namespace

static int build_nodes4(gvt_hip_mesh *M, BuildArena *A);

// Reduces n boxes 32:1 level by level into `levels` (device arrays out of the build's arena); returns the table.
static int build_box_levels(const float4 *lo0, const float4 *hi0, unsigned n, BuildArena &A, BoxLevels &T, hipStream_t st) {
  T.n_levels = 1;
  T.lo[0] = lo0; T.hi[0] = hi0;
  unsigned cnt = n;
  while (cnt > 32u && T.n_levels < GVT_BOX_LEVELS) {
    const unsigned n_out = (cnt + 31u) / 32u;
    float4 *l = nullptr, *h = nullptr;
    int rc = A.take(&l, n_out);
    if (!rc) rc = A.take(&h, n_out);
    if (rc) return rc;
    const unsigned threads = n_out * 32u;
    k_reduce32<<<(threads + 255u) / 256u, 256, 0, st>>>(T.lo[T.n_levels - 1], T.hi[T.n_levels - 1], cnt, l, h);
    T.lo[T.n_levels] = l; T.hi[T.n_levels] = h;
    T.n_levels++;
    cnt = n_out;
  }
  return 0;
}

// The per-mesh statistic behind the packet choice: sum over the INNER nodes of area(node) / area(root), i.e. the inner nodes a random line through
// the mesh's box pierces, in 16.16 fixed point (integer adds: the same sum whatever the order).  Every node adds the boxes of its inner children.
__global__ __launch_bounds__(256) void k_sah_sum(const BvhNode *__restrict__ nodes, unsigned n, float inv_root_area, unsigned long long *acc) {
  // a fixed grid striding over the nodes, ONE atomic per block (a word takes ~90 atomics per microsecond: one per wave of 6.4 M nodes was 1.2 ms of the build)
  __shared__ unsigned long long s_part[4];
  unsigned long long v = 0ull;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const BvhNode nd = nodes[i];
    Slot4 c;
    float s = 0.f;
    slot_from(nd, 0, c); if (c.ref >= 0) s += slot_area(c);
    slot_from(nd, 1, c); if (c.ref >= 0) s += slot_area(c);
    v += (unsigned long long)(fminf(s * inv_root_area, 4.0f) * 65536.0f);
  }
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if (lane_id() == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3]; if (t) atomicAdd(acc, t); }
}

int build_lbvh(gvt_hip_mesh *M) {
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  const unsigned n = (unsigned)M->nT;
  if (n == 0) { M->nNodes = 0; M->d_nodes = nullptr; M->d_tri = nullptr; return 0; }

  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, st));
  const bool trace = getenv("GVT_HIP_BUILD_TRACE") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!trace) return;
    hipStreamSynchronize(st);
    fprintf(stderr, "[build] %-28s %8.3f ms (hipMalloc so far %.3f ms)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), g_alloc_ms);
  };
  g_alloc_ms = 0.0;

  float4 *plo = nullptr, *phi = nullptr, *slo = nullptr, *shi = nullptr, *ilo = nullptr, *ihi = nullptr;
  unsigned long long *keys = nullptr, *keys2 = nullptr;
  unsigned *vals = nullptr, *sorted = nullptr, *live = nullptr, *newidx = nullptr, *leaf_of = nullptr;
  const int leaf_max = C.leaf_max < 1 ? 1 : (C.leaf_max > 4 ? 4 : C.leaf_max);
  const bool want_q = C.quad != 0; // the quad-per-ray layouts (nodes4q + transposed leaf blocks)
  M->leaf_max = leaf_max;
  int *cl = nullptr, *cr = nullptr, *rf = nullptr, *rl = nullptr;
  void *tmp = nullptr, *tmp2 = nullptr;
  BuildArena A;
  size_t tb_sort = 0, tb_scan = 0;
  int rc = 0;
  const unsigned B = 256, G = (n + B - 1) / B;
  const int n_inner = (int)n - 1;
  const unsigned n32 = (n + 31u) / 32u;
  float pad = 0.f;
  unsigned long long *sah_acc = nullptr, sah_host = 0ull;
  float *d_scene = nullptr;
  float *h_scene = (float *)(C.h_pinned + 32); // pinned words 32..39 of the context: an asynchronous copy into pageable memory would wait for the device at once
  // Morton key width: 3 x (bits per axis), the cells 64 times finer per axis than the mean triangle spacing n^(1/3) -- 42 bits at 10 M triangles (6 radix
  // passes of 8 bits instead of the 8 the full 63 need), 36 at 70 K; never fewer than 30, never more than 63
  int key_bits = 18;
  for (unsigned k = n; k > 1; k >>= 3) key_bits += 3;
  key_bits = key_bits < 30 ? 30 : (key_bits > 63 ? 63 : key_bits);

#define OK(x) do { if ((rc = (x)) != 0) goto done; } while (0)
#define HOK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(_e)); rc = GVT_HIP_ERR_DEVICE; goto done; } } while (0)

  { // every temporary of the build, with 256-byte slack each (sizes of the library calls queried first)
    if ((int)n > leaf_max) {
      HOK(rocprim::radix_sort_pairs(nullptr, tb_sort, keys, keys2, vals, sorted, n, 0, key_bits, st));
      HOK(rocprim::exclusive_scan(nullptr, tb_scan, live, newidx, 0u, (size_t)(n > 1 ? n - 1 : 1), rocprim::plus<unsigned>(), st));
    }
    const size_t per_tri = 6 * sizeof(float4) + 2 * sizeof(unsigned long long) + 9 * sizeof(unsigned) + 4 * sizeof(int); // (+ k_node_boxes_chunk's spine lists)
    const size_t levels = 4 * sizeof(float4) * ((size_t)n / 31 + 64 * GVT_BOX_LEVELS);
    A.cap = per_tri * ((size_t)n + 1) + levels + tb_sort + tb_scan + 64 * 256 + 4096;
    A.base = (char *)scratch_get(21, A.cap);
    if (!A.base) { rc = GVT_HIP_ERR_DEVICE; goto done; }
  }
  mark("arena");
  OK(A.take(&plo, n32)); OK(A.take(&phi, n32)); OK(A.take(&d_scene, 8));
  k_tri_bounds32<<<G, B, 0, st>>>(M->d_verts, M->d_tris, n, plo, phi);
  { // scene box: 32:1 reductions (the first one straight from the vertices), the last <= 32 boxes by one wave; the record stays on the device (k_scene_box)
    BoxLevels T;
    OK(build_box_levels(plo, phi, n32, A, T, st));
    unsigned cnt = n32;
    for (int l = 1; l < T.n_levels; l++) cnt = (cnt + 31u) / 32u;
    k_scene_box<<<1, 64, 0, st>>>(T.lo[T.n_levels - 1], T.hi[T.n_levels - 1], cnt, d_scene);
    HOK(hipMemcpyAsync(h_scene, d_scene, 8 * sizeof(float), hipMemcpyDeviceToHost, st)); // complete at the build's next synchronisation (the node count's, or below)
  }
#define SCENE_TO_HOST() do { for (int k_ = 0; k_ < 3; k_++) { M->lo[k_] = h_scene[k_]; M->hi[k_] = h_scene[3 + k_]; } pad = h_scene[7]; } while (0)
  mark("scene box");
  OK(dalloc(&M->d_tri, (size_t)4 * n));
  OK(dalloc(&M->d_slot_of, n));
  if (want_q) { OK(A.take(&leaf_of, n)); OK(dalloc(&M->d_triq, (size_t)4 * n)); }

  if ((int)n <= leaf_max) {
    OK(dalloc(&M->d_nodes, 1));
    M->nNodes = 1; M->nLeaves = 1;
    OK(A.take(&sorted, n));
    {
      std::vector<unsigned> id(n);
      for (unsigned i = 0; i < n; i++) id[i] = i;
      HOK(hipMemcpyAsync(sorted, id.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice, st));
      HOK(hipStreamSynchronize(st));
      SCENE_TO_HOST();
    }
    float4 *tlo = nullptr, *thi = nullptr; // n <= 4 triangle boxes
    OK(A.take(&tlo, n)); OK(A.take(&thi, n));
    k_tri_bounds<<<G, B, 0, st>>>(M->d_verts, M->d_tris, n, tlo, thi);
    k_single_node<<<1, 64, 0, st>>>(tlo, thi, n, pad, M->d_nodes);
    k_emit_tris<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, n, M->d_tri, M->d_slot_of, nullptr, nullptr);
    if (want_q) {
      std::vector<unsigned> lo_(n, (unsigned)~leaf_ref(0u, n));
      HOK(hipMemcpyAsync(leaf_of, lo_.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice, st));
      HOK(hipStreamSynchronize(st));
      k_emit_trisq<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, leaf_of, n, M->d_triq);
    }
  } else {
    OK(A.take(&keys, n)); OK(A.take(&keys2, n)); OK(A.take(&vals, n)); OK(A.take(&sorted, n));
    k_morton<<<G, B, 0, st>>>(M->d_verts, M->d_tris, n, d_scene, key_bits, keys, vals);
    {
      OK(A.take((char **)&tmp, tb_sort));
      HOK(rocprim::radix_sort_pairs(tmp, tb_sort, keys, keys2, vals, sorted, n, 0, key_bits, st));
    }
    mark("morton + sort");
    OK(A.take(&cl, n)); OK(A.take(&cr, n)); OK(A.take(&rf, n)); OK(A.take(&rl, n));
    OK(A.take(&slo, n)); OK(A.take(&shi, n)); OK(A.take(&ilo, n)); OK(A.take(&ihi, n));
    OK(A.take(&live, n)); OK(A.take(&newidx, n + 1));
    k_karras<<<(n_inner + B - 1) / B, B, 0, st>>>(keys2, (int)n, cl, cr, rf, rl);
    k_emit_tris<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, n, M->d_tri, M->d_slot_of, slo, shi); // the slots, and the sorted boxes on the way
    {
      BoxLevels T;
      OK(build_box_levels(slo, shi, n, A, T, st));
      if (n >= 4 * NB_CHUNK && !getenv("GVT_HIP_NODE_BOXES_TABLE")) {
        const unsigned n_chunks = (n + NB_CHUNK - 1) / NB_CHUNK;
        int *spine = nullptr;
        unsigned *spine_n = nullptr;
        OK(A.take(&spine, (size_t)n_chunks * NB_CHUNK)); OK(A.take(&spine_n, n_chunks));
        k_node_boxes_chunk<<<n_chunks, NB_THREADS, 0, st>>>((int)n, cl, cr, rf, rl, slo, shi, ilo, ihi, spine, spine_n); // nodes inside one chunk of sorted triangles: in LDS
        k_node_boxes_spine<<<n_chunks, 64, 0, st>>>(spine, spine_n, rf, rl, T, ilo, ihi);                          // the spines across chunk boundaries: the range-union table
      } else k_node_boxes<<<(n_inner + B - 1) / B, B, 0, st>>>(n_inner, rf, rl, T, ilo, ihi);
    }
    k_mark_live<<<(n_inner + B - 1) / B, B, 0, st>>>(n_inner, rf, rl, live, leaf_max);
    {
      OK(A.take((char **)&tmp2, tb_scan));
      HOK(rocprim::exclusive_scan(tmp2, tb_scan, live, newidx, 0u, (size_t)n_inner, rocprim::plus<unsigned>(), st));
    }
    mark("karras + boxes + scan");
    HOK(hipMemcpyAsync(C.h_pinned + 40, newidx + (n_inner - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HOK(hipMemcpyAsync(C.h_pinned + 41, live + (n_inner - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HOK(hipStreamSynchronize(st));
    SCENE_TO_HOST();
    M->nNodes = (size_t)C.h_pinned[40] + C.h_pinned[41];
    OK(dalloc(&M->d_nodes, M->nNodes));
    M->nLeaves = M->nNodes + 1;
    k_emit_nodes<<<(n_inner + B - 1) / B, B, 0, st>>>(n_inner, live, newidx, slo, shi, ilo, ihi, cl, cr, rf, rl, d_scene, M->d_nodes, leaf_max, leaf_of);
    if (want_q) k_emit_trisq<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, leaf_of, n, M->d_triq);
    mark("emit nodes + slots");
  }
  if (gctx().wide4 || want_q) OK(build_nodes4(M, &A)); // the traversal layout; counted in the build time
  mark("4-wide collapse");
  { // packet-friendly?  (a surface: a few dozen inner nodes on a random line, whatever the triangle count; a soup: N^(1/3) of them)
    const float ex = M->hi[0] - M->lo[0], ey = M->hi[1] - M->lo[1], ez = M->hi[2] - M->lo[2];
    const float ra = ex * ey + ey * ez + ez * ex;
    M->sah_inner = 0.f; M->packet_ok = false;
    if (ra > 0.f && M->nNodes > 1) {
      OK(A.take(&sah_acc, 1));
      HOK(hipMemsetAsync(sah_acc, 0, sizeof(unsigned long long), st));
      k_sah_sum<<<(unsigned)std::min<size_t>(2048, (M->nNodes + 255) / 256), 256, 0, st>>>(M->d_nodes, (unsigned)M->nNodes, 1.f / ra, sah_acc);
      HOK(hipMemcpyAsync(&sah_host, sah_acc, sizeof sah_host, hipMemcpyDeviceToHost, st));
    }
  }
  HOK(hipEventRecord(e1, st));
  HOK(hipEventSynchronize(e1));
  HOK(hipEventElapsedTime(&M->build_ms, e0, e1));
  gctx().stats.ms_build += M->build_ms;
  if (M->nNodes > 1) {
    M->sah_inner = 1.f + (float)((double)sah_host / 65536.0);
    M->packet_ok = M->sah_inner <= (float)C.packet_sah_max;
  }
done:
  hipStreamSynchronize(st);
  if (A.cap > ((size_t)1 << 30)) scratch_release(21); // a large scene's temporaries are not kept
  hipEventDestroy(e0); hipEventDestroy(e1);
  return rc;
#undef OK
#undef HOK
#undef SCENE_TO_HOST
}

// radix sort of (key, value) pairs on the adapter stream; temporary storage from the grow-only scratch arena
int sort_pairs_u32(unsigned *keys_in, unsigned *keys_out, unsigned *vals_in, unsigned *vals_out, size_t n, int end_bit) {
  Ctx &C = gctx();
  size_t tb = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, C.stream));
  void *tmp = scratch_get(12, tb ? tb : 16);
  if (!tmp) return GVT_HIP_ERR_DEVICE;
  HIPCHK(rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, C.stream));
  return 0;
}

// Compressed 4-wide collapse of the emitted binary tree (the traversal layout of k_trace's wide4 variant)
static int build_nodes4(gvt_hip_mesh *M, BuildArena *A) {
  if (M->d_nodes4 || !M->nNodes) return 0;
  // k_trace addresses a node by a 32-bit BYTE offset (index << 6, trace_lane.inc): 2^26 nodes is the layout's limit (a mesh of ~130 M triangles)
  if (M->nNodes >= ((size_t)1 << 26)) { set_error("mesh: %zu tree nodes exceed the traversal layout's 2^26 (cut the mesh into several instances)", M->nNodes); return GVT_HIP_ERR_CAPACITY; }
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  int *fa = nullptr, *fb = nullptr;
  unsigned *cnt = nullptr;
  int rc = dalloc(&M->d_nodes4, (size_t)GVT_NODE4_F4 * M->nNodes);
  if (!rc && M->d_triq) rc = dalloc(&M->d_nodes4q, (size_t)GVT_NODE4_F4 * M->nNodes);
  if (A) { // called from build_lbvh: the frontier arrays out of the build's arena
    if (!rc) rc = A->take(&fa, M->nNodes);
    if (!rc) rc = A->take(&fb, M->nNodes);
    if (!rc) rc = A->take(&cnt, GVT_COLLAPSE_LEVELS + 1);
  } else {
    if (!rc) rc = dalloc(&fa, M->nNodes);
    if (!rc) rc = dalloc(&fb, M->nNodes);
    if (!rc) rc = dalloc(&cnt, GVT_COLLAPSE_LEVELS + 1);
  }
  if (!rc) {
    // levels[l] = frontier size of level l (levels[0] = 1: the root); levels are launched in batches, the sizes read once per batch
    std::vector<unsigned> h_levels(GVT_COLLAPSE_LEVELS + 1, 0u);
    h_levels[0] = 1u;
    const int root = 0;
    hipError_t e = hipMemcpyAsync(fa, &root, sizeof root, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(cnt, h_levels.data(), sizeof(unsigned) * (GVT_COLLAPSE_LEVELS + 1), hipMemcpyHostToDevice, st);
    unsigned base = 0;
    int level = 0;
    bool more = true;
    while (e == hipSuccess && more) {
      if (level >= GVT_COLLAPSE_LEVELS) { set_error("4-wide collapse: more than %d levels", GVT_COLLAPSE_LEVELS); rc = GVT_HIP_ERR_CAPACITY; break; }
      // first batch: the depth a tree of this size has at least (log4 of its nodes) and a few levels more; then four at a time
      int first = 3;
      for (size_t k = M->nNodes; k > 1; k >>= 2) first++;
      const int batch_end = std::min(GVT_COLLAPSE_LEVELS, level == 0 ? first : level + 4);
      const int batch_begin = level;
      for (; level < batch_end; level++) {
        // a level holds at most 4^level nodes (a node has at most four children) and at most every node
        const size_t bound = level >= 16 ? M->nNodes : std::min<size_t>(M->nNodes, (size_t)1 << (2 * level));
        k_collapse4<<<(unsigned)((bound + GVT_COLLAPSE_BLOCK - 1) / GVT_COLLAPSE_BLOCK), GVT_COLLAPSE_BLOCK, 0, st>>>(M->d_nodes, fa, cnt, level, fb, M->d_nodes4, M->d_nodes4q);
        int *t = fa; fa = fb; fb = t;
      }
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(h_levels.data(), cnt, sizeof(unsigned) * (GVT_COLLAPSE_LEVELS + 1), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      for (int l = batch_begin; l < batch_end; l++) base += h_levels[l];
      more = h_levels[batch_end] != 0u;
    }
    M->nNodes4 = base;
    M->levels4.clear();
    for (int l = 0; l < GVT_COLLAPSE_LEVELS && h_levels[l]; l++) M->levels4.push_back(h_levels[l]);
    if (e != hipSuccess) { set_error("4-wide collapse: %s", hipGetErrorString(e)); rc = GVT_HIP_ERR_DEVICE; }
    if (getenv("GVT_HIP_BUILD_TRACE")) {
      int depth = 0;
      while (depth < GVT_COLLAPSE_LEVELS && h_levels[depth]) depth++;
      fprintf(stderr, "[build] 4-wide collapse: %d levels (%zu nodes), %d launched\n", depth, (size_t)base, level);
    }
  }
  if (!A) { hipFree(fa); hipFree(fb); hipFree(cnt); }
  if (rc) { hipFree(M->d_nodes4); M->d_nodes4 = nullptr; hipFree(M->d_nodes4q); M->d_nodes4q = nullptr; }
  return rc;
}
int build_nodes4(gvt_hip_mesh *M) { return build_nodes4(M, nullptr); }

// ------------------------------------------------------------------------------------------------
// The CLUSTER layout of the 4-wide nodes, for the traversals that give a ray a whole wave (k_finish: trace_wave.inc wave_closest_run_c / wave_any_run_c).
// Such a traversal is a chain of dependent node fetches, one per level of the tree (the wave opens its ray's whole frontier per step): ~12 steps of a microsecond
// each for a tile of a million triangles.  Here every node of an EVEN level is followed in memory by its inner children (odd level), and a reference to an even
// node carries, in its low four bits, which of its children are inner nodes: (slot << 4) | mask.  A step then fetches a node AND its children at once -- the
// children's addresses follow from the reference alone -- and descends two levels per memory round trip.  The array is a permutation of nodes4 (every node once, the
// same 64 bytes but for the references of the odd-level nodes); leaf references are unchanged.
namespace {
struct Levels4 { unsigned off[GVT_COLLAPSE_LEVELS + 2]; int n; }; // off[L] = index of level L's first node in the breadth-first array, off[n] = all nodes
__device__ inline int level_of4(const Levels4 &T, unsigned i) {
  int lo = 0, hi = T.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.off[mid] <= i) lo = mid; else hi = mid - 1; }
  return lo;
}
__device__ inline unsigned inner_mask4(const uint4 *__restrict__ nd) { // bit c: child c is an inner node
  const uint4 w2 = nd[2], w3 = nd[3];
  return ((int)w2.z >= 0 ? 1u : 0u) | ((int)w2.w >= 0 ? 2u : 0u) | ((int)w3.x >= 0 ? 4u : 0u) | ((int)w3.y >= 0 ? 8u : 0u);
}
// slots an even-level node's cluster takes (itself + its inner children); 0 for an odd-level node (it lives in its parent's cluster)
__global__ __launch_bounds__(256) void k_cluster_weight(const uint4 *__restrict__ nodes4, unsigned n4, Levels4 T, unsigned *__restrict__ w) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  w[i] = (level_of4(T, i) & 1) ? 0u : 1u + (unsigned)__popc(inner_mask4(nodes4 + (size_t)GVT_NODE4_F4 * i));
}
__global__ __launch_bounds__(256) void k_cluster_emit(const uint4 *__restrict__ nodes4, unsigned n4, Levels4 T, const unsigned *__restrict__ pos, uint4 *__restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4 || (level_of4(T, i) & 1)) return;
  const uint4 *nd = nodes4 + (size_t)GVT_NODE4_F4 * i;
  uint4 *head = out + (size_t)GVT_NODE4_F4 * pos[i];
  const uint4 h2 = nd[2], h3 = nd[3];
  head[0] = nd[0]; head[1] = nd[1]; head[2] = h2; head[3] = h3; // (its references to inner children are not followed: the children sit behind it)
  const int refs[4] = { (int)h2.z, (int)h2.w, (int)h3.x, (int)h3.y };
  unsigned slot = pos[i] + 1u;
  for (int c = 0; c < 4; c++) {
    if (refs[c] < 0) continue;
    const uint4 *ch = nodes4 + (size_t)GVT_NODE4_F4 * (unsigned)refs[c];
    uint4 c2 = ch[2], c3 = ch[3];
    int r[4] = { (int)c2.z, (int)c2.w, (int)c3.x, (int)c3.y };
    for (int k = 0; k < 4; k++)
      if (r[k] >= 0) r[k] = (int)((pos[(unsigned)r[k]] << 4) | inner_mask4(nodes4 + (size_t)GVT_NODE4_F4 * (unsigned)r[k])); // a grandchild: an even-level node again
    c2.z = (unsigned)r[0]; c2.w = (unsigned)r[1]; c3.x = (unsigned)r[2]; c3.y = (unsigned)r[3];
    uint4 *dst = out + (size_t)GVT_NODE4_F4 * slot;
    dst[0] = ch[0]; dst[1] = ch[1]; dst[2] = c2; dst[3] = c3;
    slot++;
  }
}
} // namespace

int build_nodes4c(gvt_hip_mesh *M) {
  static std::mutex once; // (tracers of several contexts -- threads -- may be created over one mesh at the same time)
  std::lock_guard<std::mutex> lock(once);
  if (M->d_nodes4c || !M->d_nodes4 || !M->nNodes4 || M->levels4.empty()) return 0;
  if (M->levels4.size() > (size_t)GVT_COLLAPSE_LEVELS || M->nNodes4 >= ((size_t)1 << 26)) return 0; // (the wave-per-ray traversals walk nodes4 then)
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  const unsigned n4 = (unsigned)M->nNodes4;
  Levels4 T;
  std::memset(&T, 0, sizeof T);
  T.n = (int)M->levels4.size();
  unsigned run = 0;
  for (int l = 0; l < T.n; l++) { T.off[l] = run; run += M->levels4[l]; }
  T.off[T.n] = run;
  if (run != n4) { set_error("cluster layout: the level sizes add up to %u of %u nodes", run, n4); return GVT_HIP_ERR_DEVICE; }
  unsigned *w = nullptr, *pos = nullptr;
  void *tmp = nullptr;
  size_t tb = 0;
  int rc = dalloc(&w, n4);
  if (!rc) rc = dalloc(&pos, n4);
  if (!rc) rc = dalloc(&M->d_nodes4c, (size_t)GVT_NODE4_F4 * n4);
  if (!rc && rocprim::exclusive_scan(nullptr, tb, w, pos, 0u, (size_t)n4, rocprim::plus<unsigned>(), st) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (!rc && hipMalloc(&tmp, tb ? tb : 16) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  uint4 root[4];
  if (!rc) {
    k_cluster_weight<<<(n4 + 255) / 256, 256, 0, st>>>(M->d_nodes4, n4, T, w);
    hipError_t e = rocprim::exclusive_scan(tmp, tb, w, pos, 0u, (size_t)n4, rocprim::plus<unsigned>(), st);
    if (e == hipSuccess) { k_cluster_emit<<<(n4 + 255) / 256, 256, 0, st>>>(M->d_nodes4, n4, T, pos, M->d_nodes4c); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(root, M->d_nodes4, 64, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("cluster layout: %s", hipGetErrorString(e)); rc = GVT_HIP_ERR_DEVICE; }
  }
  if (!rc) {
    const int refs[4] = { (int)root[2].z, (int)root[2].w, (int)root[3].x, (int)root[3].y };
    unsigned mask = 0;
    for (int c = 0; c < 4; c++) if (refs[c] >= 0) mask |= 1u << c;
    M->root_entry4c = (int)mask; // (the root's cluster starts at slot 0)
  }
  hipFree(w); hipFree(pos); hipFree(tmp);
  if (rc) { hipFree(M->d_nodes4c); M->d_nodes4c = nullptr; if (!*gvt_hip_last_error()) set_error("cluster layout: device allocation failed"); }
  return rc;
}

// diagnostic: marks[k] = 1 where binary node k is the root of a `width`-wide node (width 2..8); *n_wide = number of wide nodes
int wide_root_marks(gvt_hip_mesh *M, int width, unsigned char *d_marks, size_t *n_wide) {
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  int *fa = nullptr, *fb = nullptr;
  unsigned *cnt = nullptr;
  int rc = dalloc(&fa, M->nNodes);
  if (!rc) rc = dalloc(&fb, M->nNodes);
  if (!rc) rc = dalloc(&cnt, 1);
  size_t total = 0;
  if (!rc) {
    const int root = 0;
    hipError_t e = hipMemsetAsync(d_marks, 0, M->nNodes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(fa, &root, sizeof root, hipMemcpyHostToDevice, st);
    unsigned n_in = 1;
    while (e == hipSuccess && n_in) {
      e = hipMemsetAsync(cnt, 0, sizeof(unsigned), st);
      if (e != hipSuccess) break;
      k_collapse_mark<<<(n_in + 255) / 256, 256, 0, st>>>(M->d_nodes, fa, n_in, width, fb, cnt, d_marks);
      unsigned n_next = 0;
      e = hipMemcpyAsync(&n_next, cnt, sizeof n_next, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      total += n_in;
      n_in = n_next;
      int *t = fa; fa = fb; fb = t;
    }
    if (e != hipSuccess) { set_error("wide_root_marks: %s", hipGetErrorString(e)); rc = GVT_HIP_ERR_DEVICE; }
  }
  hipFree(fa); hipFree(fb); hipFree(cnt);
  if (n_wide) *n_wide = total;
  return rc;
}
