// lbvh.hip -- acceleration-structure build on the device (replaces rtcCommit of
// EmbreeMeshAdapter::EmbreeMeshAdapter, EmbreeMeshAdapter.cpp:125-162).
//
// Pipeline (all on the adapter stream, one pass over HBM each):
//   tri bounds + scene box  ->  63-bit Morton keys  ->  radix sort (rocPRIM)  ->  Karras 2012 topology
//   ->  bottom-up box fit (agent-scope release/acquire per level, the per-XCD L2s are not coherent)
//   ->  collapse subtrees of <= 4 triangles into leaves, compact live nodes (prefix scan)
//   ->  emit 64-byte nodes (both child boxes in the parent) + 64-byte triangle slots in leaf order.
// Results of the closest/any-hit queries do not depend on the tree (conservative, padded boxes); only
// speed does.
#include <string.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gvt_internal.h"

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
                                                    float4 *__restrict__ plo, float4 *__restrict__ phi, unsigned *__restrict__ box /*6 ordered*/) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  float lo[3] = { GVT_FLT_MAX, GVT_FLT_MAX, GVT_FLT_MAX }, hi[3] = { -GVT_FLT_MAX, -GVT_FLT_MAX, -GVT_FLT_MAX };
  if (i < n) {
    int a = tris[3 * i], b = tris[3 * i + 1], c = tris[3 * i + 2];
    for (int k = 0; k < 3; k++) {
      float x = verts[3 * a + k], y = verts[3 * b + k], z = verts[3 * c + k];
      lo[k] = fminf(x, fminf(y, z));
      hi[k] = fmaxf(x, fmaxf(y, z));
    }
    plo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    phi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
  // wave reduction, then one atomic per wave and bound
  for (int k = 0; k < 3; k++) {
    float l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = fminf(l, __shfl_xor(l, off));
      h = fmaxf(h, __shfl_xor(h, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&box[k], f2ord(l));
      atomicMax(&box[3 + k], f2ord(h));
    }
  }
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

__global__ __launch_bounds__(256) void k_morton(const float4 *__restrict__ plo, const float4 *__restrict__ phi, unsigned n, float3 blo,
                                                float3 inv_ext, unsigned long long *__restrict__ keys, unsigned *__restrict__ vals) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 l = plo[i], h = phi[i];
  float cx = (0.5f * (l.x + h.x) - blo.x) * inv_ext.x;
  float cy = (0.5f * (l.y + h.y) - blo.y) * inv_ext.y;
  float cz = (0.5f * (l.z + h.z) - blo.z) * inv_ext.z;
  unsigned qx = (unsigned)fminf(fmaxf(cx * 2097152.f, 0.f), 2097151.f);
  unsigned qy = (unsigned)fminf(fmaxf(cy * 2097152.f, 0.f), 2097151.f);
  unsigned qz = (unsigned)fminf(fmaxf(cz * 2097152.f, 0.f), 2097151.f);
  keys[i] = (expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz);
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
                                                int *__restrict__ child_r, int *__restrict__ parent_inner, int *__restrict__ parent_leaf,
                                                int *__restrict__ rfirst, int *__restrict__ rlast) {
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
  if (first == gamma) { cl = ~gamma; parent_leaf[gamma] = i; } else { cl = gamma; parent_inner[gamma] = i; }
  if (last == gamma + 1) { cr = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { cr = gamma + 1; parent_inner[gamma + 1] = i; }
  child_l[i] = cl; child_r[i] = cr;
  rfirst[i] = first; rlast[i] = last;
  if (i == 0) parent_inner[0] = -1;
}

__device__ inline void child_box(int c, const unsigned *__restrict__ sorted, const float4 *__restrict__ plo, const float4 *__restrict__ phi,
                                 const float4 *ilo, const float4 *ihi, float4 &lo, float4 &hi) {
  if (c < 0) { unsigned p = sorted[~c]; lo = plo[p]; hi = phi[p]; }
  else { lo = ilo[c]; hi = ihi[c]; }
}

// bottom-up fit: the second thread to arrive at a node computes its box.  Inter-workgroup visibility on
// gfx950 needs agent-scope release before the arrival and acquire after it (MI355X guide, G16).
__global__ __launch_bounds__(256) void k_refit(int n, const unsigned *__restrict__ sorted, const float4 *__restrict__ plo,
                                               const float4 *__restrict__ phi, const int *__restrict__ child_l, const int *__restrict__ child_r,
                                               const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf, float4 *ilo,
                                               float4 *ihi, unsigned *flags) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int cur = parent_leaf[p];
  while (cur >= 0) {
    __threadfence();
    unsigned old = atomicAdd(&flags[cur], 1u);
    if (old == 0u) return;
    __threadfence();
    float4 al, ah, bl, bh;
    child_box(child_l[cur], sorted, plo, phi, ilo, ihi, al, ah);
    child_box(child_r[cur], sorted, plo, phi, ilo, ihi, bl, bh);
    ilo[cur] = make_float4(fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), 0.f);
    ihi[cur] = make_float4(fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), 0.f);
    cur = parent_inner[cur];
  }
}

__global__ __launch_bounds__(256) void k_mark_live(int n_inner, const int *__restrict__ rfirst, const int *__restrict__ rlast,
                                                   unsigned *__restrict__ live) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  live[i] = (rlast[i] - rfirst[i] + 1 > GVT_LEAF_MAX) ? 1u : 0u;
}

__device__ inline int final_ref(int c, const int *__restrict__ rfirst, const int *__restrict__ rlast, const unsigned *__restrict__ newidx) {
  if (c < 0) return leaf_ref((unsigned)(~c), 1u);
  int cnt = rlast[c] - rfirst[c] + 1;
  if (cnt <= GVT_LEAF_MAX) return leaf_ref((unsigned)rfirst[c], (unsigned)cnt);
  return (int)newidx[c];
}

__global__ __launch_bounds__(256) void k_emit_nodes(int n_inner, const unsigned *__restrict__ live, const unsigned *__restrict__ newidx,
                                                    const unsigned *__restrict__ sorted, const float4 *__restrict__ plo,
                                                    const float4 *__restrict__ phi, const float4 *__restrict__ ilo, const float4 *__restrict__ ihi,
                                                    const int *__restrict__ child_l, const int *__restrict__ child_r, const int *__restrict__ rfirst,
                                                    const int *__restrict__ rlast, float pad, BvhNode *__restrict__ nodes, unsigned *n_leaves) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner || !live[i]) return;
  int cl = child_l[i], cr = child_r[i];
  float4 al, ah, bl, bh;
  child_box(cl, sorted, plo, phi, ilo, ihi, al, ah);
  child_box(cr, sorted, plo, phi, ilo, ihi, bl, bh);
  BvhNode nd;
  nd.n0 = make_float4(al.x - pad, ah.x + pad, al.y - pad, ah.y + pad);
  nd.n1 = make_float4(bl.x - pad, bh.x + pad, bl.y - pad, bh.y + pad);
  nd.n2 = make_float4(al.z - pad, ah.z + pad, bl.z - pad, bh.z + pad);
  int r0 = final_ref(cl, rfirst, rlast, newidx), r1 = final_ref(cr, rfirst, rlast, newidx);
  nd.n3 = make_float4(__int_as_float(r0), __int_as_float(r1), 0.f, 0.f);
  nodes[newidx[i]] = nd;
  unsigned nl = (r0 < 0) + (r1 < 0);
  if (nl) atomicAdd(n_leaves, nl);
}

__global__ __launch_bounds__(256) void k_emit_tris(const float *__restrict__ verts, const int *__restrict__ tris, const unsigned *__restrict__ sorted,
                                                   unsigned n, float4 *__restrict__ out) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  unsigned p = sorted[s];
  int a = tris[3 * p], b = tris[3 * p + 1], c = tris[3 * p + 2];
  V3 v0 = ld3(verts + 3 * a), v1 = ld3(verts + 3 * b), v2 = ld3(verts + 3 * c);
  V3 e1 = sub3(v0, v1), e2 = sub3(v2, v0);
  V3 Ng = cross3(e1, e2); // same float ops as evaluating it per test (no contraction): bit-identical
  out[4 * s + 0] = make_float4(v0.x, v0.y, v0.z, __int_as_float((int)p));
  out[4 * s + 1] = make_float4(e1.x, e1.y, e1.z, 0.f);
  out[4 * s + 2] = make_float4(e2.x, e2.y, e2.z, 0.f);
  out[4 * s + 3] = make_float4(Ng.x, Ng.y, Ng.z, 0.f);
}

// n <= GVT_LEAF_MAX: one node, child0 = the only leaf, child1 = empty leaf behind an inverted box
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

// 4-wide collapse, breadth first: every thread turns one binary node (and its two children) into one 4-wide node,
// allocates indices for the grandchildren that are inner nodes and queues them for the next level.
struct Frontier { int b; int idx4; };
__device__ inline void put_child(float4 *w, int k, float4 lo_hi_xy /*lo.x,hi.x,lo.y,hi.y*/, float lz, float hz, int ref) {
  ((float *)&w[0])[k] = lo_hi_xy.x; ((float *)&w[1])[k] = lo_hi_xy.y; ((float *)&w[2])[k] = lo_hi_xy.z; ((float *)&w[3])[k] = lo_hi_xy.w;
  ((float *)&w[4])[k] = lz; ((float *)&w[5])[k] = hz; ((int *)&w[6])[k] = ref;
}
__global__ __launch_bounds__(256) void k_collapse4(const BvhNode *__restrict__ nodes, const Frontier *__restrict__ fin, unsigned n_in,
                                                   Frontier *__restrict__ fout, unsigned *__restrict__ counters /*0: nodes4, 1: next frontier*/,
                                                   float4 *__restrict__ nodes4) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const Frontier f = fin[i];
  const BvhNode nb = nodes[f.b];
  float4 w[GVT_NODE4_F4];
  for (int k = 0; k < GVT_NODE4_F4; k++) w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < 4; k++) put_child(w, k, make_float4(GVT_FLT_MAX, -GVT_FLT_MAX, GVT_FLT_MAX, -GVT_FLT_MAX), GVT_FLT_MAX, -GVT_FLT_MAX, leaf_ref(0u, 0u));
  int n = 0;
  for (int side = 0; side < 2; side++) {
    const int ref = __float_as_int(side ? nb.n3.y : nb.n3.x);
    if (ref >= 0) { // inner binary child: adopt its two children
      const BvhNode nc = nodes[ref];
      for (int s2 = 0; s2 < 2; s2++) {
        int r2 = __float_as_int(s2 ? nc.n3.y : nc.n3.x);
        if (r2 >= 0) {
          const int idx4 = (int)atomicAdd(&counters[0], 1u);
          const unsigned slot = atomicAdd(&counters[1], 1u);
          Frontier g; g.b = r2; g.idx4 = idx4;
          fout[slot] = g;
          r2 = idx4;
        }
        put_child(w, n++, s2 ? nc.n1 : nc.n0, s2 ? nc.n2.z : nc.n2.x, s2 ? nc.n2.w : nc.n2.y, r2);
      }
    } else {
      put_child(w, n++, side ? nb.n1 : nb.n0, side ? nb.n2.z : nb.n2.x, side ? nb.n2.w : nb.n2.y, ref);
    }
  }
  float4 *dst = nodes4 + (size_t)GVT_NODE4_F4 * f.idx4;
  for (int k = 0; k < GVT_NODE4_F4; k++) dst[k] = w[k];
}

template <typename T> int dalloc(T **p, size_t n) {
  hipError_t e = hipMalloc((void **)p, sizeof(T) * (n ? n : 1));
  if (e != hipSuccess) { set_error("hipMalloc(%zu B) failed: %s", sizeof(T) * n, hipGetErrorString(e)); return GVT_HIP_ERR_DEVICE; }
  return 0;
}

} // namespace

int build_lbvh(gvt_hip_mesh *M) {
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  const unsigned n = (unsigned)M->nT;
  if (n == 0) { M->nNodes = 0; M->d_nodes = nullptr; M->d_tri = nullptr; return 0; }

  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, st));

  float4 *plo = nullptr, *phi = nullptr, *ilo = nullptr, *ihi = nullptr;
  unsigned long long *keys = nullptr, *keys2 = nullptr;
  unsigned *vals = nullptr, *sorted = nullptr, *box = nullptr, *flags = nullptr, *live = nullptr, *newidx = nullptr, *nleaves = nullptr;
  int *cl = nullptr, *cr = nullptr, *pin = nullptr, *pleaf = nullptr, *rf = nullptr, *rl = nullptr;
  void *tmp = nullptr;
  int rc = 0;
  const unsigned B = 256, G = (n + B - 1) / B;
  const int n_inner = (int)n - 1;
  unsigned hbox[6];
  float pad = 0.f;

#define OK(x) do { if ((rc = (x)) != 0) goto done; } while (0)
#define HOK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(_e)); rc = GVT_HIP_ERR_DEVICE; goto done; } } while (0)

  OK(dalloc(&plo, n)); OK(dalloc(&phi, n)); OK(dalloc(&box, 8));
  {
    unsigned init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    HOK(hipMemcpyAsync(box, init, sizeof init, hipMemcpyHostToDevice, st));
  }
  k_tri_bounds<<<G, B, 0, st>>>(M->d_verts, M->d_tris, n, plo, phi, box);
  HOK(hipMemcpyAsync(hbox, box, sizeof hbox, hipMemcpyDeviceToHost, st));
  HOK(hipStreamSynchronize(st));
  for (int k = 0; k < 3; k++) { M->lo[k] = ord2f(hbox[k]); M->hi[k] = ord2f(hbox[3 + k]); }
  {
    float ext = 0.f;
    for (int k = 0; k < 3; k++) { ext = fmaxf(ext, fabsf(M->lo[k])); ext = fmaxf(ext, fabsf(M->hi[k])); }
    pad = ext * 1e-5f; // keeps the slab test conservative w.r.t. the triangle test's rounding
  }
  OK(dalloc(&M->d_tri, (size_t)4 * n));

  if (n <= GVT_LEAF_MAX) {
    OK(dalloc(&M->d_nodes, 1));
    M->nNodes = 1; M->nLeaves = 1;
    OK(dalloc(&sorted, n));
    {
      std::vector<unsigned> id(n);
      for (unsigned i = 0; i < n; i++) id[i] = i;
      HOK(hipMemcpyAsync(sorted, id.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice, st));
      HOK(hipStreamSynchronize(st));
    }
    k_single_node<<<1, 64, 0, st>>>(plo, phi, n, pad, M->d_nodes);
    k_emit_tris<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, n, M->d_tri);
  } else {
    OK(dalloc(&keys, n)); OK(dalloc(&keys2, n)); OK(dalloc(&vals, n)); OK(dalloc(&sorted, n));
    {
      float3 blo = make_float3(M->lo[0], M->lo[1], M->lo[2]);
      float ex = M->hi[0] - M->lo[0], ey = M->hi[1] - M->lo[1], ez = M->hi[2] - M->lo[2];
      float3 inv = make_float3(ex > 0 ? 1.f / ex : 0.f, ey > 0 ? 1.f / ey : 0.f, ez > 0 ? 1.f / ez : 0.f);
      k_morton<<<G, B, 0, st>>>(plo, phi, n, blo, inv, keys, vals);
    }
    {
      size_t tb = 0;
      HOK(rocprim::radix_sort_pairs(nullptr, tb, keys, keys2, vals, sorted, n, 0, 63, st));
      HOK(hipMalloc(&tmp, tb ? tb : 1));
      HOK(rocprim::radix_sort_pairs(tmp, tb, keys, keys2, vals, sorted, n, 0, 63, st));
    }
    OK(dalloc(&cl, n)); OK(dalloc(&cr, n)); OK(dalloc(&pin, n)); OK(dalloc(&pleaf, n)); OK(dalloc(&rf, n)); OK(dalloc(&rl, n));
    OK(dalloc(&ilo, n)); OK(dalloc(&ihi, n)); OK(dalloc(&flags, n)); OK(dalloc(&live, n)); OK(dalloc(&newidx, n + 1)); OK(dalloc(&nleaves, 1));
    HOK(hipMemsetAsync(flags, 0, sizeof(unsigned) * n, st));
    HOK(hipMemsetAsync(nleaves, 0, sizeof(unsigned), st));
    k_karras<<<(n_inner + B - 1) / B, B, 0, st>>>(keys2, (int)n, cl, cr, pin, pleaf, rf, rl);
    k_refit<<<G, B, 0, st>>>((int)n, sorted, plo, phi, cl, cr, pin, pleaf, ilo, ihi, flags);
    k_mark_live<<<(n_inner + B - 1) / B, B, 0, st>>>(n_inner, rf, rl, live);
    {
      size_t tb = 0;
      HOK(hipFree(tmp)); tmp = nullptr;
      HOK(rocprim::exclusive_scan(nullptr, tb, live, newidx, 0u, (size_t)n_inner, rocprim::plus<unsigned>(), st));
      HOK(hipMalloc(&tmp, tb ? tb : 1));
      HOK(rocprim::exclusive_scan(tmp, tb, live, newidx, 0u, (size_t)n_inner, rocprim::plus<unsigned>(), st));
    }
    unsigned last_idx = 0, last_live = 0;
    HOK(hipMemcpyAsync(&last_idx, newidx + (n_inner - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HOK(hipMemcpyAsync(&last_live, live + (n_inner - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HOK(hipStreamSynchronize(st));
    M->nNodes = (size_t)last_idx + last_live;
    OK(dalloc(&M->d_nodes, M->nNodes));
    k_emit_nodes<<<(n_inner + B - 1) / B, B, 0, st>>>(n_inner, live, newidx, sorted, plo, phi, ilo, ihi, cl, cr, rf, rl, pad, M->d_nodes, nleaves);
    k_emit_tris<<<G, B, 0, st>>>(M->d_verts, M->d_tris, sorted, n, M->d_tri);
    unsigned nl = 0;
    HOK(hipMemcpyAsync(&nl, nleaves, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HOK(hipStreamSynchronize(st));
    M->nLeaves = nl;
  }
  HOK(hipEventRecord(e1, st));
  HOK(hipEventSynchronize(e1));
  HOK(hipEventElapsedTime(&M->build_ms, e0, e1));
  gctx().stats.ms_build += M->build_ms;
done:
  hipStreamSynchronize(st);
  hipFree(plo); hipFree(phi); hipFree(ilo); hipFree(ihi); hipFree(keys); hipFree(keys2); hipFree(vals); hipFree(sorted);
  hipFree(box); hipFree(flags); hipFree(live); hipFree(newidx); hipFree(nleaves); hipFree(cl); hipFree(cr); hipFree(pin);
  hipFree(pleaf); hipFree(rf); hipFree(rl); hipFree(tmp);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return rc;
#undef OK
#undef HOK
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

// 4-wide collapse of the emitted binary tree (optional traversal layout, built on first use)
int build_nodes4(gvt_hip_mesh *M) {
  if (M->d_nodes4 || !M->nNodes) return 0;
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  Frontier *fa = nullptr, *fb = nullptr;
  unsigned *cnt = nullptr;
  int rc = dalloc(&M->d_nodes4, (size_t)GVT_NODE4_F4 * M->nNodes);
  if (!rc) rc = dalloc(&fa, M->nNodes);
  if (!rc) rc = dalloc(&fb, M->nNodes);
  if (!rc) rc = dalloc(&cnt, 2);
  if (!rc) {
    Frontier root; root.b = 0; root.idx4 = 0;
    unsigned h[2] = { 1u, 0u };
    hipError_t e = hipMemcpyAsync(fa, &root, sizeof root, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(cnt, h, sizeof h, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    unsigned n_in = 1;
    while (e == hipSuccess && n_in) {
      k_collapse4<<<(n_in + 255) / 256, 256, 0, st>>>(M->d_nodes, fa, n_in, fb, cnt, M->d_nodes4);
      e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      n_in = h[1];
      unsigned zero = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(cnt + 1, &zero, sizeof zero, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      Frontier *t = fa; fa = fb; fb = t;
    }
    M->nNodes4 = h[0];
    if (e != hipSuccess) { set_error("4-wide collapse: %s", hipGetErrorString(e)); rc = GVT_HIP_ERR_DEVICE; }
  }
  hipFree(fa); hipFree(fb); hipFree(cnt);
  if (rc) { hipFree(M->d_nodes4); M->d_nodes4 = nullptr; }
  return rc;
}
