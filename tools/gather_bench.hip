// gather_bench.hip -- how should a wave fetch 64 different 64-byte BVH nodes (one per lane)?
//   A: every lane issues 4 x global_load_dwordx4 on its own node (what k_trace does)
//   B: the 4 lanes of a quad fetch one node together (lane k -> 16-byte piece k), one node per instruction, and the
//      quad transposes the pieces with DPP/shuffles so that lane j ends up with node j
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/gather_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Node { float4 a, b, c, d; };

__global__ __launch_bounds__(256) void k_own(const Node *__restrict__ nodes, const unsigned *__restrict__ idx, unsigned n_idx, int iters, float *out) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned cur = idx[tid % n_idx];
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    const Node *nd = nodes + cur;
    float4 a = nd->a, b = nd->b, c = nd->c, d = nd->d;
    acc += a.x + b.y + c.z + d.w;
    cur = __float_as_uint(d.x); // dependent chase: next node index stored in the node
  }
  out[tid] = acc;
}

__global__ __launch_bounds__(256) void k_quad(const Node *__restrict__ nodes, const unsigned *__restrict__ idx, unsigned n_idx, int iters, float *out) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned cur = idx[tid % n_idx];
  const int lane = threadIdx.x & 63, k = lane & 3, qbase = lane & ~3;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    // instruction j: the quad fetches node of its lane j; lane k takes piece k
    float4 p[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned nj = __shfl(cur, qbase + j);
      p[j] = ((const float4 *)(nodes + nj))[k];
    }
    // transpose inside the quad: lane j needs piece m of node j, which lane m holds in p[j]
    float4 mine[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      // value for lane j from lane m: p[j] (indexed by the RECEIVING lane's j): select locally per sending lane
      float4 send;
      // every lane sends to lane j its p[j]; do it as 4 rounds of shuffles with rotating partner
      (void)send;
    }
    // rounds: in round r lane l exchanges with lane l^r ... simple form: for each source m, dest j gets p_m[j]
#pragma unroll
    for (int m = 0; m < 4; m++) {
      float4 v;
      // lane j wants p[j] held by lane (qbase+m): lanes pick their own index j=k when reading from lane m
      float4 sel = (k == 0) ? p[0] : (k == 1) ? p[1] : (k == 2) ? p[2] : p[3]; // what I would send if asked for "my" j... (placeholder)
      (void)sel;
      v.x = __shfl(p[0].x, qbase + m); (void)v;
    }
    // straightforward (compiler lowers quad shuffles to DPP): piece m of my node = lane m's p[k]
#pragma unroll
    for (int m = 0; m < 4; m++) {
      float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
      // lane m must expose p[dest]; dest varies per receiving lane -> 4 shuffles per component, select by k
      float x0 = __shfl(q0.x, qbase + m), x1 = __shfl(q1.x, qbase + m), x2 = __shfl(q2.x, qbase + m), x3 = __shfl(q3.x, qbase + m);
      float y0 = __shfl(q0.y, qbase + m), y1 = __shfl(q1.y, qbase + m), y2 = __shfl(q2.y, qbase + m), y3 = __shfl(q3.y, qbase + m);
      float z0 = __shfl(q0.z, qbase + m), z1 = __shfl(q1.z, qbase + m), z2 = __shfl(q2.z, qbase + m), z3 = __shfl(q3.z, qbase + m);
      float w0 = __shfl(q0.w, qbase + m), w1 = __shfl(q1.w, qbase + m), w2 = __shfl(q2.w, qbase + m), w3 = __shfl(q3.w, qbase + m);
      mine[m].x = k == 0 ? x0 : k == 1 ? x1 : k == 2 ? x2 : x3;
      mine[m].y = k == 0 ? y0 : k == 1 ? y1 : k == 2 ? y2 : y3;
      mine[m].z = k == 0 ? z0 : k == 1 ? z1 : k == 2 ? z2 : z3;
      mine[m].w = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : w3;
    }
    acc += mine[0].x + mine[1].y + mine[2].z + mine[3].w;
    cur = __float_as_uint(mine[3].x);
  }
  out[tid] = acc;
}

// C: quad-cooperative fetch through LDS: lane k stores piece k of node j to LDS row j of its quad, then reads its row
__global__ __launch_bounds__(256) void k_quad_lds(const Node *__restrict__ nodes, const unsigned *__restrict__ idx, unsigned n_idx, int iters, float *out) {
  __shared__ float4 tile[256 * 4];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned cur = idx[tid % n_idx];
  const int lane = threadIdx.x & 63, k = lane & 3, qbase = lane & ~3;
  const int tq = (threadIdx.x & ~3); // first thread of my quad within the block
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned nj = __shfl(cur, qbase + j);
      tile[(tq + j) * 4 + k] = ((const float4 *)(nodes + nj))[k];
    }
    __builtin_amdgcn_wave_barrier();
    const float4 *m = &tile[threadIdx.x * 4];
    float4 a = m[0], b = m[1], c = m[2], d = m[3];
    __builtin_amdgcn_wave_barrier();
    acc += a.x + b.y + c.z + d.w;
    cur = __float_as_uint(d.x);
  }
  out[tid] = acc;
}

// D: quad-cooperative fetch with LDS-DIRECT loads (gfx950: global_load_lds_dwordx4 -- the 16-byte pieces go from the L1 straight into
// LDS, no VGPR staging, no cross-lane transpose): instruction j brings the node of the quad's lane j (its four lanes address the four
// pieces of ONE 64-byte line: one look-up in the L1's tag pipe instead of four), lane L's piece lands at M0 + L * 16; lane j then reads its
// node as four ds_read_b128 from tile j.  Tiles are 1024 + 64 bytes apart (a quad's four reads hit different banks).
#define TILE_STRIDE (1024 + 64)
__global__ __launch_bounds__(256) void k_quad_ldsdirect(const Node *__restrict__ nodes, const unsigned *__restrict__ idx, unsigned n_idx, int iters, float *out) {
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4 * 4 * TILE_STRIDE];
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned cur = idx[tid % n_idx];
  const int lane = threadIdx.x & 63, k = lane & 3, qbase = lane & ~3, wv = threadIdx.x >> 6;
  if (n_idx == 0xffffffffu) ((volatile unsigned char *)tiles)[0] = 0; // (keeps the array: the asm below is its only writer)
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)tiles + (unsigned)wv * 4u * TILE_STRIDE));
  const float4 *mine = (const float4 *)(tiles + (wv * 4 + k) * TILE_STRIDE) + qbase;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned nj = __shfl(cur, qbase + j);
      const float4 *p = ((const float4 *)(nodes + nj)) + k;
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(base + (unsigned)j * TILE_STRIDE), "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads are done before the next step's loads overwrite the tiles
    acc += a.x + b.y + c.z + d.w;
    cur = __float_as_uint(d.x);
  }
  out[tid] = acc;
}

int main(int argc, char **argv) {
  const size_t n_nodes = argc > 1 ? atoll(argv[1]) : 3500000; // 224 MB like the 10 M-triangle tree
  const int iters = 64, blocks = 256 * 4, threads = 256;
  std::vector<Node> h(n_nodes);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  // each node points to a random node among the "hot" subset (like traversal reuse): hot fraction selectable
  const double hot = argc > 2 ? atof(argv[2]) : 0.05;
  const size_t n_hot = (size_t)(n_nodes * hot) + 1;
  for (size_t i = 0; i < n_nodes; i++) {
    unsigned nxt = rnd() % n_hot;
    h[i].a = make_float4(1, 2, 3, 4); h[i].b = h[i].a; h[i].c = h[i].a;
    h[i].d = make_float4(__builtin_bit_cast(float, nxt), 0, 0, 1);
  }
  std::vector<unsigned> hidx(blocks * threads);
  for (auto &v : hidx) v = rnd() % n_hot;
  Node *d_nodes; unsigned *d_idx; float *d_out;
  hipMalloc(&d_nodes, n_nodes * sizeof(Node)); hipMalloc(&d_idx, hidx.size() * 4); hipMalloc(&d_out, hidx.size() * 4);
  hipMemcpy(d_nodes, h.data(), n_nodes * sizeof(Node), hipMemcpyHostToDevice);
  hipMemcpy(d_idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 4; v++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (v == 0) k_own<<<blocks, threads>>>(d_nodes, d_idx, (unsigned)hidx.size(), iters, d_out);
      if (v == 1) k_quad<<<blocks, threads>>>(d_nodes, d_idx, (unsigned)hidx.size(), iters, d_out);
      if (v == 2) k_quad_lds<<<blocks, threads>>>(d_nodes, d_idx, (unsigned)hidx.size(), iters, d_out);
      if (v == 3) k_quad_ldsdirect<<<blocks, threads>>>(d_nodes, d_idx, (unsigned)hidx.size(), iters, d_out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fetches = (double)blocks * threads * iters;
      std::vector<float> o(hidx.size()); hipMemcpy(o.data(), d_out, 4 * o.size(), hipMemcpyDeviceToHost);
      double sum = 0; for (float f : o) sum += f;
      printf("variant %d rep %d: %.3f ms  %.1f Gnodes/s  %.2f TB/s  (check %.1f, sum %.1f)\n", v, rep, ms, fetches / ms / 1e6, fetches * 64 / ms / 1e9, o[0], sum);
    }
  }
  return 0;
}
