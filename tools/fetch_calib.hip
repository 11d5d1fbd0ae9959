// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the traversal's access pattern (MI355X guide, "HBM": widths other than
// a wide streaming read are uncalibrated).  Every 64-byte node of a table far larger than the 256 MiB Infinity Cache is read exactly
// once, scattered (index = i * odd mod 2^k), with the kernel's own four global_load_dwordx4 per lane.  Useful bytes are exactly
// N * 64; the ratio FETCH_SIZE / (N * 64) is the factor by which the counter over- or under-states this pattern, and the elapsed time
// bounds what a node fetch really moves (N * 64 B/s vs N * 128 B/s against ~6 TB/s achievable).
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Node { float4 a, b, c, d; };
__global__ __launch_bounds__(256) void k_gather_once(const Node *__restrict__ nodes, unsigned log2n, unsigned per_thread, float *out) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long mask = (1ull << log2n) - 1;
  float acc = 0.f;
  for (unsigned k = 0; k < per_thread; k++) {
    const unsigned long long i = tid * per_thread + k;
    const unsigned long long j = (i * 0x9E3779B1ull) & mask; // odd multiplier: a bijection on [0, 2^log2n)
    const Node *nd = nodes + j;
    const float4 a = nd->a, b = nd->b, c = nd->c, d = nd->d;
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ p, size_t n4, float *out) { // the guide's calibrated case
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
int main(int argc, char **argv) {
  const unsigned log2n = argc > 1 ? atoi(argv[1]) : 25; // 2^25 nodes = 2 GiB
  const size_t n = 1ull << log2n;
  Node *d; float *o;
  if (hipMalloc(&d, n * sizeof(Node)) != hipSuccess || hipMalloc(&o, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, n * sizeof(Node));
  const unsigned per_thread = 16, threads = 256;
  const unsigned blocks = (unsigned)(n / per_thread / threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k_gather_once<<<blocks, threads>>>(d, log2n, per_thread, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gather_once: %zu nodes, useful %.1f MB, %.3f ms -> %.2f Gnodes/s = %.2f TB/s at 64 B/node (%.2f TB/s if a node fetch moves a 128-B line)\n",
           n, n * 64 / 1e6, ms, n / ms / 1e6, n * 64.0 / ms / 1e9, n * 128.0 / ms / 1e9);
  }
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k_stream<<<256 * 32, threads>>>((const float4 *)d, n * 4, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stream: %.1f MB, %.3f ms -> %.2f TB/s\n", n * 64 / 1e6, ms, n * 64.0 / ms / 1e9);
  }
  return 0;
}
