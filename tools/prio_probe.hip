// Does the dispatcher honour HIP stream priorities between two launches that both want every CU slot?  (Round 5: would a frame cut into two
// slices pipeline -- slice 2's closest-hit launch filling in as slice 1's drains -- if slice 1's stream has the higher priority?)
// Kernel A: 4 x the chip's resident capacity of 256-thread blocks with 24 KiB of LDS each (like k_trace), every block busy for `us` microseconds.
// Kernel B: one capacity's worth of the same blocks, launched on another stream while A's first generation is resident.  Each block records when
// it started.  Reported: when B's blocks started relative to A's generations, for B's stream at the same / a higher / a lower priority than A's.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/prio_probe tools/prio_probe.hip && /tmp/prio_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_busy(unsigned long long *start, unsigned long long *stop, unsigned long long ticks) {
  __shared__ int pad[6144]; // 24 KiB: 5-6 blocks per CU, like the traversal kernels
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) start[blockIdx.x] = t0;
  pad[threadIdx.x] = (int)t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) stop[blockIdx.x] = wall_clock64() + (pad[(threadIdx.x + 1) & 255] & 0);
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  int lo = 0, hi = 0;
  CHK(hipDeviceGetStreamPriorityRange(&lo, &hi)); // lo = least priority (largest number), hi = greatest
  int wall_khz = 0;
  CHK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  std::printf("%s, %d CUs, stream priorities %d (least) .. %d (greatest), wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, lo, hi, wall_khz);
  int per_cu = 0;
  CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_busy, 256, 0));
  const int cap = per_cu * prop.multiProcessorCount, nA = 4 * cap, nB = cap;
  const double khz = wall_khz ? wall_khz : 100000.0;
  const unsigned long long ticks = (unsigned long long)(100.0 * khz / 1000.0); // 100 us per block
  unsigned long long *sA, *eA, *sB, *eB;
  CHK(hipMalloc(&sA, 8 * nA)); CHK(hipMalloc(&eA, 8 * nA)); CHK(hipMalloc(&sB, 8 * nB)); CHK(hipMalloc(&eB, 8 * nB));
  std::printf("%d blocks per CU -> capacity %d blocks; A = %d blocks (4 generations of 100 us), B = %d blocks, launched ~30 us after A\n", per_cu, cap, nA, nB);
  const struct { const char *name; int pa, pb; } cases[] = { { "same priority", 0, 0 }, { "B greater", lo, hi }, { "B lesser", hi, lo } };
  for (const auto &c : cases) {
    hipStream_t a, b;
    CHK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, c.pa));
    CHK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, c.pb));
    for (int rep = 0; rep < 2; rep++) { // the first repetition warms up
      k_busy<<<nA, 256, 0, a>>>(sA, eA, ticks);
      k_busy<<<1, 256, 0, b>>>(sB, eB, ticks * 3 / 10); // ~30 us of head start for A (B's stream is busy with this block meanwhile)
      k_busy<<<nB, 256, 0, b>>>(sB, eB, ticks);
      CHK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> hsA(nA), hsB(nB), heA(nA), heB(nB);
    CHK(hipMemcpy(hsA.data(), sA, 8 * nA, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hsB.data(), sB, 8 * nB, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(heA.data(), eA, 8 * nA, hipMemcpyDeviceToHost)); CHK(hipMemcpy(heB.data(), eB, 8 * nB, hipMemcpyDeviceToHost));
    const unsigned long long t0 = *std::min_element(hsA.begin(), hsA.end());
    auto us = [&](unsigned long long t) { return (double)(t - t0) / khz * 1000.0; };
    std::sort(hsA.begin(), hsA.end()); std::sort(hsB.begin(), hsB.end());
    std::printf("%-14s A starts: 25%% %.0f  50%% %.0f  75%% %.0f  last %.0f us, A ends %.0f | B starts: first %.0f  25%% %.0f  50%% %.0f  75%% %.0f  last %.0f us, B ends %.0f\n", c.name,
                us(hsA[nA / 4]), us(hsA[nA / 2]), us(hsA[3 * nA / 4]), us(hsA[nA - 1]), us(*std::max_element(heA.begin(), heA.end())), us(hsB[0]), us(hsB[nB / 4]), us(hsB[nB / 2]),
                us(hsB[3 * nB / 4]), us(hsB[nB - 1]), us(*std::max_element(heB.begin(), heB.end())));
    CHK(hipStreamDestroy(a)); CHK(hipStreamDestroy(b));
  }
  return 0;
}
