// How well does the hardware dispatcher keep the CUs full when a launch has more blocks than resident slots?
// Blocks of 256 threads with 32 KB of LDS (five per CU, 1280 resident) that spin for a given number of cycles: uniform durations and
// durations that vary per block (+-50 %).  Work-conserving bound = sum of durations / 1280.  A persistent form (1280 blocks that take
// items from an atomic counter) runs the same work for comparison.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_dispatch tools/probe_dispatch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ void spin(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
__device__ __forceinline__ long long duration(int item, int base, int vary) {
  return vary ? base / 2 + (long long)(hash((unsigned)item) % (unsigned)base) : base;     // mean = base either way
}
__global__ __launch_bounds__(256) void plain_kernel(int base, int vary, float* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  spin(duration(blockIdx.x, base, vary));
  if (lds[threadIdx.x] == -1.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void persistent_kernel(int base, int vary, int items, int* counter, float* out) {
  extern __shared__ float lds[];
  __shared__ int item_s;
  lds[threadIdx.x] = threadIdx.x;
  for (;;) {
    if (threadIdx.x == 0) item_s = atomicAdd(counter, 1);
    __syncthreads();
    const int item = item_s;
    __syncthreads();
    if (item >= items) break;
    spin(duration(item, base, vary));
  }
  if (lds[threadIdx.x] == -1.f) out[0] = 1.f;
}
int main() {
  float* out; CK(hipMalloc(&out, 64));
  int* ctr; CK(hipMalloc(&ctr, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(plain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 - 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // wall_clock64: the 100 MHz constant clock -> "cycles" here are 10 ns ticks
  // 1. which dynamic LDS size really leaves five blocks per CU resident?  1280 uniform blocks of 20 us: one round = 20 us, two = 40
  const int sizes[] = {24576, 28672, 30720, 31744, 32000, 32256, 32512, 32768};
  for (int b : sizes) {
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(plain_kernel, dim3(1280), dim3(256), b, 0, 2000, 0, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    int occ = -1;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(plain_kernel), 256, b));
    printf("dynamic LDS %5d B: 1280 blocks of 20 us take %.1f us (occupancy API: %d per CU)\n", b, ms * 1e3, occ);
  }
  // 2. refill behaviour at a size that does hold five per CU (wall_clock64: the 100 MHz constant clock -> 10 ns ticks)
  const int LDS = 30720;
  const int bases[] = {2000, 6000};      // 20 us, 60 us
  const int counts[] = {1280, 1920, 2904, 3456, 5808, 12800};
  for (int base : bases)
    for (int vary = 0; vary < 2; ++vary)
      for (int n : counts) {
        float ms[2] = {0.f, 0.f};
        for (int form = 0; form < 2; ++form)
          for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ctr, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (form == 0) hipLaunchKernelGGL(plain_kernel, dim3(n), dim3(256), LDS, 0, base, vary, out);
            else hipLaunchKernelGGL(persistent_kernel, dim3(1280), dim3(256), LDS - 64, 0, base, vary, n, ctr, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[form], e0, e1));
          }
        const double bound = (double)n / 1280.0 * base * 0.01;
        printf("block %3d us %s, %5d blocks (%.2f rounds): plain %.1f us, persistent %.1f us | work-conserving bound %.1f us\n",
               base / 100, vary ? "+-50 %" : "uniform", n, n / 1280.0, ms[0] * 1e3, ms[1] * 1e3, bound);
      }
  return 0;
}
