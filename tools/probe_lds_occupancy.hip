// How many blocks of 256 threads fit a CU for a given dynamic LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor): finds the LDS
// allocation granule of gfx950 (160 KB per CU).   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_lds tools/probe_lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* out) {
  extern __shared__ float s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int sizes[] = {29696, 30720, 31744, 32000, 32256, 32768, 33280, 36864, 40960, 53760, 54272, 54613, 65536, 81920};
  for (int b : sizes) {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), 256, b);
    printf("dynamic LDS %6d B: %d blocks per CU (%s)\n", b, n, hipGetErrorString(e));
  }
  return 0;
}
