// Calibration probe: what HBM rate does this MI355X box actually sustain for f4 streaming?
// (MI355X_MICROARCH.md quotes 6.29 TB/s for a f4 copy; K2's roofline fraction is judged
// against the 8 TB/s spec, this probe says how much of the gap is the platform's.)
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o gpurun_out/hbm_probe && gpurun_out/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (; i + 256 * (UNROLL - 1) < n; i += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + 256 * u) : src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + 256 * u); else dst[i + 256 * u] = v[u];
    }
  }
}
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ src, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  float acc = 0.f;
  for (; i + 256 * (UNROLL - 1) < n; i += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + 256 * u) : src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 1234.5f) out[0] = acc;
}
template <int NT>
__global__ __launch_bounds__(256) void write_k(f4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

template <typename F>
static float time_us(F f, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const size_t bytes = (size_t)700 << 20, n = bytes / 16;
  f4 *src, *dst; float* out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
    float t;
    t = time_us([&] { hipLaunchKernelGGL((copy_k<0, 4>), dim3(grid), dim3(256), 0, 0, src, dst, n); }, 20);
    printf("grid %6d copy   u4     %8.1f us  %7.1f GB/s\n", grid, t, 2.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((copy_k<1, 4>), dim3(grid), dim3(256), 0, 0, src, dst, n); }, 20);
    printf("grid %6d copy   u4 nt  %8.1f us  %7.1f GB/s\n", grid, t, 2.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((copy_k<0, 8>), dim3(grid), dim3(256), 0, 0, src, dst, n); }, 20);
    printf("grid %6d copy   u8     %8.1f us  %7.1f GB/s\n", grid, t, 2.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((read_k<0, 8>), dim3(grid), dim3(256), 0, 0, src, out, n); }, 20);
    printf("grid %6d read   u8     %8.1f us  %7.1f GB/s\n", grid, t, 1.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((read_k<1, 8>), dim3(grid), dim3(256), 0, 0, src, out, n); }, 20);
    printf("grid %6d read   u8 nt  %8.1f us  %7.1f GB/s\n", grid, t, 1.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((write_k<0>), dim3(grid), dim3(256), 0, 0, dst, n); }, 20);
    printf("grid %6d write         %8.1f us  %7.1f GB/s\n", grid, t, 1.0 * bytes / t / 1e3);
    t = time_us([&] { hipLaunchKernelGGL((write_k<1>), dim3(grid), dim3(256), 0, 0, dst, n); }, 20);
    printf("grid %6d write  nt     %8.1f us  %7.1f GB/s\n", grid, t, 1.0 * bytes / t / 1e3);
  }
  return 0;
}
