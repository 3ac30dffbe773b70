#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// LDS-DMA to offsets above 64 KB: dynamic LDS of 144 KB, each wave lands 1 KB at base + off, then read back
__global__ void k(const float* src, int nbytes, float* dst, unsigned off) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long a = reinterpret_cast<unsigned long long>(src);
  i32x4 d = {(int)(unsigned)a, (int)(unsigned)(a >> 32) & 0xffff, nbytes, 0x00020000};
  reinterpret_cast<float4*>(lds + off)[tid] = make_float4(-1.f, -1.f, -1.f, -1.f);
  reinterpret_cast<float4*>(lds)[tid] = make_float4(-2.f, -2.f, -2.f, -2.f);
  __syncthreads();
  unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + off + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024;
  int voff = tid * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff), "s"(d) : "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  reinterpret_cast<float4*>(dst)[tid] = reinterpret_cast<float4*>(lds + off)[tid];
  reinterpret_cast<float4*>(dst)[256 + tid] = reinterpret_cast<float4*>(lds)[tid];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *s, *d; (void)hipMalloc(&s, n * 4); (void)hipMalloc(&d, 2 * n * 4);
  (void)hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  for (unsigned off : {0u, 32768u, 65536u, 98304u, 131072u, 143360u - 4096u}) {
    (void)hipMemset(d, 0, 2 * n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 147456, 0, s, n * 4, d, off);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> o(2 * n); (void)hipMemcpy(o.data(), d, 2 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0, low = 0;
    for (int i = 0; i < n; ++i) if (o[i] != (float)i) ++bad;
    for (int i = 0; i < n; ++i) if (off != 0 && o[n + i] != -2.f) ++low;
    printf("off %6u: err %d bad %d (first got %f)  low-region clobbered %d\n", off, (int)e, bad, o[0], low);
  }
  return 0;
}
