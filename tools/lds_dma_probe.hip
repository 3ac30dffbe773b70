#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each block: 256 threads; loads 4 KB from src (with per-lane offsets) directly to LDS, then copies LDS to dst
__global__ void k(const float* src, int nbytes, float* dst, int oob_lane) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // fill lds with sentinel
  reinterpret_cast<float4*>(lds)[tid] = make_float4(-1.f, -1.f, -1.f, -1.f);
  __syncthreads();
  int voff = tid * 16;
  if (lane == oob_lane) voff = (int)0x80000000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  reinterpret_cast<float4*>(dst)[tid] = reinterpret_cast<float4*>(lds)[tid];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
  hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, n * 4, d, 5);
  std::vector<float> o(n); hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) { float e = (float)i; int t = i / 4; if ((t & 63) == 5) e = 0.f; if (o[i] != e) { if (bad < 10) printf("i=%d got %f exp %f\n", i, o[i], e); ++bad; } }
  printf("bad=%d  (lane5 values: %f %f)\n", bad, o[20], o[21]);
  return 0;
}
