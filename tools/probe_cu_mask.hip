// Probe: does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel to its CUs on this box, how do mask bits map to
// (XCD, CU), and what HBM rate does a copy kernel reach on a 1/8 or 1/4 slice of the chip -- alone and beside a compute-bound kernel
// on the complementary slice?  (Round 4: can the HBM-bound Winograd input transforms run BESIDE the MFMA-bound units kernel?)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/probe_cu_mask tools/probe_cu_mask.hip && tools/_bin/probe_cu_mask
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // burn a little time so that one wave per CU is not enough to swallow the grid
  float a = threadIdx.x;
  for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw + (a == 12345.f); }
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 12345.f) out[threadIdx.x] = s;
}

static hipStream_t masked(const std::vector<unsigned>& m) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (unsigned)m.size(), m.data()));
  return s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("CUs %d\n", ncu);
  const int words = (ncu + 31) / 32;
  std::vector<unsigned> first32(words, 0u), every8(words, 0u), rest8(words, 0u), every4(words, 0u), rest4(words, 0u);
  first32[0] = 0xffffffffu;
  // (bit i lands on XCD i % 8 -- the first 32 bits are four CUs of every XCD; a mask that leaves an XCD without any CU is ignored
  //  altogether: "every 8th bit" ran on all 256 CUs.)  every8 / rest8 here: bits [0, 32) / [32, 256); every4 / rest4: [0, 64) / [64, 256)
  for (int i = 0; i < ncu; ++i) {
    (i < 32 ? every8 : rest8)[i / 32] |= 1u << (i % 32);
    (i < 64 ? every4 : rest4)[i / 32] |= 1u << (i % 32);
  }
  hipStream_t s_first = masked(first32), s_e8 = masked(every8), s_r8 = masked(rest8), s_e4 = masked(every4), s_r4 = masked(rest4), s_all;
  CK(hipStreamCreate(&s_all));

  unsigned* d_where; CK(hipMalloc(&d_where, 8 * 4096));
  std::vector<unsigned> h(2 * 4096);
  auto survey = [&](const char* name, hipStream_t st) {
    CK(hipMemsetAsync(d_where, 0xff, 8 * 4096, st));
    hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, st, d_where);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d_where, 8 * 4096, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int per_xcc[8] = {0};
    std::set<unsigned> per_xcc_cu[8];
    for (int i = 0; i < 4096; ++i) {
      const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
      per_xcc_cu[xcc & 7].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%-10s distinct (xcc, se, sh, cu): %zu | per XCD:", name, cus.size());
    for (int x = 0; x < 8; ++x) { per_xcc[x] = (int)per_xcc_cu[x].size(); printf(" %d", per_xcc[x]); }
    printf("\n");
  };
  survey("all", s_all);
  survey("first32", s_first);
  survey("low 32", s_e8);
  survey("high 224", s_r8);
  survey("low 64", s_e4);
  survey("high 192", s_r4);

  const size_t bytes = 1ull << 30, n = bytes / 16;
  float4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  float* d_out; CK(hipMalloc(&d_out, 4096));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  auto time_copy = [&](const char* name, hipStream_t st, int blocks) {
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, src, dst, n);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, src, dst, n);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy 1 GiB on %-10s (%5d blocks): %.3f ms = %.2f TB/s (read + write)\n", name, blocks, ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
  };
  time_copy("all", s_all, 256 * 8);
  time_copy("low 32", s_e8, 32 * 8);
  time_copy("low 32", s_e8, 32 * 16);
  time_copy("low 32", s_e8, 32 * 32);
  time_copy("low 64", s_e4, 64 * 8);
  time_copy("low 64", s_e4, 64 * 16);
  time_copy("high 224", s_r8, 224 * 8);

  auto time_mfma = [&](const char* name, hipStream_t st, int blocks, int iters) {
    hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(f0, st));
    hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters);
    CK(hipEventRecord(f1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, f0, f1));
    const double fl = (double)blocks * 4 * iters * 4 * 32 * 32 * 2 * 2;
    printf("mfma on %-10s (%5d blocks x %d iters): %.3f ms = %.1f TF\n", name, blocks, iters, ms, fl / (ms * 1e-3) / 1e12);
    return ms;
  };
  const int iters = 10000;
  time_mfma("all", s_all, 256 * 8, iters);
  time_mfma("high 224", s_r8, 224 * 8, iters);
  time_mfma("all", s_all, 224 * 8, iters);

  // both at once: the MFMA kernel on 7/8 of the CUs, the copy on the other 1/8
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(f0, s_r8));
    hipLaunchKernelGGL(mfma_kernel, dim3(224 * 8), dim3(256), 0, s_r8, d_out, iters);
    CK(hipEventRecord(f1, s_r8));
    CK(hipEventRecord(e0, s_e8));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(copy_kernel, dim3(32 * 16), dim3(256), 0, s_e8, src, dst, n);
    CK(hipEventRecord(e1, s_e8));
    CK(hipDeviceSynchronize());
    float ma, mb; CK(hipEventElapsedTime(&ma, f0, f1)); CK(hipEventElapsedTime(&mb, e0, e1));
    printf("concurrent: mfma on high 224 %.3f ms | 2 copies on low 32 %.3f ms = %.2f TB/s\n", ma, mb, 4.0 * bytes / (mb * 1e-3) / 1e12);
  }
  // the same pair WITHOUT masks (two plain streams): what the round-4 lane experiment had
  {
    hipStream_t s2; CK(hipStreamCreate(&s2));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(f0, s_all));
    hipLaunchKernelGGL(mfma_kernel, dim3(256 * 8), dim3(256), 0, s_all, d_out, iters);
    CK(hipEventRecord(f1, s_all));
    CK(hipEventRecord(e0, s2));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, s2, src, dst, n);
    CK(hipEventRecord(e1, s2));
    CK(hipDeviceSynchronize());
    float ma, mb; CK(hipEventElapsedTime(&ma, f0, f1)); CK(hipEventElapsedTime(&mb, e0, e1));
    printf("concurrent, no masks: mfma %.3f ms | 2 copies %.3f ms\n", ma, mb);
  }
  return 0;
}
