// Probe: what does v_mfma_f32_32x32x2_f32 sustain on MI355X under the loop shapes the fp32 kernels use?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_probe.hip -o tools/_bin/mfma_f32_probe
// variants: NACC accumulator tiles per wave (1 = the 64x64 conv tile, 2 = 64x128, 4 = 128x128, 5 = K1),
//           LDS: 0 operands in registers, 1 operands re-read from LDS with ds_read_b128 per 4 MFMAs (as the kernels do),
//           BAR: barrier every 16*NACC MFMAs (one per K-tile), blocks per CU via grid / LDS padding.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// second experiment (round 2): the same loop with VALU extra vector-ALU instructions and WR ds_write_b128 per K-tile, the
// operand reads alternating between two LDS stages -- what the real conv loop carries besides its MFMAs
template <int NACC, int VALU, int WR>
__global__ __launch_bounds__(256) void mfma_busy_k(float* out, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int STAGE = (32 * 4 + 32 * NACC * 4) * 36;
  for (int i = threadIdx.x; i < 2 * STAGE; i += 256) smem[i] = (float)((i * 7) & 15) * 0.125f;
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int r32 = lane & 31, h = lane >> 5;
  unsigned junk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) junk[j] = threadIdx.x * (j + 3);
  float4 wv = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int kt = 0; kt < ktiles; ++kt) {
    const float* As = smem + (kt & 1) * STAGE + wave * 32 * 36;
    const float* Bs = smem + (kt & 1) * STAGE + 4 * 32 * 36;
#pragma unroll
    for (int v = 0; v < VALU; ++v) {       // address-math stand-in: dependent chains of 8 independent lanes of work
      junk[v & 7] = junk[v & 7] * 1664525u + (unsigned)kt;
      asm volatile("" : "+v"(junk[v & 7]));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(As + r32 * 36 + 8 * g + 4 * h);
      float4 b[NACC];
#pragma unroll
      for (int t = 0; t < NACC; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + (t * 32 + r32) * 36 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int w = 0; w < WR; ++w)           // staging stand-in: stores into the OTHER stage (rows this wave owns)
      *reinterpret_cast<float4*>(smem + ((kt + 1) & 1) * STAGE + ((threadIdx.x >> 3) + 32 * w) * 36 + 4 * (threadIdx.x & 7)) = wv;
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  unsigned js = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) js += junk[j];
  if (s == 12345.678f || js == 0x12345u) out[0] = s;
}

template <int NACC, int VALU, int WR>
static int run_busy(int blocks_per_cu, float* out) {
  const int ktiles = 2000 / NACC;
  const size_t need = 2 * (size_t)(32 * 4 + 32 * NACC * 4) * 36 * 4;
  size_t lds = 160 * 1024 / blocks_per_cu - 512;
  if (lds < need) lds = need;
  auto kern = mfma_busy_k<NACC, VALU, WR>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double flops = (double)grid * 4 * ktiles * 16.0 * NACC * 4096.0;
  printf("%d acc, %3d VALU + %d ds_write_b128 per K-tile, 2 stages   blocks/CU %d  %8.3f ms  %7.1f TF  (%.0f %%)\n", NACC, VALU, WR,
         blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
  return 0;
}

template <int NACC, int LDS, int BAR>
__global__ __launch_bounds__(256) void mfma_k(float* out, int ktiles, int lds_pad_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // operand tiles: A 32 rows x 32 k, B NACC*32 rows x 32 k, row stride 36 floats (as LDS_K)
  for (int i = threadIdx.x; i < (32 * 4 + 32 * NACC * 4) * 36; i += 256) smem[i] = (float)((i * 7) & 15) * 0.125f;
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int r32 = lane & 31, h = lane >> 5;
  const float* As = smem + wave * 32 * 36;
  const float* Bs = smem + 4 * 32 * 36;
  float4 a = *reinterpret_cast<const float4*>(As + r32 * 36 + 4 * h);
  float4 b[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + (t * 32 + r32) * 36 + 4 * h);
  for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (LDS) {
        a = *reinterpret_cast<const float4*>(As + r32 * 36 + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < NACC; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + (t * 32 + r32) * 36 + 8 * g + 4 * h);
      }
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
      }
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC, int LDS, int BAR>
static int run(const char* name, int blocks_per_cu, float* out) {
  const int ktiles = 2000 / NACC;
  const size_t need = (size_t)(32 * 4 + 32 * NACC * 4) * 36 * 4;
  size_t lds = 160 * 1024 / blocks_per_cu - 512;          // pad so that exactly blocks_per_cu blocks fit a CU
  if (lds < need) lds = need;
  auto kern = mfma_k<NACC, LDS, BAR>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles, 0);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, ktiles, 0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double flops = (double)grid * 4 /*waves*/ * ktiles * 16.0 * NACC * 4096.0;
  printf("%-44s blocks/CU %d  %8.3f ms  %7.1f TF  (%.0f %% of 157.3)\n", name, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
  return 0;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 4));
  for (int bpc : {1, 2, 4}) {
    run<1, 0, 0>("1 acc, regs, no barrier", bpc, out);
    run<1, 1, 0>("1 acc, LDS operands, no barrier", bpc, out);
    run<1, 1, 1>("1 acc, LDS operands, barrier / K-tile", bpc, out);
    run<2, 1, 1>("2 acc, LDS operands, barrier / K-tile", bpc, out);
    run<4, 1, 1>("4 acc, LDS operands, barrier / K-tile", bpc, out);
    run<5, 1, 1>("5 acc (K1), LDS operands, barrier / K-tile", bpc, out);
    run<5, 0, 0>("5 acc, regs, no barrier", bpc, out);
  }
  for (int bpc : {1, 2, 4}) {
    run_busy<1, 0, 0>(bpc, out);
    run_busy<1, 32, 0>(bpc, out);
    run_busy<1, 64, 0>(bpc, out);
    run_busy<1, 128, 0>(bpc, out);
    run_busy<1, 0, 4>(bpc, out);
    run_busy<1, 64, 4>(bpc, out);
    run_busy<4, 0, 0>(bpc, out);
    run_busy<4, 128, 0>(bpc, out);
    run_busy<4, 128, 8>(bpc, out);
    run_busy<4, 256, 8>(bpc, out);
  }
  return 0;
}
