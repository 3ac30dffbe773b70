// K4w7f -- AN EXPERIMENT THAT LOST (round 4; profiles/r04/wino7_fused_attempt.txt): correct, bit-identical to the three-kernel form,
// 0.84 ms against 0.54 ms at P = 384.  Not on the forward; reachable through offk_winograd_conv7x7s2 | OFFK_CONV_WINO7_FUSED, where
// tests/test_gpu_paths.py pins it and tools/w7f_timing_run.py (-DOFFK_W7F_TIMING) reproduces the cycle split: the kernel is bound by
// ISSUING its window loads (133 cycles per buffer_load_dwordx2 of eight 64-byte pieces), not by the transform's arithmetic.
//
// The polyphase Winograd F(5x5, 4x4) form of the 7x7 / stride 2 conv (winograd7.hip) with the INPUT TRANSFORM INSIDE THE GEMM
// KERNEL: the transformed input V (1.0 GB at P = 384 -- written by wino7_input_kernel at 5 TB/s, read back by the 64 batched GEMMs)
// never exists in HBM.
//
// Why this is not "transform in the GEMM's loader": a GEMM block of the three-kernel form owns ONE point p for 64 tiles; an element
// V_p[tile][ci] is an 8x8 dot product of the raw window, and computing it per point throws away the separable transform's sharing
// across the 64 points (64 FMAs per element instead of 7.5; the fp32 MFMA shares its lanes with the VALU).  So a block here owns MANY
// points of FEW tiles: 32 tiles x 32 points (one half of the 8x8 point grid: point rows 4 PH .. 4 PH + 3) x all 64 output channels --
// 32 x 32 x 64 x 4 B = 256 KB of accumulators, the whole accumulator file of a CU (four waves x 256 registers; a wave = one point
// row = 8 points, each a 32 x 64 tile of two v_mfma_f32_32x32x2_f32 accumulators).  Per K step (one phase image, 16 channels):
//   every thread loads the 8x8 window of ONE (tile, channel pair) of that phase image (64 x 8-byte loads, issued one step ahead),
//   transforms it for the block's four point rows with packed fp32 math (column pass for 4 rows, row pass: ~280 v_pk_* per step against
//   8192 cycles of MFMAs per wave) and writes the 32 x float2 results into the LDS image V [32 points][32 tiles][16 channels] (80-byte
//   rows: the ds_read_b128 of a lane group is conflict-free);
//   every wave multiplies its 8 points: A = V_p (LDS), B = the transformed weights straight from L2 in OPERAND ORDER (wino7_pack_fused:
//   per (point, K step) one 4 KB block [co half][k group][lane] float4 -- a wave instruction reads 1 KB contiguous.  The first version read
//   wino7_weight_kernel's [point][co][k] rows: 16 bytes per lane from 32 different rows, four times the line traffic, and the kernel took
//   857 us against 540 us for the two launches it replaces).
// The two blocks of a tile group (PH = 0 / 1) repeat the window loads but share no arithmetic: the separable transform splits by point
// row.  Zero products are skipped as in winograd7.hip: point row 0 only meets the phases with pa = 1, point column 0 those with pb = 1.
// Output: M [64 points][T tiles][64] exactly as the batched GEMMs leave it -- wino7_output_kernel finishes the conv.
// Needs Co == 64, Ci % 16 == 0 and the input below 1 GiB (an out-of-image window sample is a buffer offset past the descriptor).
#include <cstdio>
#include <cstdlib>

// (out of the product build since round 5; to build it again: add to build.py SOURCES, restore the declarations in offk_internal.h and the
// OFFK_CONV_WINO7_FUSED branch of offk_winograd_conv7x7s2 from git history)
#include "../../optical-flow-guided-feature-pytorch_amd/csrc/offk_common.h"
#include "../../optical-flow-guided-feature-pytorch_amd/csrc/offk_internal.h"

namespace offk {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTB = 32;                  // tiles per block
constexpr int kCh = 16;                  // channels per K step
constexpr int kVRow = 80;                // bytes of one (point, tile) row of the V image: 16 channels + 16 bytes of padding
constexpr int kVPoint = kTB * kVRow;     // 2560
constexpr int kVBytes = 32 * kVPoint;    // 81,920: one stage (a wave multiplies and transforms in turn -- nothing to overlap)
constexpr int kOOB = 0x40000000;         // a window row / column outside the phase image: past every descriptor (< 1 GiB input)

// B^T of F(5, 4), points 0, 1, -1, 1/2, -1/2, 2, -2, infinity (winograd7.hip bt8, tools/gen_winograd_f54.py) on channel PAIRS
__device__ __forceinline__ void bt8v(const f32x2 (&d)[8], f32x2 (&t)[8]) {
  t[0] = -5.25f * d[4] + (5.25f * d[2] - d[0]) + d[6];
  t[1] = -4.25f * d[4] + (-4.25f * d[3] + (d[1] + d[2])) + d[5] + d[6];
  t[2] = -4.25f * d[4] + (4.25f * d[3] + (-d[1] + d[2])) - d[5] + d[6];
  t[3] = 0.5f * d[5] + (-5.0f * d[4] + (-2.5f * d[3] + (4.0f * d[2] + 2.0f * d[1]))) + d[6];
  t[4] = -0.5f * d[5] + (-5.0f * d[4] + (2.5f * d[3] + (4.0f * d[2] - 2.0f * d[1]))) + d[6];
  t[5] = 2.0f * d[5] + (-1.25f * d[4] + (-2.5f * d[3] + (0.25f * d[2] + 0.5f * d[1]))) + d[6];
  t[6] = -2.0f * d[5] + (-1.25f * d[4] + (2.5f * d[3] + (0.25f * d[2] - 0.5f * d[1]))) + d[6];
  t[7] = -5.25f * d[5] + (5.25f * d[3] - d[1]) + d[7];
}
}  // namespace

struct Wino7FusedArgs {
  const float* x; int x_cs;      // x already points at the conv's first channel: [n_img * 784][x_cs]
  unsigned x_bytes;              // bytes behind x (< 2^30)
  int n_img, Ci;
  const float* U;                // transformed weights in operand order (wino7_pack_fused_launch): [64 points][NS steps][2][2][64] float4
  float* M;                      // [64][T][64]
#ifdef OFFK_W7F_TIMING
  unsigned long long* dbg;       // cycle sums of thread 0 of every block (tools only)
#endif
};

// PH: which half of the point rows.  The two instantiations are NOT inlined into the kernel: inlined side by side hipcc spilled 216
// registers in the second body (alone each compiles to 480 registers, no spill), and with PH a runtime value 254.
template <int PH>
__device__ __attribute__((noinline)) void w7f_body(const Wino7FusedArgs& a, char* lds, int grp) {
  constexpr int Co = 64;
  const int T = a.n_img * kWino7Tiles, Ci = a.Ci;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p = 4 * PH + wave;                                   // this wave's point row (scalar)

  // ---- transform side: the thread's (tile, channel pair) ----
  const int cp = lane & 7, tl = wave * 8 + (lane >> 3);
  int rowoff[8], coloff[8];
  {
    const int tile = grp * kTB + tl;
    const bool tile_ok = tile < T;
    const int img = tile / kWino7Tiles, tt = tile - img * kWino7Tiles, oy = (tt / 3) * 5, ox = (tt % 3) * 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = oy - 2 + i, c = ox - 2 + i;
      rowoff[i] = tile_ok && (unsigned)r < 14u ? (img * 784 + 2 * r * 28) * a.x_cs * 4 : kOOB;
      coloff[i] = (unsigned)c < 14u ? (2 * c * a.x_cs + 2 * cp) * 4 : kOOB;
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  f32x2 win[8][8];
  auto load_window = [&](int s) {
    const int ph = s / (Ci / kCh), c0 = (s - ph * (Ci / kCh)) * kCh, pa = ph >> 1, pb = ph & 1;
    const int soff = ((pa * 28 + pb) * a.x_cs + c0) * 4;         // phase image (pa, pb): pixel (2 r + pa, 2 c + pb)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int ro = rowoff[i];
      asm volatile("" : "+v"(ro));       // the 64 sums below are loop-invariant: hoisted they are 64 registers for the whole K loop
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, ro + coloff[j], soff, 0);
        win[i][j] = f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
      }
    }
  };
  char* const vwr = lds + tl * kVRow + cp * 8;                   // + point * kVPoint
  auto transform = [&]() {
    // column pass for the block's four point rows (B^T d), then the row pass ((B^T d) B) per point row
    f32x2 tcol[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x2 d[8], t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = win[i][j];
      bt8v(d, t);                                                // (the four unused rows are dead code)
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) tcol[pp][j] = t[4 * PH + pp];
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      f32x2 v[8];
      bt8v(tcol[pp], v);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x2*>(vwr + (pp * 8 + q) * kVPoint) = v[q];
    }
  };

  // ---- GEMM side: the wave's 8 points x 32 tiles x 64 output channels ----
  f32x16 acc[8][2];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][h][r] = 0.f;
  const int j32 = lane & 31, kh = lane >> 5;
  const int NS = 4 * (Ci / kCh);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, 64 * NS * 4096, 0x00020000);
  const int ulane = lane * 16;
  const char* const ard = lds + wave * 8 * kVPoint + j32 * kVRow + kh * 16;        // + q * kVPoint + g * 32
  auto mma_step = [&](int s) {
    const int pb = (s / (Ci / kCh)) & 1;
    // (point row 0 meets the phases pa = 1 only: its operand blocks of the other two phases are zeros -- the wave multiplies them
    //  rather than leave the step: the block waits for its busiest waves either way, and a wave-uniform exit here, with 256 accumulator
    //  registers and the prefetched window live across it, made hipcc spill 600 registers)
    f32x4 av[2][2], uv[2][2][2];                                   // [set][g], [set][h][g]
    auto ld = [&](const int set, const int q) {
#pragma unroll
      for (int g = 0; g < 2; ++g) av[set][g] = *reinterpret_cast<const f32x4*>(ard + q * kVPoint + g * 32);
      const int sb = ((p * 8 + q) * NS + s) * 4096;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, sb + (h * 2 + g) * 1024, 0);
          uv[set][h][g] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
    };
    if (pb) ld(0, 0); else ld(1, 1);                              // point column 0 meets the phases pb = 1 only
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q == 0 && !pb) continue;
      const int set = q & 1;
      if (q + 1 < 8) ld(set ^ 1, q + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[q][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][g].x, uv[set][h][g].x, acc[q][h], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[q][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][g].y, uv[set][h][g].y, acc[q][h], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[q][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][g].z, uv[set][h][g].z, acc[q][h], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[q][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][g].w, uv[set][h][g].w, acc[q][h], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  load_window(0);
  transform();
  __syncthreads();
#ifdef OFFK_W7F_TIMING
  unsigned long long tm[5] = {0, 0, 0, 0, 0}, tq = __builtin_readcyclecounter();
#define W7F_LAP(i) { const unsigned long long q_ = __builtin_readcyclecounter(); tm[i] += q_ - tq; tq = q_; }
#else
#define W7F_LAP(i)
#endif
#pragma unroll 1
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) load_window(s + 1);                          // in flight behind the step's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    W7F_LAP(0)
    mma_step(s);
    __builtin_amdgcn_sched_barrier(0);
    W7F_LAP(1)
    __syncthreads();                                             // every wave has read V of step s
    W7F_LAP(2)
    if (s + 1 < NS) transform();
    W7F_LAP(3)
    __syncthreads();
    W7F_LAP(4)
  }
#ifdef OFFK_W7F_TIMING
  if (a.dbg && (threadIdx.x & 63) == 0) {
    const int w_ = threadIdx.x >> 6;
    for (int i = 0; i < 5; ++i) atomicAdd(a.dbg + w_ * 8 + i, tm[i]);
    atomicAdd(a.dbg + w_ * 8 + 5, 1ull);
  }
#endif
#undef W7F_LAP

  // ---- M[point][tile][co]: register r of a 32x32 accumulator = tile row (r & 3) + 8 (r >> 2) + 4 kh, lane column j32 ----
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int mi = p != 0 && q != 0 ? (p - 1) * 7 + (q - 1) : (p == 0 && q != 0 ? 49 + q - 1 : (q == 0 && p != 0 ? 56 + p - 1 : 63));
    float* mp = a.M + ((size_t)mi * T + (size_t)grp * kTB) * Co + j32;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = acc_row(r, kh);
        if (grp * kTB + i < T) mp[(size_t)i * Co + 32 * h] = acc[q][h][r];
      }
  }
}

__global__ __launch_bounds__(256, 1) void wino7_fused_kernel(Wino7FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int grp = blockIdx.x >> 1;
  if (blockIdx.x & 1) w7f_body<1>(a, lds, grp);
  else w7f_body<0>(a, lds, grp);
}

// U (the four groups of wino7_weight_kernel, Co = 64) -> operand order: block (point pq = 8 p + q, step s = phase * Ci / 16 + chunk) =
// [h 2][g 2][lane 64] float4 with element e = U_(p,q)[co = 32 h + (lane & 31)][phase][ci = 16 chunk + 8 g + 4 (lane >> 5) + e]; the
// (point, phase) products that winograd7.hip skips (point row 0 x pa = 0, point column 0 x pb = 0) are zero blocks
__global__ __launch_bounds__(256) void wino7_pack_fused_kernel(const float* __restrict__ U, int Ci, float4* __restrict__ out) {
  constexpr int Co = 64;
  const int NS = 4 * (Ci / kCh);
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)64 * NS * 256) return;
  const int lane = (int)(item & 63), g = (int)(item >> 6) & 1, h = (int)(item >> 7) & 1;
  const int s = (int)((item >> 8) % NS), pq = (int)((item >> 8) / NS), p = pq >> 3, q = pq & 7;
  const int ph = s / (Ci / kCh), c0 = (s - ph * (Ci / kCh)) * kCh, pa = ph >> 1, pb = ph & 1;
  const int co = 32 * h + (lane & 31), ci = c0 + 8 * g + 4 * (lane >> 5);
  long long src = -1;
  if (p != 0 && q != 0) src = ((long long)(((p - 1) * 7 + (q - 1)) * Co + co) * 4 + (2 * pa + pb)) * Ci + ci;
  else if (p == 0 && q != 0) { if (pa == 1) src = (long long)196 * Co * Ci + ((long long)((q - 1) * Co + co) * 2 + pb) * Ci + ci; }
  else if (q == 0 && p != 0) { if (pb == 1) src = (long long)210 * Co * Ci + ((long long)((p - 1) * Co + co) * 2 + pa) * Ci + ci; }
  else if (pa == 1 && pb == 1) src = (long long)224 * Co * Ci + (long long)co * Ci + ci;
  out[item] = src >= 0 ? *reinterpret_cast<const float4*>(U + src) : make_float4(0.f, 0.f, 0.f, 0.f);
}
size_t wino7_fused_weight_floats(int Ci) { return (size_t)64 * 4 * (Ci / kCh) * 1024; }
hipError_t wino7_pack_fused_launch(const float* U, int Ci, float* out, hipStream_t st) {
  const long long n = (long long)64 * 4 * (Ci / kCh) * 256;
  hipLaunchKernelGGL(wino7_pack_fused_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, U, Ci, reinterpret_cast<float4*>(out));
  return hipGetLastError();
}

bool wino7_fused_supported(int n_img, int Ci, int Co, int x_cs) {
  return Co == 64 && Ci >= kCh && Ci % kCh == 0 && (unsigned long long)n_img * 784ull * x_cs * 4ull < (unsigned long long)kOOB;
}

hipError_t wino7_fused_launch(const float* x, int x_cs, int x_coff, int n_img, int Ci, int Co, const float* U, float* M, hipStream_t st) {
  if (!wino7_fused_supported(n_img, Ci, Co, x_cs) || !x || !U || !M || x_coff % 2) return hipErrorInvalidValue;
  Wino7FusedArgs a;
  a.x = x + x_coff; a.x_cs = x_cs; a.x_bytes = (unsigned)(((unsigned long long)n_img * 784ull * x_cs - x_coff) * 4ull);
  a.n_img = n_img; a.Ci = Ci; a.U = U; a.M = M;
#ifdef OFFK_W7F_TIMING
  {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc(reinterpret_cast<void**>(&dbg), 256); (void)hipMemset(dbg, 0, 256); }
    a.dbg = dbg;
    if (getenv("OFFK_W7F_TIMING_DUMP")) {
      unsigned long long hb[32];
      (void)hipMemcpy(hb, dbg, 256, hipMemcpyDeviceToHost);
      for (int w_ = 0; w_ < 4; ++w_)
        if (hb[w_ * 8 + 5])
          fprintf(stderr, "[w7f timing, wave %d, cycles per block] load issue %llu  mma %llu  barrier1 %llu  transform %llu  barrier2 %llu  (blocks %llu)\n", w_,
                  hb[w_ * 8] / hb[w_ * 8 + 5], hb[w_ * 8 + 1] / hb[w_ * 8 + 5], hb[w_ * 8 + 2] / hb[w_ * 8 + 5], hb[w_ * 8 + 3] / hb[w_ * 8 + 5],
                  hb[w_ * 8 + 4] / hb[w_ * 8 + 5], hb[w_ * 8 + 5]);
      (void)hipMemset(dbg, 0, 256);
    }
  }
#endif
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(wino7_fused_kernel), kVBytes);
  if (e != hipSuccess) return e;
  const int groups = (n_img * kWino7Tiles + kTB - 1) / kTB;
  hipLaunchKernelGGL(wino7_fused_kernel, dim3(2 * groups), dim3(256), kVBytes, st, a);
  return hipGetLastError();
}

}  // namespace offk
