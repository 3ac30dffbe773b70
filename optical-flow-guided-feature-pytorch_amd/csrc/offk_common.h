// Shared device-side building blocks for liboffk (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace offk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------
// fp32 MFMA tile core.
//
// v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D[i][j] lives in lane (j + 32*((i>>2)&1)), register (i&3) + 4*(i>>3).
// Both LDS tiles are stored row-major [row][k] with a row stride of LDS_K floats
// (BK = 32 plus 4 pad words): every global->LDS store and LDS->register load is 16 B
// wide, and the 36-word stride makes the ds_read_b128 of 16 consecutive rows
// conflict-free (36*i mod 64 takes 16 distinct multiples of 4).  A lane reads FOUR
// consecutive k of its row at once; MFMA step e of group g therefore multiplies
// k = 8g+e (lanes 0-31) and k = 8g+4+e (lanes 32-63).  A and B use the same
// permutation of k, so the contraction is unchanged (only the fp32 summation order
// differs from a k-ascending loop).
// ---------------------------------------------------------------------------------
constexpr int BK = 32;
constexpr int LDS_K = BK + 4;

template <int TM, int TN>
struct WaveAcc {
  f32x16 acc[TM][TN];

  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  // As / Bs: first LDS row of this wave's A / B slab.
  // The operand quads of k-group g + 1 are read from LDS BEFORE the MFMAs of group g are issued (two register sets,
  // a scheduling barrier pins the order): hipcc otherwise issues a group's reads behind the previous group's MFMAs
  // and the wave sits out an LDS latency per group.
  __device__ __forceinline__ void mma_ktile(const float* As, const float* Bs, int lane) {
    const int r = lane & 31, h = lane >> 5;
    float4 a[2][TM], b[2][TN];
    auto rd = [&](int g, float4 (&aa)[TM], float4 (&bb)[TN]) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
        aa[t] = *reinterpret_cast<const float4*>(As + (t * 32 + r) * LDS_K + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < TN; ++t)
        bb[t] = *reinterpret_cast<const float4*>(Bs + (t * 32 + r) * LDS_K + 8 * g + 4 * h);
    };
    rd(0, a[0], b[0]);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) rd(g + 1, a[(g + 1) & 1], b[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].x, b[g & 1][tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].y, b[g & 1][tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].z, b[g & 1][tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].w, b[g & 1][tn].w, acc[tm][tn], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// (Rounds 1 - 4: the two-plane "bf16x3" tile core lived here -- split4 / b3_store; retired with that mode in round 5.  The split-fp32
//  kernels, pw_tdiff_split.hip and wino_gemm_split.hip, cut their operands into THREE planes themselves.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// XCD-aware tile order inside one group of n blocks that start at any block id: blocks are dealt round-robin over the
// 8 XCDs (private 4 MiB L2 each), so blocks j, j+8, j+16 ... of the group share an XCD; they get a CONTIGUOUS range
// of logical tiles (bijective for every n).  Neighbouring tiles share partial 128-byte lines (rows of HW floats are
// only 16-byte aligned) and weight rows: on one XCD the second touch is an L2 hit instead of a second fetch.
__device__ __forceinline__ int xcd_contiguous(int j, int n) {
#ifdef OFFK_NO_XCD_REMAP
  return j;
#endif
  const int c = j & 7, idx = j >> 3, q = n >> 3, r = n & 7;
  return (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + idx;
}

// row (M index) of accumulator register `reg` for a lane in half `h` (lane>>5)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// ---- reproducible dropout multiplier of the units' spatial branch (training) ------------------------
// splitmix64 finaliser, the same integer function as offk_amd/synth.py (raw_u64 / dropout_keep): draw
// q = (pair*HW + pixel)*8 + channel quad of the site's stream serves the quad's four channels, 16 bits each.
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float4 drop_mul(unsigned long long base, unsigned long long q, unsigned thresh, float scale) {
  const unsigned long long u = mix64(base + (q + 1) * 0x9E3779B97F4A7C15ull);
  return make_float4(((unsigned)u & 0xFFFFu) >= thresh ? scale : 0.f, ((unsigned)(u >> 16) & 0xFFFFu) >= thresh ? scale : 0.f,
                     ((unsigned)(u >> 32) & 0xFFFFu) >= thresh ? scale : 0.f, (unsigned)(u >> 48) >= thresh ? scale : 0.f);
}

}  // namespace offk
