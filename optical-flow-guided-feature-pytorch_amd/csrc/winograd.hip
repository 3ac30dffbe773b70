// K4w: Winograd F(4x4, 3x3) for the 3x3 / stride 1 / pad 1 fusion convolutions on 7x7 maps (fp32 arithmetic throughout).
//
// Stands for motion_conv3_trans_14b (128 -> 512, RGB_OFF.py:777-780), motion_conv_trans (832 -> 256, :833-834) and
// motion_conv2_trans (256 -> 256, :837-838): 0.95 ms of the 5.2 ms forward as direct implicit GEMMs at 74-80 % of the fp32
// matrix peak.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 4x4 output tile (Lavin & Gray, arXiv:1509.09308): 36 multiplies
// per 16 outputs instead of 144.  A 7x7 map is 2 x 2 tiles (8x8 outputs computed, 49 kept: 64 / 49 of the work), so the
// contraction shrinks 3.06x; it becomes 36 independent GEMMs [tiles][Ci] x [Ci][Co] that run on the existing 1x1
// implicit-GEMM kernel with a batch index (conv_igemm.hip), between two bandwidth-bound transform kernels:
//   wino_input_kernel    V[p][tile][ci] = (B^T d B)[p]   one thread per (tile, channel), lanes along channels
//   (36 batched GEMMs)   M[p][tile][co] = sum_ci V[p][tile][ci] * U[p][co][ci]
//   wino_output_kernel   Y = A^T M A + bias, the conv epilogue (ReLU / residual / ReLU), the valid 7x7 pixels stored; optional
//                        per-tile column sums of the stored values (the 14-head's average pool, heads.hip fc_pooled)
//   wino_weight_kernel   U[p][co][ci] = (G g G^T)[p], once per weight change
// The 5x5 / stride 2 / pad 2 conv motion_conv_trans_14 (1056 -> 128 on 14x14, :762-763; 0.96 ms direct) goes the same way in
// polyphase form: its four phase images are 7x7, its four phase kernels at most 3x3, and their contractions concatenate along
// K (4 x 1056 = 4224), so the 36 GEMMs, the output transform and this file's transform matrices serve it unchanged (NPH = 4).
// Same arithmetic type as the reference (fp32), different summation: the transforms' constants (up to 8 and 1/24) cost a few
// ulps -- measured against the oracle in tests/test_gpu_parity.py (stage tensors and logits), well inside the 1e-3 budget;
// OFFK_WINOGRAD=0 at offk_create keeps the direct kernels.
#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
// 1-D input transform B^T (6 -> 6)
__device__ __forceinline__ void bt6(const float (&d)[6], float (&t)[6]) {
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// 1-D output transform A^T (6 -> 4)
__device__ __forceinline__ void at4(const float (&m)[6], float (&s)[4]) {
  const float a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
  s[0] = m[0] + a + c;
  s[1] = b + 2.f * e;
  s[2] = a + 4.f * c;
  s[3] = b + 8.f * e + m[5];
}
// 1-D weight transform G (3 -> 6)
__device__ __forceinline__ void g6(const float (&g)[3], float (&u)[6]) {
  u[0] = g[0] * 0.25f;
  u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
  u[2] = (-g[0] + g[1] - g[2]) * (1.f / 6.f);
  u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[5] = g[2];
}
// Polyphase form (NPH = 4): a phase kernel with a == 1 (b == 1) has two taps, its transform is zero at the point "infinity"
// (row / column 5 of the 6x6 point grid).  Those (point, phase) products are skipped: the 25 points with i, j < 5 contract all
// four phases (K = 4 Ci), the five points of row 5 only the phases with a == 0, the five of column 5 those with b == 0 (K = 2 Ci),
// point (5, 5) phase 0 alone (K = Ci): 121 instead of 144 phase-points of GEMM work and of transformed-input traffic.  Points are
// stored group by group -- A: i * 5 + j, B (i == 5): 25 + j, C (j == 5): 30 + i, D: 35 -- each group with its own K.
// place(): for point (i, j) and phase (pa, pb), the group's first float (in units of rows * Ci), the point's index inside the
// group, the group's K in units of Ci and the phase's slot in it; false if the product is identically zero.
struct WinoPlace { int base_ci, idx, kmul, slot; };
__device__ __forceinline__ bool wino_place4(int i, int j, int pa, int pb, WinoPlace& o) {
  if (i < 5 && j < 5) { o = WinoPlace{0, i * 5 + j, 4, 2 * pa + pb}; return true; }
  if (i == 5 && j < 5) { o = WinoPlace{100, j, 2, pb}; return pa == 0; }             // group A holds 25 * 4 = 100 row-Ci units
  if (j == 5 && i < 5) { o = WinoPlace{110, i, 2, pa}; return pb == 0; }             // + 5 * 2
  o = WinoPlace{120, 0, 1, 0};                                                         // + 5 * 2
  return pa == 0 && pb == 0;
}
// index of point (i, j) in the GEMM-output array M [36][T][Co]
__device__ __forceinline__ int wino_mindex(int nph, int i, int j) {
  if (nph == 1) return i * 6 + j;
  return i < 5 && j < 5 ? i * 5 + j : (i == 5 && j < 5 ? 25 + j : (j == 5 && i < 5 ? 30 + i : 35));
}
}  // namespace

// x: channels-last [n_img * W * W][x_cs], Ci channels at x_coff.  V: [36][n_img * 4][NPH * Ci].
// NPH = 1: W = 7, the 3x3 / stride 1 conv.  NPH = 4: W = 14, the polyphase form of a 5x5 / stride 2 / pad 2 conv
//   y[o] = sum_{a, b in {0, 1}} conv3x3_pad1(X_ab, G_ab)[o],   X_ab[i][j] = x[2i + a][2j + b] (7x7),   G_ab[u][v] = w[2u + a][2v + b]
// (taps with 2u + a > 4 are zero): four 3x3 / stride 1 convs on 7x7 phase images whose contraction is concatenated along K
// (phase ph = 2a + b occupies K range [ph * Ci, ph * Ci + Ci)), so the SAME 36 GEMMs and the same output transform serve it.
template <int NPH>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int x_cs, int x_coff, int n_img, int Ci,
                                                         float* __restrict__ V) {
  constexpr int W = NPH == 1 ? 7 : 14, STEP = NPH == 1 ? 1 : 2;
  const int T = n_img * 4, K = NPH * Ci;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)T * K) return;
  const int tile = (int)(gid / K), kk = (int)(gid - (long long)tile * K);
  const int ph = kk / Ci, c = kk - ph * Ci, pa = ph >> 1, pb = ph & 1;
  const int img = tile >> 2, oy = (tile & 2) * 2, ox = (tile & 1) * 4;      // output tile origin (0 / 4)
  const float* xi = x + (size_t)img * W * W * x_cs + x_coff + c;
  float t[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {            // column j of the 6x6 input tile: rows oy - 1 .. oy + 4, column ox - 1 + j (of the phase image)
    const int col = ox - 1 + j;
    float d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = oy - 1 + i;
      const bool ok = (unsigned)row < 7u && (unsigned)col < 7u;
      d[i] = ok ? xi[(size_t)((STEP * row + pa) * W + STEP * col + pb) * x_cs] : 0.f;
    }
    float tc[6];
    bt6(d, tc);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = tc[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float v[6];
    bt6(t[i], v);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if constexpr (NPH == 1) {
        V[((size_t)(i * 6 + j) * T + tile) * K + kk] = v[j];
      } else {
        WinoPlace pl;
        if (wino_place4(i, j, pa, pb, pl))
          V[((size_t)pl.base_ci * T + ((size_t)pl.idx * T + tile) * pl.kmul + pl.slot) * Ci + c] = v[j];
      }
    }
  }
}

struct WinoOutArgs {
  const float* M;                // [36][T][Co]
  const float* bias;             // [Co]
  const float* res; int res_cs, res_coff;
  float* y; int y_cs, y_coff;
  float* pool_part;              // != nullptr: [T][Co] sums of the stored values of each tile (4 tiles = one image)
  int n_img, Co, flags;
  int nph;                       // 1: M indexed i * 6 + j; 4: the polyphase point order (wino_mindex)
};
__global__ __launch_bounds__(256) void wino_output_kernel(WinoOutArgs a) {
  const int T = a.n_img * 4;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)T * a.Co) return;
  const int tile = (int)(gid / a.Co), c = (int)(gid - (long long)tile * a.Co);
  const int img = tile >> 2, oy = (tile & 2) * 2, ox = (tile & 1) * 4;
  const float* mp = a.M + (size_t)tile * a.Co + c;
  const size_t pstride = (size_t)T * a.Co;
  float s[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = mp[(size_t)wino_mindex(a.nph, i, j) * pstride];
    float sc[4];
    at4(m, sc);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i][j] = sc[i];
  }
  const float bv = a.bias ? a.bias[c] : 0.f;
  const bool relu_pre = a.flags & OFFK_CONV_RELU_PRE_, relu_post = a.flags & OFFK_CONV_RELU_POST_;
  float psum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float yv[4];
    at4(s[i], yv);
    const int row = oy + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = ox + j;
      if (row < 7 && col < 7) {
        const size_t pix = (size_t)img * 49 + row * 7 + col;
        float v = yv[j] + bv;
        if (relu_pre) v = fmaxf(v, 0.f);
        if (a.res) v += a.res[pix * a.res_cs + a.res_coff + c];
        if (relu_post) v = fmaxf(v, 0.f);
        a.y[pix * a.y_cs + a.y_coff + c] = v;
        psum += v;
      }
    }
  }
  if (a.pool_part) a.pool_part[(size_t)tile * a.Co + c] = psum;
}

// w: the library's packed weight [Co][Ci / 32][KS * KS][32] (pack_conv_weight_launch).  U: [36][Co][NPH * Ci].
// NPH = 1: KS = 3.  NPH = 4: KS = 5, phase (a, b) takes the taps (2u + a, 2v + b), u, v in 0..2 (zero beyond tap 4).
template <int NPH>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int Co, int Ci, float* __restrict__ U) {
  constexpr int KS = NPH == 1 ? 3 : 5, STEP = NPH == 1 ? 1 : 2;
  const int K = NPH * Ci;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)Co * K) return;
  const int co = (int)(gid / K), kk = (int)(gid - (long long)co * K);
  const int ph = kk / Ci, ci = kk - ph * Ci, pa = ph >> 1, pb = ph & 1;
  const float* wp = w + ((size_t)(co * (Ci / 32) + ci / 32) * (KS * KS)) * 32 + (ci & 31);
  float t[6][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {            // column j of the phase's 3x3 kernel
    float g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int kh = STEP * i + pa, kw = STEP * j + pb;
      g[i] = kh < KS && kw < KS ? wp[(size_t)(kh * KS + kw) * 32] : 0.f;
    }
    float u[6];
    g6(g, u);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = u[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float u[6];
    g6(t[i], u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if constexpr (NPH == 1) {
        U[((size_t)(i * 6 + j) * Co + co) * K + kk] = u[j];
      } else {
        WinoPlace pl;
        if (wino_place4(i, j, pa, pb, pl))
          U[((size_t)pl.base_ci * Co + ((size_t)pl.idx * Co + co) * pl.kmul + pl.slot) * Ci + ci] = u[j];
      }
    }
  }
}

int wino_groups(int phases, long long T, int Ci, int Co, WinoGroup out[4]) {
  if (phases == 1) { out[0] = WinoGroup{36, 1, 0, 0, 0}; return 1; }
  const int batch[4] = {25, 5, 5, 1}, kmul[4] = {4, 2, 2, 1}, base[4] = {0, 100, 110, 120}, first[4] = {0, 25, 30, 35};
  for (int g = 0; g < 4; ++g)
    out[g] = WinoGroup{batch[g], kmul[g], (long long)base[g] * T * Ci, (long long)base[g] * Co * Ci, (long long)first[g] * T * Co};
  return 4;
}

hipError_t wino_weight_launch(const float* w_packed, int Co, int Ci, int phases, float* U, hipStream_t st) {
  const long long n = (long long)Co * Ci * phases;
  if (phases == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_packed, Co, Ci, U);
  else hipLaunchKernelGGL(wino_weight_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_packed, Co, Ci, U);
  return hipGetLastError();
}

hipError_t wino_input_launch(const float* x, int x_cs, int x_coff, int n_img, int Ci, int phases, float* V, hipStream_t st) {
  const long long n = (long long)n_img * 4 * Ci * phases;
  if (phases == 4) hipLaunchKernelGGL(wino_input_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, x_cs, x_coff, n_img, Ci, V);
  else hipLaunchKernelGGL(wino_input_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, x_cs, x_coff, n_img, Ci, V);
  return hipGetLastError();
}

hipError_t wino_output_launch(const float* M, int n_img, int Co, int phases, const float* bias, const float* res, int res_cs,
                              int res_coff, int flags, float* y, int y_cs, int y_coff, float* pool_part, hipStream_t st) {
  WinoOutArgs a;
  a.nph = phases;
  a.M = M; a.bias = bias; a.res = res; a.res_cs = res_cs; a.res_coff = res_coff;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.pool_part = pool_part;
  a.n_img = n_img; a.Co = Co; a.flags = flags;
  const long long n = (long long)n_img * 4 * Co;
  hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace offk
