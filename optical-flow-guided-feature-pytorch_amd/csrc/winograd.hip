// K4w: Winograd for the 3x3 / stride 1 / pad 1 fusion convolutions on 7x7 maps (fp32 arithmetic throughout).
//
// Stands for motion_conv3_trans_14b (128 -> 512, RGB_OFF.py:777-780), motion_conv_trans (832 -> 256, :833-834),
// motion_conv2_trans (256 -> 256, :837-838) and the two 128 -> 128 convs of fusion@14: 1.07 ms of the 5.2 ms forward as direct
// implicit GEMMs at 74-80 % of the fp32 matrix peak.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A per output tile (Lavin & Gray,
// arXiv:1509.09308).  A 7-wide axis is cut 7 = 4 + 3: tile 0 is F(4, 3) (six points 0, +-1, +-2, infinity; outputs 0..3 from the
// samples -1..4), tile 1 is F(3, 3) (five points 0, +-1, 2, infinity; outputs 4..6 from the samples 3..7), so a map is four tiles of
// the classes cls = 2 cy + cx with 6x6, 6x5, 5x6 and 5x5 points: 121 multiplies per 49 outputs instead of 441 (3.64 x), and no
// output is computed that is not kept (four F(4x4, 3x3) tiles -- the first version of this file -- spend 144 for 64 outputs).
// The contraction becomes 121 independent GEMMs [images][Ci] x [Ci][Co] -- batch b = first(cls) + i * nx + j -- that run on the
// existing 1x1 implicit-GEMM kernel with a batch index (conv_igemm.hip), between two bandwidth-bound transform kernels:
//   wino_input_kernel    V[b][img][ci] = (B^T d B)[i][j]   a wave = 64 channels of one (class, image, phase)
//   (121 batched GEMMs)  M[b][img][co] = sum_ci V[b][img][ci] * U[b][co][ci]
//   wino_output_kernel   Y = A^T M A + bias, the conv epilogue (ReLU / residual / ReLU); optional per-tile column sums of the
//                        stored values (the 14-head's average pool, heads.hip fc_pooled)
//   wino_weight_kernel   U[b][co][ci] = (G g G^T)[i][j] for all four classes, once per weight change
// The 5x5 / stride 2 / pad 2 conv motion_conv_trans_14 (1056 -> 128 on 14x14, :762-763; 0.96 ms direct) goes the same way in
// polyphase form: its four phase images are 7x7, its four phase kernels at most 3x3, and their contractions concatenate along
// K (4 x 1056 = 4224), so the same GEMMs, output transform and transform matrices serve it (NPH = 4).
// Same arithmetic type as the reference (fp32), different summation: the transforms' constants (up to 8 and 1/24) cost a few
// ulps -- measured against the oracle in tests/test_gpu_parity.py (stage tensors and logits), well inside the 1e-3 budget;
// OFFK_WINOGRAD=0 at offk_create keeps the direct kernels.
#include "offk_common.h"
#include "offk_internal.h"
#include "winograd_common.h"

namespace offk {

namespace {
// one (class, image, phase) for 64 channels: load the NY x NX window, transform, store
template <int NPH, int CY, int CX>
__device__ __forceinline__ void wino_input_tile(const float* __restrict__ xi, int x_cs, int R, int img, int Ci, int pa, int pb,
                                                unsigned c, float* __restrict__ V) {
  constexpr int NY = CY ? 5 : 6, NX = CX ? 5 : 6, W = NPH == 1 ? 7 : 14, STEP = NPH == 1 ? 1 : 2;
  constexpr int oy = CY ? 4 : 0, ox = CX ? 4 : 0;          // output origin; the window starts one sample up / left of it
  const unsigned c4 = c * 4u;
  float t[NY][NX];
#pragma unroll
  for (int j = 0; j < NX; ++j)
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int row = oy - 1 + i, col = ox - 1 + j;          // (compile-time: which samples are the zero padding)
      t[i][j] = row >= 0 && row < 7 && col >= 0 && col < 7
                    ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xi + (size_t)((STEP * row + pa) * W + STEP * col + pb) * x_cs) + c4)
                    : 0.f;
    }
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    float d[NY], tc[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) d[i] = t[i][j];
    wbt(d, tc);
#pragma unroll
    for (int i = 0; i < NY; ++i) t[i][j] = tc[i];
  }
  const int K = NPH * Ci;
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    float v[NX];
    wbt(t[i], v);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      if constexpr (NPH == 1) {
        *reinterpret_cast<float*>(reinterpret_cast<char*>(V + ((size_t)wino_mindex<1, CY, CX>(i, j) * R + img) * K) + c4) = v[j];
      } else {
        WinoPlace pl;
        if (wino_place4<CY, CX>(i, j, pa, pb, pl))     // V of the polyphase conv (0.63 GB at P = 384) is written once and read once, by
                                                       // the GEMMs: non-temporal (the 7x7-map convs' V fits the caches and stays temporal)
          __builtin_nontemporal_store(v[j], reinterpret_cast<float*>(reinterpret_cast<char*>(V + ((size_t)pl.base_ci * R + ((size_t)pl.idx * R + img) * pl.kmul + pl.slot) * Ci) + c4));
      }
    }
  }
}
}  // namespace

// x: channels-last [n_img * W * W][x_cs], Ci channels at x_coff.  V: NPH = 1: [121][n_img][Ci]; NPH = 4: the four K groups.
// NPH = 1: W = 7, the 3x3 / stride 1 conv.  NPH = 4: W = 14, the polyphase form of a 5x5 / stride 2 / pad 2 conv
//   y[o] = sum_{a, b in {0, 1}} conv3x3_pad1(X_ab, G_ab)[o],   X_ab[i][j] = x[2i + a][2j + b] (7x7),   G_ab[u][v] = w[2u + a][2v + b]
// (taps with 2u + a > 4 are zero): four 3x3 / stride 1 convs on 7x7 phase images whose contraction is concatenated along K
// (phase ph = 2a + b occupies K range [ph * Ci, ph * Ci + Ci)), so the SAME GEMMs and the same output transform serve it.
// A wave = 64 consecutive channels of one (class, image, phase): everything but the channel is wave-uniform.
template <int NPH>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int x_cs, int x_coff, int n_img, int Ci,
                                                         float* __restrict__ V) {
  constexpr int W = NPH == 1 ? 7 : 14;
  const int cw = (Ci + 63) >> 6;
  const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wid >= 4 * n_img * NPH * cw) return;
  // image-major: the four classes (and phases) of an image read overlapping windows -- close in time, the second read hits L2
  const int img = wid / (4 * NPH * cw), r0 = wid - img * (4 * NPH * cw);
  const int cls = r0 / (NPH * cw), r1 = r0 - cls * (NPH * cw), ph = r1 / cw, pa = ph >> 1, pb = ph & 1;
  const unsigned c = (unsigned)(r1 - ph * cw) * 64u + (threadIdx.x & 63u);
  if (c >= (unsigned)Ci) return;
  const float* xi = x + (size_t)img * W * W * x_cs + x_coff;
  switch (cls) {
    case 0: wino_input_tile<NPH, 0, 0>(xi, x_cs, n_img, img, Ci, pa, pb, c, V); break;
    case 1: wino_input_tile<NPH, 0, 1>(xi, x_cs, n_img, img, Ci, pa, pb, c, V); break;
    case 2: wino_input_tile<NPH, 1, 0>(xi, x_cs, n_img, img, Ci, pa, pb, c, V); break;
    default: wino_input_tile<NPH, 1, 1>(xi, x_cs, n_img, img, Ci, pa, pb, c, V); break;
  }
}

struct WinoOutArgs {
  const float* M;                // [121][n_img][Co]
  const float* bias;             // [Co]
  const float* res; int res_cs, res_coff;
  float* y; int y_cs, y_coff;
  float* pool_part;              // != nullptr: [n_img * 4][Co] sums of the stored values of each tile (row = img * 4 + cls)
  int n_img, Co, flags;
};
namespace {
template <int NPH, int CY, int CX>
__device__ __forceinline__ void wino_output_tile(const WinoOutArgs& a, int img, unsigned c) {
  constexpr int NY = CY ? 5 : 6, NX = CX ? 5 : 6, OY = CY ? 3 : 4, OX = CX ? 3 : 4, oy = CY ? 4 : 0, ox = CX ? 4 : 0;
  const float* mp = a.M + (size_t)img * a.Co + c;
  const size_t pstride = (size_t)a.n_img * a.Co;
  float s[OY][NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    float m[NY], sc[OY];
#pragma unroll
    for (int i = 0; i < NY; ++i) m[i] = mp[(size_t)wino_mindex<NPH, CY, CX>(i, j) * pstride];
    wat(m, sc);
#pragma unroll
    for (int i = 0; i < OY; ++i) s[i][j] = sc[i];
  }
  const float bv = a.bias ? a.bias[c] : 0.f;
  const bool relu_pre = a.flags & OFFK_CONV_RELU_PRE_, relu_post = a.flags & OFFK_CONV_RELU_POST_;
  float psum = 0.f;
#pragma unroll
  for (int i = 0; i < OY; ++i) {
    float yv[OX];
    wat(s[i], yv);
#pragma unroll
    for (int j = 0; j < OX; ++j) {
      const size_t pix = (size_t)img * 49 + (oy + i) * 7 + ox + j;
      float v = yv[j] + bv;
      if (relu_pre) v = fmaxf(v, 0.f);
      if (a.res) v += a.res[pix * a.res_cs + a.res_coff + c];
      if (relu_post) v = fmaxf(v, 0.f);
      a.y[pix * a.y_cs + a.y_coff + c] = v;
      psum += v;
    }
  }
  if (a.pool_part) a.pool_part[((size_t)img * 4 + 2 * CY + CX) * a.Co + c] = psum;
}
}  // namespace
template <int NPH>
__global__ __launch_bounds__(256) void wino_output_kernel(WinoOutArgs a) {
  const int cw = (a.Co + 63) >> 6;
  const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wid >= 4 * a.n_img * cw) return;
  // class-major (measured against image-major: 33 vs 39 us for 128 -> 512): consecutive waves read consecutive rows of the same M planes
  const int cls = wid / (a.n_img * cw), r0 = wid - cls * (a.n_img * cw), img = r0 / cw;
  const unsigned c = (unsigned)(r0 - img * cw) * 64u + (threadIdx.x & 63u);
  if (c >= (unsigned)a.Co) return;
  switch (cls) {
    case 0: wino_output_tile<NPH, 0, 0>(a, img, c); break;
    case 1: wino_output_tile<NPH, 0, 1>(a, img, c); break;
    case 2: wino_output_tile<NPH, 1, 0>(a, img, c); break;
    default: wino_output_tile<NPH, 1, 1>(a, img, c); break;
  }
}

// w: the library's packed weight [Co][Ci / 32][KS * KS][32] (pack_conv_weight_launch).  U: NPH = 1: [121][Co][Ci]; NPH = 4: the four K groups.
// NPH = 1: KS = 3.  NPH = 4: KS = 5, phase (a, b) takes the taps (2u + a, 2v + b), u, v in 0..2 (zero beyond tap 4).
namespace {
template <int NPH, int CY, int CX>
__device__ __forceinline__ void wino_weight_tile(const float (&g)[3][3], int Co, int co, int Ci, int ci, int pa, int pb, int kk,
                                                 float* __restrict__ U) {
  constexpr int NY = CY ? 5 : 6, NX = CX ? 5 : 6;
  float t[NY][3];
#pragma unroll
  for (int v = 0; v < 3; ++v) {            // column v of the (phase's) 3x3 kernel
    float gc[3], u[NY];
#pragma unroll
    for (int i = 0; i < 3; ++i) gc[i] = g[i][v];
    wg(gc, u);
#pragma unroll
    for (int i = 0; i < NY; ++i) t[i][v] = u[i];
  }
  const int K = NPH * Ci;
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    float u[NX];
    wg(t[i], u);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      if constexpr (NPH == 1) {
        U[((size_t)wino_mindex<1, CY, CX>(i, j) * Co + co) * K + kk] = u[j];
      } else {
        WinoPlace pl;
        if (wino_place4<CY, CX>(i, j, pa, pb, pl))
          U[((size_t)pl.base_ci * Co + ((size_t)pl.idx * Co + co) * pl.kmul + pl.slot) * Ci + ci] = u[j];
      }
    }
  }
}
}  // namespace
template <int NPH>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int Co, int Ci, float* __restrict__ U) {
  constexpr int KS = NPH == 1 ? 3 : 5, STEP = NPH == 1 ? 1 : 2;
  const int K = NPH * Ci;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)Co * K) return;
  const int co = (int)(gid / K), kk = (int)(gid - (long long)co * K);
  const int ph = kk / Ci, ci = kk - ph * Ci, pa = ph >> 1, pb = ph & 1;
  const float* wp = w + ((size_t)(co * (Ci / 32) + ci / 32) * (KS * KS)) * 32 + (ci & 31);
  float g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kh = STEP * i + pa, kw = STEP * j + pb;
      g[i][j] = kh < KS && kw < KS ? wp[(size_t)(kh * KS + kw) * 32] : 0.f;
    }
  wino_weight_tile<NPH, 0, 0>(g, Co, co, Ci, ci, pa, pb, kk, U);
  wino_weight_tile<NPH, 0, 1>(g, Co, co, Ci, ci, pa, pb, kk, U);
  wino_weight_tile<NPH, 1, 0>(g, Co, co, Ci, ci, pa, pb, kk, U);
  wino_weight_tile<NPH, 1, 1>(g, Co, co, Ci, ci, pa, pb, kk, U);
}

// rows = images per batch entry
int wino_groups(int phases, long long rows, int Ci, int Co, WinoGroup out[4]) {
  if (phases == 1) { out[0] = WinoGroup{kWinoPoints, 1, 0, 0, 0}; return 1; }
  const int batch[4] = {81, 18, 18, 4}, kmul[4] = {4, 2, 2, 1}, base[4] = {0, 324, 360, 396}, first[4] = {0, 81, 99, 117};
  for (int g = 0; g < 4; ++g)
    out[g] = WinoGroup{batch[g], kmul[g], (long long)base[g] * rows * Ci, (long long)base[g] * Co * Ci, (long long)first[g] * rows * Co};
  return 4;
}

hipError_t wino_weight_launch(const float* w_packed, int Co, int Ci, int phases, float* U, hipStream_t st) {
  const long long n = (long long)Co * Ci * phases;
  if (phases == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_packed, Co, Ci, U);
  else hipLaunchKernelGGL(wino_weight_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_packed, Co, Ci, U);
  return hipGetLastError();
}

hipError_t wino_input_launch(const float* x, int x_cs, int x_coff, int n_img, int Ci, int phases, float* V, hipStream_t st) {
  const long long waves = (long long)4 * n_img * phases * ((Ci + 63) / 64);
  if (phases == 4) hipLaunchKernelGGL(wino_input_kernel<4>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, x_cs, x_coff, n_img, Ci, V);
  else hipLaunchKernelGGL(wino_input_kernel<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, x_cs, x_coff, n_img, Ci, V);
  return hipGetLastError();
}

hipError_t wino_output_launch(const float* M, int n_img, int Co, int phases, const float* bias, const float* res, int res_cs,
                              int res_coff, int flags, float* y, int y_cs, int y_coff, float* pool_part, hipStream_t st) {
  WinoOutArgs a;
  a.M = M; a.bias = bias; a.res = res; a.res_cs = res_cs; a.res_coff = res_coff;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.pool_part = pool_part;
  a.n_img = n_img; a.Co = Co; a.flags = flags;
  const long long waves = (long long)4 * n_img * ((Co + 63) / 64);
  if (phases == 4) hipLaunchKernelGGL(wino_output_kernel<4>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wino_output_kernel<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace offk
