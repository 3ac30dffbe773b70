// K4: channels-last implicit-GEMM convolution on the fp32 matrix cores (gfx950).
//
// Stands for the fusion convolutions of the OFF sub-network (reference
// RGB_OFF.py:657-685 fusion@28, :762-780 fusion@14, :833-841 fusion@7): nn.Conv2d +
// bias with the surrounding ReLU / residual-add folded into the epilogue.
//
// GEMM view:  Y[m][co] = sum_k A[m][k] * Wt[co][k]
//   m  = (image, ho, wo) output pixel, k = (kh, kw, ci)  -- channels-last makes every
//   (pixel, tap) a contiguous run of Ci floats, so a 32-wide K tile is one tap and 32
//   consecutive channels (Ci % 32 == 0 for every fusion conv): one bounds check and one
//   16-B load per lane, no per-element im2col arithmetic.
//   M = pixels sits on the MFMA row axis, N = out-channels on the lane axis, so each
//   accumulator register stores 32 consecutive channels of one pixel (128-B runs).
#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

struct ConvArgs {
  const float* x; int x_cs, x_coff;
  int n_img, H, W, Ci, Ho, Wo, pad;
  const float* w;      // [Co][KH*KW*Ci]
  const float* bias;   // [Co]
  int Co;
  const float* res; int res_cs, res_coff;
  float* y; int y_cs, y_coff;
  int flags, M;
};

template <int KH, int KW, int S, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int RA = BM / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As0 = smem;                         // [2][BM][LDS_K]
  float* Bs0 = smem + 2 * BM * LDS_K;        // [2][BN][LDS_K]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int K = KH * KW * p.Ci;
  const int cpt = p.Ci / BK;                 // K tiles per tap
  const int nkt = KH * KW * cpt;

  // per-thread A rows: (tid>>3) + 32*r ; the pixel decode is loop invariant
  int hi0[RA], wi0[RA];
  size_t pix0[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    int m = m0 + (tid >> 3) + 32 * r;
    bool ok = m < p.M;
    int mm = ok ? m : 0;
    int img = mm / (p.Ho * p.Wo);
    int rem = mm - img * (p.Ho * p.Wo);
    int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    hi0[r] = ok ? ho * S - p.pad : -100000;  // pushes every tap out of bounds
    wi0[r] = wo * S - p.pad;
    pix0[r] = (size_t)img * p.H * p.W;
  }
  const float* xbase = p.x + p.x_coff + 4 * (tid & 7);
  const bool relu_in = p.flags & OFFK_CONV_RELU_IN_;
  // weight rows (tid>>3) + 32*r of this block's N slab, 16 B at k = 4*(tid&7)
  const float* wbase = p.w + (size_t)(n0 + (tid >> 3)) * K + 4 * (tid & 7);

  constexpr int RB = BN / 32;
  float4 rg[RA + RB];   // prefetch registers: A rows then B rows
  auto load_tile = [&](int kt) {
    int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
    int kh = tap / KW, kw = tap - kh * KW;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      int hi = hi0[r] + kh, wi = wi0[r] + kw;
      bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(xbase + (pix0[r] + (size_t)(hi * p.W + wi)) * p.x_cs + c0);
      rg[r] = relu_in ? relu4(v) : v;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
      rg[RA + r] = *reinterpret_cast<const float4*>(wbase + (size_t)32 * r * K + kt * BK);
  };
  auto store_tile = [&](int stage) {
    float* As = As0 + stage * BM * LDS_K;
#pragma unroll
    for (int r = 0; r < RA; ++r)
      *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[r];
    float* Bs = Bs0 + stage * BN * LDS_K;
#pragma unroll
    for (int r = 0; r < RB; ++r)
      *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[RA + r];
  };

  WaveAcc<TM, TN> acc;
  acc.zero();
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    acc.mma_ktile(As0 + st * BM * LDS_K + wm * (32 * TM) * LDS_K,
                  Bs0 + st * BN * LDS_K + wn * (32 * TN) * LDS_K, lane);
    if (kt + 1 < nkt) store_tile(st ^ 1);
    __syncthreads();
  }

  // epilogue: y = post( pre(acc + bias) + res )
  const int r32 = lane & 31, h = lane >> 5;
  const bool relu_pre = p.flags & OFFK_CONV_RELU_PRE_, relu_post = p.flags & OFFK_CONV_RELU_POST_;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int co = n0 + wn * 32 * TN + tn * 32 + r32;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int m = m0 + wm * 32 * TM + tm * 32 + acc_row(reg, h);
        if (m < p.M) {
          float v = acc.acc[tm][tn][reg] + bv;
          if (relu_pre) v = fmaxf(v, 0.f);
          if (p.res) v += p.res[(size_t)m * p.res_cs + p.res_coff + co];
          if (relu_post) v = fmaxf(v, 0.f);
          p.y[(size_t)m * p.y_cs + p.y_coff + co] = v;
        }
      }
    }
  }
}

template <int KH, int KW, int S, int TM, int TN, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  auto kern = conv_igemm_kernel<KH, KW, S, TM, TN, WM, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  dim3 grid((a.M + BM - 1) / BM, a.Co / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// Tile choice: Co % 128 == 0 -> 128x128 block (2x2 waves of 64x64); else 128x64.
template <int KH, int KW, int S>
static hipError_t launch_shape(const ConvArgs& a, hipStream_t st) {
  if (a.Co % 128 == 0) return launch_cfg<KH, KW, S, 2, 2, 2, 2>(a, st);
  return launch_cfg<KH, KW, S, 1, 2, 4, 1>(a, st);
}

hipError_t conv2d_launch(const ConvDesc& d, hipStream_t st, const char** why) {
  *why = nullptr;
  if (d.Ci % 32 || d.Co % 64 || d.x_cs % 4 || d.x_coff % 4 || d.y_cs <= 0) {
    *why = "conv2d: need Ci % 32 == 0, Co % 64 == 0, 16-byte aligned channel slices";
    return hipErrorInvalidValue;
  }
  ConvArgs a;
  a.x = d.x; a.x_cs = d.x_cs; a.x_coff = d.x_coff;
  a.n_img = d.n_img; a.H = d.H; a.W = d.W; a.Ci = d.Ci; a.pad = d.pad;
  a.Ho = (d.H + 2 * d.pad - d.KH) / d.stride + 1;
  a.Wo = (d.W + 2 * d.pad - d.KW) / d.stride + 1;
  a.w = d.w; a.bias = d.bias; a.Co = d.Co;
  a.res = d.res; a.res_cs = d.res_cs; a.res_coff = d.res_coff;
  a.y = d.y; a.y_cs = d.y_cs; a.y_coff = d.y_coff;
  a.flags = d.flags;
  long long M = (long long)d.n_img * a.Ho * a.Wo;
  if (M <= 0 || M > 0x7fffffffLL - 256) { *why = "conv2d: bad problem size"; return hipErrorInvalidValue; }
  a.M = (int)M;
  const int key = d.KH * 100 + d.KW * 10 + d.stride;
  switch (key) {
    case 111: return launch_shape<1, 1, 1>(a, st);
    case 331: return launch_shape<3, 3, 1>(a, st);
    case 552: return launch_shape<5, 5, 2>(a, st);
    case 772: return launch_shape<7, 7, 2>(a, st);
    default: *why = "conv2d: unsupported kernel/stride (have 1x1s1, 3x3s1, 5x5s2, 7x7s2)"; return hipErrorInvalidValue;
  }
}

// [Co][Ci][KH][KW] -> [Co][KH][KW][Ci]
__global__ void pack_oihw_to_ohwi(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int KHW) {
  size_t n = (size_t)Co * Ci * KHW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int ci = (int)(i % Ci);
    size_t t = i / Ci;
    int tap = (int)(t % KHW);
    int co = (int)(t / KHW);
    dst[i] = src[((size_t)co * Ci + ci) * KHW + tap];
  }
}

hipError_t pack_conv_weight_launch(const float* src, int Co, int Ci, int KH, int KW, float* dst, hipStream_t st) {
  size_t n = (size_t)Co * Ci * KH * KW;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_oihw_to_ohwi, dim3(blocks), dim3(256), 0, st, src, dst, Co, Ci, KH * KW);
  return hipGetLastError();
}

}  // namespace offk
