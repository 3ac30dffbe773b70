// K2: spatial gradient + temporal difference + concat, the HBM-bound half of an OFF unit.
//
// Stands for, per tap site (reference RGB_OFF.py, site 3a lines):
//   T[b*(L-1)+t] = G[b*L+t+1] - G[b*L+t]                     :599-604 (view / slice / sub)
//   S = depthwise3x3(D) (+bias)                               :611   (RGB_OFF: learned)
//     = SobelFilter_Diagonal(D)                               Flow_OFF.py:622, util.py:52-77
//   motion_s = cat(S, T); fusion = cat(motion_*, carried)     :616, :656, :760, :832
// The kernel writes [S | T] straight into channels [m_coff, m_coff+160) of the fusion
// buffer, so neither concat ever exists as a copy.  dropout(p=0.8) (:612) is the
// identity in eval mode.
//
// Layout: everything channels-last.  G [N*HW][128], D [P*HW][32], M [P*HW][m_cs].
// One block = (site, clip b, strip of `rows` image rows).  All nine sites share one
// grouped launch.
//   temporal: a thread owns (pixel, 4 channels) and walks t with the previous frame's
//             value kept in registers, so every G element is read exactly once with
//             16-B loads (1 KiB contiguous per wave instruction) -- algo 0; algo 1 puts
//             t on the lanes (8 t-slots x 8 channel quads) and takes the difference
//             with a wavefront shuffle (kept for the A/B measurement in DESIGN.md).
//   spatial : the D strip plus a one-pixel zero halo is staged in LDS per pair; a thread
//             owns a fixed channel quad (its 9 tap weights + bias live in registers) and
//             reads its 3x3 neighbourhood from LDS with ds_read_b128.
// Algorithmic HBM bytes per (clip, site): H*H*4*(128*L + 192*(L-1))  (SURVEY.md 8d).
#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

constexpr int ST_THREADS = 256;

void st_plan(int H, int* strips, int* rows) {
  // <= 196 pixels per block: 28x28 planes are cut into four 7-row strips
  *rows = H > 14 ? 7 : H;
  *strips = (H + *rows - 1) / *rows;
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 fma4(float4 w, float4 x, float4 acc) {
  return make_float4(fmaf(w.x, x.x, acc.x), fmaf(w.y, x.y, acc.y), fmaf(w.z, x.z, acc.z), fmaf(w.w, x.w, acc.w));
}

template <int ALGO>
__global__ __launch_bounds__(ST_THREADS) void sobel_tdiff_kernel(StParams p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [(rows+2)][(W+2)][32]

  // block -> site: field-wise scalar select chain (see pw_reduce.hip)
  StSite S;
  S.G = p.s[0].G; S.D = p.s[0].D; S.dw = p.s[0].dw; S.db = p.s[0].db; S.M = p.s[0].M; S.H = p.s[0].H;
  S.m_cs = p.s[0].m_cs; S.m_coff = p.s[0].m_coff; S.strips = p.s[0].strips; S.rows = p.s[0].rows;
  S.blk_begin = p.s[0].blk_begin;
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.x >= p.s[i].blk_begin) {
      S.G = p.s[i].G; S.D = p.s[i].D; S.dw = p.s[i].dw; S.db = p.s[i].db; S.M = p.s[i].M; S.H = p.s[i].H;
      S.m_cs = p.s[i].m_cs; S.m_coff = p.s[i].m_coff; S.strips = p.s[i].strips; S.rows = p.s[i].rows;
      S.blk_begin = p.s[i].blk_begin;
    }
  const int H = S.H, W = S.H, HW = H * H;
  const int local = (int)blockIdx.x - S.blk_begin;
  const int b = local / S.strips, strip = local - b * S.strips;
  const int y0 = strip * S.rows;
  const int R = min(S.rows, H - y0);
  const int L = p.L, T = L - 1;
  const int tid = threadIdx.x;
  const size_t f0 = (size_t)b * L;        // first frame of the clip
  const size_t p0 = (size_t)b * T;        // first pair of the clip
  const int npix = R * W, q0 = y0 * W;

  // ---------------- temporal difference: M[.., coff+32 .. coff+160) ----------------
  if (ALGO == 0) {
    for (int task = tid; task < npix * 32; task += ST_THREADS) {
      const int q = q0 + (task >> 5), c4 = (task & 31) * 4;
      const float* g = S.G + (f0 * HW + q) * kGenCh + c4;
      float* m = S.M + (p0 * HW + q) * S.m_cs + S.m_coff + kDownCh + c4;
      float4 prev = *reinterpret_cast<const float4*>(g);
      for (int t = 0; t < T; ++t) {
        g += (size_t)HW * kGenCh;
        float4 cur = *reinterpret_cast<const float4*>(g);
        *reinterpret_cast<float4*>(m) = sub4(cur, prev);
        m += (size_t)HW * S.m_cs;
        prev = cur;
      }
    }
  } else {
    // lanes: slot = lane>>3 (t within a group of 8), cq = lane&7; a wave covers 8 t x 32 channels
    // per step, the next group of t overlaps by one so every pair has both ends in one wave.
    const int lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 3, cq = lane & 7;
    for (int unit = wave; unit < npix * 4; unit += ST_THREADS / 64) {
      const int q = q0 + (unit >> 2), c4 = ((unit & 3) * 8 + cq) * 4;
      for (int tb = 0; tb < T; tb += 7) {
        const int t = tb + slot;                       // frame index within the clip
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < L) v = *reinterpret_cast<const float4*>(S.G + ((f0 + t) * HW + q) * kGenCh + c4);
        float4 nx;
        nx.x = __shfl_down(v.x, 8); nx.y = __shfl_down(v.y, 8);
        nx.z = __shfl_down(v.z, 8); nx.w = __shfl_down(v.w, 8);
        if (slot < 7 && t < T)
          *reinterpret_cast<float4*>(S.M + ((p0 + t) * HW + q) * S.m_cs + S.m_coff + kDownCh + c4) = sub4(nx, v);
      }
    }
  }

  // ---------------- spatial gradient: M[.., coff .. coff+32) ------------------------
  const int cq4 = (tid & 7) * 4;            // this thread's channel quad, fixed for the whole block
  float4 wt[9], bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const float4*>(S.dw + k * kDownCh + cq4);
  if (S.db) bias4 = *reinterpret_cast<const float4*>(S.db + cq4);
  const int TW = W + 2, TR = R + 2;
  for (int t = 0; t < T; ++t) {
    const float* d = S.D + (p0 + t) * HW * kDownCh;
    // stage rows y0-1 .. y0+R with zero halo; task = (tile pixel, channel quad)
    for (int task = tid; task < TR * TW * 8; task += ST_THREADS) {
      const int tp = task >> 3;
      const int ty = tp / TW, tx = tp - ty * TW;
      const int y = y0 - 1 + ty, x = tx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
        v = *reinterpret_cast<const float4*>(d + (size_t)(y * W + x) * kDownCh + cq4);
      *reinterpret_cast<float4*>(tile + tp * kDownCh + cq4) = v;
    }
    __syncthreads();
    float* mrow = S.M + (p0 + t) * HW * S.m_cs + S.m_coff + cq4;
    for (int task = tid; task < npix * 8; task += ST_THREADS) {
      const int px = task >> 3;
      const int r = px / W, x = px - r * W;
      const float* c = tile + ((r + 1) * TW + (x + 1)) * kDownCh + cq4;
      float4 acc = bias4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          acc = fma4(wt[dy * 3 + dx], *reinterpret_cast<const float4*>(c + ((dy - 1) * TW + (dx - 1)) * kDownCh), acc);
      *reinterpret_cast<float4*>(mrow + (size_t)(q0 + px) * S.m_cs) = acc;
    }
    __syncthreads();
  }
}

hipError_t sobel_tdiff_launch(const StParams& p, int algo, hipStream_t st) {
  if (p.total_blocks <= 0) return hipSuccess;
  int maxH = 0, rows = 0;
  for (int i = 0; i < p.nsites; ++i)
    if (p.s[i].H > maxH) { maxH = p.s[i].H; rows = p.s[i].rows; }
  size_t lds = (size_t)(rows + 2) * (maxH + 2) * kDownCh * sizeof(float);
  if (algo == 0) hipLaunchKernelGGL(sobel_tdiff_kernel<0>, dim3(p.total_blocks), dim3(ST_THREADS), lds, st, p);
  else hipLaunchKernelGGL(sobel_tdiff_kernel<1>, dim3(p.total_blocks), dim3(ST_THREADS), lds, st, p);
  return hipGetLastError();
}

// [32][1][3][3] -> [9][32]
__global__ void repack_dw_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  int i = threadIdx.x;  // 288 threads
  if (i < 9 * kDownCh) {
    int tap = i / kDownCh, c = i - tap * kDownCh;
    dst[i] = src[c * 9 + tap];
  }
}
hipError_t repack_dw_launch(const float* w, float* out, hipStream_t st) {
  hipLaunchKernelGGL(repack_dw_kernel, dim3(1), dim3(320), 0, st, w, out);
  return hipGetLastError();
}

}  // namespace offk
