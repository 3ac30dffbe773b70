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
constexpr int ST_STAGE_MAX = 6;   // >= ceil((rows+2)*(W+2)*8 / 256) for every plan below
constexpr int ST_OUT_MAX = 4;     // >= ceil(rows*W*8 / 256): output pixels per thread
constexpr int ST_TGROUP = 6;      // frames loaded per temporal step (L = 7 -> one step)

void st_plan(int H, int* strips, int* rows) {
  // 98..112 pixels per block: fine-grained enough that the 1664 blocks at B = 64 balance over
  // the CUs, coarse enough that the halo rows re-read by the neighbouring strip stay < 50 % of D
  *rows = H >= 28 ? 4 : 7;
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
  const size_t gstride = (size_t)HW * kGenCh, mstride = (size_t)HW * S.m_cs;

  // ---------------- temporal difference: M[.., coff+32 .. coff+160) ----------------
  if (ALGO == 0) {
    // A thread owns (pixel, 4 channels); up to ST_TGROUP frames are in flight per step (all of
    // them for L = 7), the previous frame's value stays in registers: each G element is read
    // exactly once, 16 B per lane, 1 KiB contiguous per wave instruction.
    for (int task = tid; task < npix * 32; task += ST_THREADS) {
      const int q = q0 + (task >> 5), c4 = (task & 31) * 4;
      const float* g = S.G + (f0 * HW + q) * kGenCh + c4;
      float* m = S.M + (p0 * HW + q) * S.m_cs + S.m_coff + kDownCh + c4;
      float4 prev = *reinterpret_cast<const float4*>(g);
      for (int t0 = 1; t0 < L; t0 += ST_TGROUP) {
        float4 v[ST_TGROUP];
#pragma unroll
        for (int j = 0; j < ST_TGROUP; ++j)
          if (t0 + j < L) v[j] = *reinterpret_cast<const float4*>(g + (size_t)(t0 + j) * gstride);
#pragma unroll
        for (int j = 0; j < ST_TGROUP; ++j)
          if (t0 + j < L) *reinterpret_cast<float4*>(m + (size_t)(t0 + j - 1) * mstride) = sub4(v[j], j ? v[j - 1] : prev);
        prev = v[ST_TGROUP - 1];
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
  // tap weights [9][32] + bias [32] sit behind the tile in LDS (keeps ~40 VGPRs free for loads in flight)
  float* wl = tile + (S.rows + 2) * (W + 2) * kDownCh;
  for (int i = tid; i < 10 * kDownCh; i += ST_THREADS)
    wl[i] = i < 9 * kDownCh ? S.dw[i] : (S.db ? S.db[i - 9 * kDownCh] : 0.f);
  const int TW = W + 2, TR = R + 2;
  const int nstage = TR * TW;              // tile pixels; this thread stages pixels (tid>>3) + 32*j
  // source offset (in floats, within one pair's D plane) of each staged pixel, or -1 for the zero halo
  int soff[ST_STAGE_MAX];
#pragma unroll
  for (int j = 0; j < ST_STAGE_MAX; ++j) {
    const int tp = (tid >> 3) + 32 * j;
    const int ty = tp / TW, tx = tp - ty * TW;
    const int y = y0 - 1 + ty, x = tx - 1;
    soff[j] = (tp < nstage && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (y * W + x) * kDownCh + cq4 : -1;
  }
  // LDS offset of the centre tap of each output pixel (tid>>3) + 32*j of the strip, or -1
  int coff[ST_OUT_MAX];
#pragma unroll
  for (int j = 0; j < ST_OUT_MAX; ++j) {
    const int px = (tid >> 3) + 32 * j;
    const int r = px / W, x = px - r * W;
    coff[j] = px < npix ? ((r + 1) * TW + (x + 1)) * kDownCh + cq4 : -1;
  }
  float4 st[ST_STAGE_MAX];
  auto load_pair = [&](int t) {
    const float* d = S.D + (p0 + t) * HW * kDownCh;
#pragma unroll
    for (int j = 0; j < ST_STAGE_MAX; ++j) {
      st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (soff[j] >= 0) st[j] = *reinterpret_cast<const float4*>(d + soff[j]);
    }
  };
  load_pair(0);
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int j = 0; j < ST_STAGE_MAX; ++j) {
      const int tp = (tid >> 3) + 32 * j;
      if (tp < nstage) *reinterpret_cast<float4*>(tile + tp * kDownCh + cq4) = st[j];
    }
    __syncthreads();
    if (t + 1 < T) load_pair(t + 1);       // next pair's strip is in flight while this one is filtered
    // taps outer, this thread's (<= ST_OUT_MAX) output pixels inner: one weight quad live at a time
    float* mrow = S.M + (p0 + t) * HW * S.m_cs + S.m_coff + cq4;
    float4 acc[ST_OUT_MAX];
    const float4 b4 = *reinterpret_cast<const float4*>(wl + 9 * kDownCh + cq4);
#pragma unroll
    for (int j = 0; j < ST_OUT_MAX; ++j) acc[j] = b4;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 w4 = *reinterpret_cast<const float4*>(wl + (dy * 3 + dx) * kDownCh + cq4);
        const int toff = ((dy - 1) * TW + (dx - 1)) * kDownCh;
#pragma unroll
        for (int j = 0; j < ST_OUT_MAX; ++j)
          if (coff[j] >= 0) acc[j] = fma4(w4, *reinterpret_cast<const float4*>(tile + coff[j] + toff), acc[j]);
      }
#pragma unroll
    for (int j = 0; j < ST_OUT_MAX; ++j)
      if (coff[j] >= 0) *reinterpret_cast<float4*>(mrow + (size_t)(q0 + (tid >> 3) + 32 * j) * S.m_cs) = acc[j];
    __syncthreads();
  }
}

hipError_t sobel_tdiff_launch(const StParams& p, int algo, hipStream_t st) {
  if (p.total_blocks <= 0) return hipSuccess;
  size_t tile_px = 0;
  for (int i = 0; i < p.nsites; ++i) {
    const size_t px = (size_t)(p.s[i].rows + 2) * (p.s[i].H + 2);
    if (px > tile_px) tile_px = px;
    if (px > 32 * ST_STAGE_MAX || p.s[i].rows * p.s[i].H > 32 * ST_OUT_MAX) return hipErrorInvalidValue;
  }
  size_t lds = (tile_px + 10) * kDownCh * sizeof(float);   // tile + taps + bias
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
