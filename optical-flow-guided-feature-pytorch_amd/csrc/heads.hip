// K5 / K6 and the NCHW <-> channels-last helpers.
//
// K5 stands for the three classifier heads (reference RGB_OFF.py):
//   28-head :782-787  MaxPool2d(3,2,ceil_mode) (:353) -> AvgPool2d(7) (:262) -> squeeze -> Linear(256,101)
//   14-head :789-793  AvgPool2d(7) -> squeeze -> Linear(512,101)
//   7-head  :843-847  AvgPool2d(7) -> squeeze -> Linear(1024,101)
// (dropout is the identity in eval).  One block per pair: the pooled [C] vector is
// built in LDS with channel-coalesced reads of the channels-last map, then each wave
// takes output classes round-robin, lanes along C, and reduces with wavefront shuffles.
// K6 stands for SegmentConsensus 'avg' (basic_ops.py:19-21) as used at Flow_OFF.py:867-876.
#include "offk_common.h"
#include "offk_internal.h"

#include <mutex>
#include <set>
#include <utility>

namespace offk {

hipError_t lds_attr_once(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({kernel, dev})) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({kernel, dev});
  return e;
}

constexpr int HEAD_MAX_C = 1024;

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// One block per image.  Pooling: a thread owns a channel QUAD (16-B loads, 64 lanes = 1 KiB
// contiguous per pixel) and the 256 threads are split into 256/(C/4) pixel groups whose partial
// sums meet in LDS.  FC: one output class per wave at a time, lanes along C with 16-B loads of
// the weight row, wavefront-shuffle reduction.
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, int cs, int coff, int H, int W, int C, int maxpool,
                                                   const float* __restrict__ fw, const float* __restrict__ fb,
                                                   int ncls, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float part[4 * HEAD_MAX_C];   // [group][C] partial sums (groups <= 4 when C >= 256)
  __shared__ __attribute__((aligned(16))) float pooled[HEAD_MAX_C];
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* xi = x + (size_t)img * H * W * cs + coff;
  const int nq = C >> 2;                          // channel quads
  int ngroups = 256 / nq;
  if (ngroups < 1) ngroups = 1;
  if (ngroups > 4) ngroups = 4;
  const int Ho = maxpool ? (H - 3 + 1) / 2 + 1 : H, Wo = maxpool ? (W - 3 + 1) / 2 + 1 : W;
  const int nwin = Ho * Wo;
  for (int q = tid % (nq < 256 ? nq : 256); q < nq; q += 256) {
    const int g = nq < 256 ? tid / nq : 0;
    if (g < ngroups) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (maxpool) {
        // MaxPool2d(kernel 3, stride 2, pad 0, ceil_mode=True): windows clipped at the border
        for (int wdw = g; wdw < nwin; wdw += ngroups) {
          const int oy = wdw / Wo, ox = wdw - oy * Wo;
          float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int y = 2 * oy + dy, xx = 2 * ox + dx;
              if (y < H && xx < W) m = max4(m, *reinterpret_cast<const float4*>(xi + (size_t)(y * W + xx) * cs + 4 * q));
            }
          s = add4(s, m);
        }
      } else {
#pragma unroll 7
        for (int px = g; px < nwin; px += ngroups) s = add4(s, *reinterpret_cast<const float4*>(xi + (size_t)px * cs + 4 * q));
      }
      *reinterpret_cast<float4*>(part + g * C + 4 * q) = s;
    }
  }
  __syncthreads();
  const float inv = 1.f / (float)nwin;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int g = 0; g < ngroups; ++g) s += part[g * C + c];
    pooled[c] = s * inv;
  }
  __syncthreads();
  // FC: a wave takes four output classes at a time (their weight-row loads are all in flight
  // together), lanes along C with 16-B loads, wavefront-shuffle reduction
  const int lane = tid & 63, wave = tid >> 6;
  for (int o0 = wave * 4; o0 < ncls; o0 += 16) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = lane; q < nq; q += 64) {
      const float4 pv = *reinterpret_cast<const float4*>(pooled + 4 * q);
      float4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = min(o0 + j, ncls - 1);
        wv[j] = *reinterpret_cast<const float4*>(fw + (size_t)o * C + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = fmaf(wv[j].x, pv.x, a[j]); a[j] = fmaf(wv[j].y, pv.y, a[j]);
        a[j] = fmaf(wv[j].z, pv.z, a[j]); a[j] = fmaf(wv[j].w, pv.w, a[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a[j] += __shfl_down(a[j], off);
      if (lane == 0 && o0 + j < ncls) out[(size_t)img * ncls + o0 + j] = a[j] + fb[o0 + j];
    }
  }
}

hipError_t head_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, int maxpool, const float* fw,
                       const float* fb, int ncls, float* out, hipStream_t st, const char** why) {
  *why = nullptr;
  if (C > HEAD_MAX_C || C < 4 || (C & 3) || (x_cs & 3) || (x_coff & 3) || n_img <= 0) {
    *why = "head: need 4 <= C <= 1024, C and the channel slice 16-byte aligned";
    return hipErrorInvalidValue;
  }
  if (maxpool && (H < 3 || W < 3)) { *why = "head: maxpool needs H,W >= 3"; return hipErrorInvalidValue; }
  hipLaunchKernelGGL(head_kernel, dim3(n_img), dim3(256), 0, st, x, x_cs, x_coff, H, W, C, maxpool, fw, fb, ncls, out);
  return hipGetLastError();
}

// Pooling half of a head as its own launch: grid (image, C / 256 channel slabs); the 256 threads of a
// block are 64 channel quads x 4 pixel groups whose partial sums meet in LDS.  Writes pooled [n_img][C].
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x, int cs, int coff, int H, int W, int C, int maxpool,
                                                   float* __restrict__ pooled) {
  __shared__ __attribute__((aligned(16))) float part[4][256];
  const int img = blockIdx.x, c0 = blockIdx.y * 256, tid = threadIdx.x;
  const int q = tid & 63, g = tid >> 6;            // channel quad within the slab, pixel group
  const float* xi = x + (size_t)img * H * W * cs + coff + c0 + 4 * q;
  const bool live = c0 + 4 * q < C;
  const int Ho = maxpool ? (H - 3 + 1) / 2 + 1 : H, Wo = maxpool ? (W - 3 + 1) / 2 + 1 : W;
  const int nwin = Ho * Wo;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    if (maxpool) {
      for (int wdw = g; wdw < nwin; wdw += 4) {
        const int oy = wdw / Wo, ox = wdw - oy * Wo;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int y = 2 * oy + dy, xx = 2 * ox + dx;
            if (y < H && xx < W) m = max4(m, *reinterpret_cast<const float4*>(xi + (size_t)(y * W + xx) * cs));
          }
        s = add4(s, m);
      }
    } else {
#pragma unroll 13
      for (int px = g; px < nwin; px += 4) s = add4(s, *reinterpret_cast<const float4*>(xi + (size_t)px * cs));
    }
  }
  *reinterpret_cast<float4*>(&part[g][4 * q]) = s;
  __syncthreads();
  const int c = c0 + tid;
  if (c < C) pooled[(size_t)img * C + c] = (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) * (1.f / (float)nwin);
}
hipError_t pool_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, int maxpool, float* pooled,
                       hipStream_t st) {
  if ((C & 3) || (x_cs & 3) || (x_coff & 3) || n_img <= 0 || (maxpool && (H < 3 || W < 3))) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pool_kernel, dim3(n_img, (C + 255) / 256), dim3(256), 0, st, x, x_cs, x_coff, H, W, C, maxpool, pooled);
  return hipGetLastError();
}

// 28-head (RGB_OFF.py:782-787): 3x3 / stride 2 max pool (windows clipped at the border) and the sum of the pooled cells, as
// FOUR partial rows per image -- block (image, j) takes the pool rows [j * rpb, (j + 1) * rpb), rpb = ceil(Ho / 4) -- in the
// layout fc_pooled_kernel's tiles mode reads ([n_img * 4][C]).  256 threads = 64 channel quads x 4 window groups; a thread's
// three or four windows are 27 - 36 independent 16-byte loads (pool_kernel walks 12 windows per thread one after the other on a
// grid of n_img blocks: 42 us for 77 MB).
__global__ __launch_bounds__(256) void maxpool_rows_kernel(const float* __restrict__ x, int cs, int coff, int H, int W, int C,
                                                           float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[4][256];
  const int img = blockIdx.x >> 2, j = blockIdx.x & 3, c0 = blockIdx.y * 256, tid = threadIdx.x;
  const int q = tid & 63, g = tid >> 6;
  const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1, rpb = (Ho + 3) / 4;
  const int oy0 = j * rpb, nwin = max(0, min(rpb, Ho - oy0)) * Wo;
  const float* xi = x + (size_t)img * H * W * cs + coff + c0 + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 + 4 * q < C) {
#pragma unroll 2
    for (int wdw = g; wdw < nwin; wdw += 4) {
      const int oy = oy0 + wdw / Wo, ox = wdw % Wo;
      float4 v[9];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int y = min(2 * oy + dy, H - 1), xx = min(2 * ox + dx, W - 1);      // a clipped tap repeats its neighbour: same max
          v[3 * dy + dx] = *reinterpret_cast<const float4*>(xi + (size_t)(y * W + xx) * cs);
        }
      float4 m = v[0];
#pragma unroll
      for (int t = 1; t < 9; ++t) m = max4(m, v[t]);
      s = add4(s, m);
    }
  }
  *reinterpret_cast<float4*>(&red[g][4 * q]) = s;
  __syncthreads();
  const int c = c0 + tid;
  if (c < C) part[(size_t)blockIdx.x * C + c] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
hipError_t maxpool_rows_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, float* part, hipStream_t st) {
  if ((C & 3) || (x_cs & 3) || (x_coff & 3) || n_img <= 0 || H < 3 || W < 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(maxpool_rows_kernel, dim3(n_img * 4, (C + 255) / 256), dim3(256), 0, st, x, x_cs, x_coff, H, W, C, part);
  return hipGetLastError();
}

// FC half of a head: grid (image, ceil(classes / 8)); every thread takes one channel quad for eight
// classes, so all of a block's weight loads are in flight at once (the op is pure latency), then a
// wavefront-shuffle + LDS reduction.
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ pooled, int C, const float* __restrict__ fw,
                                                 const float* __restrict__ fb, int ncls, float* __restrict__ out) {
  __shared__ float red[4][8];
  const int img = blockIdx.x, o0 = blockIdx.y * 8, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c4 = tid * 4; c4 < C; c4 += 1024) {
    const float4 pv = *reinterpret_cast<const float4*>(pooled + (size_t)img * C + c4);
    float4 wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = *reinterpret_cast<const float4*>(fw + (size_t)min(o0 + j, ncls - 1) * C + c4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = fmaf(wv[j].x, pv.x, a[j]); a[j] = fmaf(wv[j].y, pv.y, a[j]);
      a[j] = fmaf(wv[j].z, pv.z, a[j]); a[j] = fmaf(wv[j].w, pv.w, a[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[j] += __shfl_down(a[j], off);
    if (lane == 0) red[wave][j] = a[j];
  }
  __syncthreads();
  if (tid < 8 && o0 + tid < ncls)
    out[(size_t)img * ncls + o0 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + fb[o0 + tid];
}
hipError_t fc_launch(const float* pooled, int n_img, int C, const float* fw, const float* fb, int ncls, float* out, hipStream_t st) {
  if ((C & 3) || n_img <= 0 || ncls <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fc_kernel, dim3(n_img, (ncls + 7) / 8), dim3(256), 0, st, pooled, C, fw, fb, ncls, out);
  return hipGetLastError();
}

// FC of a head whose average pool was folded into the producing conv's epilogue (conv_igemm.hip, ConvArgs.pool_part):
//   pooled[img][c] = (1 / hw) * sum over the 32-row slabs the image touches of part[slab][first or second image][c]
//   out[img][cls]  = fb[cls] + sum_c fw[cls][c] * pooled[img][c]                  (RGB_OFF.py:789-793, :843-847)
// as a small fp32-MFMA GEMM: v_mfma_f32_16x16x4_f32, A = 16 weight rows (classes), B = 16 images; a block = 16 images x 32
// classes, its four waves split K and meet in LDS (fixed order).  Replaces a pooling launch that re-read the 77 MB map
// plus a per-image VALU GEMV on a (images, 13) grid.
typedef float fcx4 __attribute__((ext_vector_type(4)));
constexpr int kFcWaves = 8;       // K is split over the block's waves: C / 8 channels each, in batches of four 16-channel steps
// tiles != 0: part = [n_img * 4][C], one row per 4x4 output tile of the Winograd path (winograd.hip): an image = 4 rows
__device__ __forceinline__ void fc_pooled_body(const float* __restrict__ part, int hw, int tiles, int n_img, int C,
                                               const float* __restrict__ fw, const float* __restrict__ fb, int ncls,
                                               float* __restrict__ out) {
  __shared__ fcx4 red[kFcWaves][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  const int img = min((int)blockIdx.x * 16 + li, n_img - 1);
  const int cls0 = blockIdx.y * 32;
  // the image's rows [img * hw, img * hw + hw) touch slabs s_lo .. s_lo + 2 at most (hw >= 32); a slab the image does not
  // reach is read with weight 0 from the image's last slab (no branch: every load of a batch goes out before the first use)
  const int r0 = img * hw, s_lo = r0 >> 5, s_hi = (r0 + hw - 1) >> 5;
  const float* ps[4];
  float pw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sl = min(s_lo + j, s_hi);
    const int slot = (sl * 32) / hw == img ? 0 : 1;          // is this image the slab's first or its second?
    ps[j] = tiles ? part + ((size_t)img * 4 + j) * C + 4 * kq : part + ((size_t)sl * 2 + slot) * C + 4 * kq;
    pw[j] = tiles || s_lo + j <= s_hi ? 1.f : 0.f;
  }
  const int per = ((C / kFcWaves + 63) / 64) * 64;       // a wave's K slice: whole batches of 64 channels (C < 512: the last waves idle)
  const int kbeg = wave * per, kend = min(C, kbeg + per);
  const float* w0 = fw + (size_t)min(cls0 + li, ncls - 1) * C + 4 * kq;
  const float* w1 = fw + (size_t)min(cls0 + 16 + li, ncls - 1) * C + 4 * kq;
  fcx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int kb = kbeg; kb < kend; kb += 64) {
    fcx4 xr[4][4], a0[4], a1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[u][j] = *reinterpret_cast<const fcx4*>(ps[j] + kb + 16 * u);
      a0[u] = *reinterpret_cast<const fcx4*>(w0 + kb + 16 * u);
      a1[u] = *reinterpret_cast<const fcx4*>(w1 + kb + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const fcx4 x = ((xr[u][0] * pw[0] + xr[u][1] * pw[1]) + xr[u][2] * pw[2]) + xr[u][3] * pw[3];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].x, x.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].x, x.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].y, x.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].y, x.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].z, x.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].z, x.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].w, x.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].w, x.w, acc1, 0, 0, 0);
    }
  }
  red[wave][0][lane] = acc0;
  red[wave][1][lane] = acc1;
  __syncthreads();
  if (wave < 2) {          // wave t finishes class tile t: lane = (image li, classes cls0 + 16 t + 4 kq .. + 3), K slices in order
    fcx4 v = red[0][wave][lane];
#pragma unroll
    for (int w = 1; w < kFcWaves; ++w) v += red[w][wave][lane];
    const int gi = blockIdx.x * 16 + li, c = cls0 + 16 * wave + 4 * kq;
    const float inv = 1.f / (float)hw;
    if (gi < n_img) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < ncls) out[(size_t)gi * ncls + c + j] = v[j] * inv + fb[c + j];
    }
  }
}
__global__ __launch_bounds__(64 * kFcWaves) void fc_pooled_kernel(const float* __restrict__ part, int hw, int tiles, int n_img, int C,
                                                                  const float* __restrict__ fw, const float* __restrict__ fb, int ncls,
                                                                  float* __restrict__ out) {
  fc_pooled_body(part, hw, tiles, n_img, C, fw, fb, ncls, out);
}
// the FCs of up to three heads in ONE launch (blockIdx.z = head): each is a few microseconds of latency-bound work on a (24, 4) grid;
// one behind the other they cost the sum, side by side the longest (the 1024-channel one)
__global__ __launch_bounds__(64 * kFcWaves) void fc_pooled_multi_kernel(FcPooledJobs j) {
  const FcPooledJob& h = j.job[blockIdx.z];
  fc_pooled_body(h.part, h.hw, h.tiles, j.n_img, h.C, h.fw, h.fb, j.ncls, h.out);
}
hipError_t fc_pooled_multi_launch(const FcPooledJobs& j, hipStream_t st) {
  if (j.njobs < 1 || j.njobs > 3 || j.n_img <= 0 || j.ncls <= 0) return hipErrorInvalidValue;
  for (int i = 0; i < j.njobs; ++i)
    if (j.job[i].C % 64 || j.job[i].hw < 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fc_pooled_multi_kernel, dim3((j.n_img + 15) / 16, (j.ncls + 31) / 32, j.njobs), dim3(64 * kFcWaves), 0, st, j);
  return hipGetLastError();
}
hipError_t fc_pooled_launch(const float* part, int hw, int tiles, int n_img, int C, const float* fw, const float* fb, int ncls, float* out,
                            hipStream_t st) {
  if (C % 64 || hw < 32 || n_img <= 0 || ncls <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fc_pooled_kernel, dim3((n_img + 15) / 16, (ncls + 31) / 32), dim3(64 * kFcWaves), 0, st, part, hw, tiles, n_img, C, fw, fb, ncls, out);
  return hipGetLastError();
}

__global__ void vec_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
hipError_t vec_add_launch(const float* a, const float* b, float* o, int n, hipStream_t st) {
  hipLaunchKernelGGL(vec_add_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, o, n);
  return hipGetLastError();
}

__global__ void consensus_kernel(const float* __restrict__ x, int T, int C, float* __restrict__ out) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[((size_t)b * T + t) * C + c];
    out[(size_t)b * C + c] = s / (float)T;
  }
}
// the SegmentConsensus of up to three heads in one launch (blockIdx.y = head)
__global__ void consensus_multi_kernel(const float* x0, const float* x1, const float* x2, int T, int C, float* o0, float* o1, float* o2) {
  const float* const x = blockIdx.y == 0 ? x0 : blockIdx.y == 1 ? x1 : x2;
  float* const out = blockIdx.y == 0 ? o0 : blockIdx.y == 1 ? o1 : o2;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[((size_t)b * T + t) * C + c];
    out[(size_t)b * C + c] = s / (float)T;
  }
}
hipError_t consensus_multi_launch(const float* const x[3], float* const out[3], int nheads, int B, int T, int C, hipStream_t st) {
  if (nheads < 1 || nheads > 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(consensus_multi_kernel, dim3(B, nheads), dim3(128), 0, st, x[0], x[1], x[2], T, C, out[0], out[1], out[2]);
  return hipGetLastError();
}
hipError_t consensus_launch(const float* x, int B, int T, int C, float* out, hipStream_t st) {
  hipLaunchKernelGGL(consensus_kernel, dim3(B), dim3(128), 0, st, x, T, C, out);
  return hipGetLastError();
}

// K7: fused[v][c] = sum_i w_i * mean_k s_i[v][k][c]; pred[v] = argmax_c (first maximum, like np.argmax)
constexpr int SF_MAX_SETS = 8;
struct ScoreFusionArgs { const float* s[SF_MAX_SETS]; float w[SF_MAX_SETS]; int n, crops, classes; float* fused; int* pred; };
__global__ __launch_bounds__(256) void score_fusion_kernel(ScoreFusionArgs a) {
  __shared__ float best_v[256];
  __shared__ int best_i[256];
  const int v = blockIdx.x, tid = threadIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < a.classes; c += 256) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < SF_MAX_SETS; ++i)
      if (i < a.n) {
        const float* p = a.s[i] + ((size_t)v * a.crops) * a.classes + c;
        float m = 0.f;
        for (int k = 0; k < a.crops; ++k) m += p[(size_t)k * a.classes];
        acc += a.w[i] * (m / (float)a.crops);
      }
    a.fused[(size_t)v * a.classes + c] = acc;
    if (acc > bv) { bv = acc; bi = c; }
  }
  if (!a.pred) return;
  best_v[tid] = bv; best_i[tid] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float ov = best_v[tid + s];
      const int oi = best_i[tid + s];
      if (ov > best_v[tid] || (ov == best_v[tid] && oi < best_i[tid])) { best_v[tid] = ov; best_i[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) a.pred[v] = best_i[0];
}
hipError_t score_fusion_launch(const float* const* scores, const float* weights, int n, int videos, int crops, int classes,
                               float* fused, int* pred, hipStream_t st) {
  if (n < 1 || n > SF_MAX_SETS) return hipErrorInvalidValue;
  ScoreFusionArgs a;
  for (int i = 0; i < SF_MAX_SETS; ++i) { a.s[i] = i < n ? scores[i] : nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
  a.n = n; a.crops = crops; a.classes = classes; a.fused = fused; a.pred = pred;
  hipLaunchKernelGGL(score_fusion_kernel, dim3(videos), dim3(256), 0, st, a);
  return hipGetLastError();
}

// [n][C][HW] -> [n*HW][C], tiled through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int HW, float* __restrict__ dst) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, q = q0 + tx;
    t[r][tx] = (c < C && q < HW) ? src[((size_t)n * C + c) * HW + q] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int q = q0 + r, c = c0 + tx;
    if (c < C && q < HW) dst[((size_t)n * HW + q) * C + c] = t[tx][r];
  }
}
hipError_t nchw_to_nhwc_launch(const float* src, int n_img, int C, int HW, float* dst, hipStream_t st) {
  dim3 grid((HW + 31) / 32, (C + 31) / 32, n_img);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, st, src, C, HW, dst);
  return hipGetLastError();
}

// channel slice [coff, coff+C) of a channels-last buffer with cs channels/pixel -> [n][C][HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int cs, int coff, int C, int HW,
                                                           float* __restrict__ dst) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int q = q0 + r, c = c0 + tx;
    t[r][tx] = (c < C && q < HW) ? src[((size_t)n * HW + q) * cs + coff + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, q = q0 + tx;
    if (c < C && q < HW) dst[((size_t)n * C + c) * HW + q] = t[tx][r];
  }
}
hipError_t nhwc_to_nchw_launch(const float* src, int cs, int coff, int n_img, int C, int HW, float* dst, hipStream_t st) {
  dim3 grid((HW + 31) / 32, (C + 31) / 32, n_img);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, st, src, cs, coff, C, HW, dst);
  return hipGetLastError();
}

}  // namespace offk
