// K1T: the OFF units' 1x1 reduce convolutions FUSED with the temporal difference (inference path).
//
// Stands for, per tap site (reference RGB_OFF.py, site 3a lines):
//   G = relu(motion_conv_gen_s(X))                         :597-598
//   T[b*(L-1)+t] = G[b*L+t+1] - G[b*L+t]                   :599-604
//   D = motion_spatial_down_s(X[:B*(L-1)])                 :609-610  (quirk Q1 via down_row)
// K1 (pw_reduce.hip) writes G to HBM and K2 (sobel_tdiff.hip) reads it back to subtract neighbouring frames:
// 2 x 128 x N x HW floats that exist only to be differenced.  Here a block owns a (clip, 32-pixel chunk) and ALL
// frames of it: seven 32-row MFMA tiles = seven frames, so the frames t and t+1 of a pixel sit in the same lane
// and register index of two accumulator tiles and T is a register subtraction in the epilogue.  G never leaves the
// chip; per clip the units then move X (31.6 MB) + T (8.1 MB) + D (2.0 MB) + S (2.0 MB) = SURVEY.md 8(d)'s
// "fully fused" 45.4 MB instead of 68.4 MB.  D still goes to HBM: the 3x3 on it needs a spatial halo, the S half
// of K2 (its S-blocks alone) runs behind this kernel.
//
// GEMM view per block: rows = (frame j, pixel) = 7 x 32, K = C, N = 160 (128 gen + 32 down).  Wave w owns gen
// columns [32w, 32w+32) for all seven frames (7 accumulator tiles) and the down columns for frames w and w+4.
// X is NCHW: per (frame, channel) the 32 pixels are one 128-byte run; 4 k-rows x 4 pixels per thread, register
// transpose into the [row][k] LDS image as in K1.  L > 7: temporal groups of 7 frames overlapping by one.
// Training keeps the unfused pair (the backward needs G for the ReLU mask).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
constexpr int PT_FT = 7;                 // frames (row tiles) per block
constexpr int PT_BM = 32 * PT_FT, PT_BN = 160;

// row of D for frame t of clip b (quirk Q1: reference_flat slices the flat frame axis)
__device__ __forceinline__ int pt_down_row(int b, int t, int L, int P, int slice_mode) {
  if (slice_mode == 0) { const int f = b * L + t; return f < P ? f : -1; }
  return t < L - 1 ? b * (L - 1) + t : -1;
}
}  // namespace

int pt_tgroups(int L) { return L <= PT_FT ? 1 : (L - 2) / (PT_FT - 1) + 1; }

typedef float f4v __attribute__((ext_vector_type(4)));
// NT bit 0: non-temporal loads of the feature map (read once); bit 1: non-temporal stores of T and D
// LEAN: buffer-descriptor addressing (every feature-map part and the weight tables < 2^31 bytes): the per-K-tile address
// arithmetic is scalar (part pick, channel offset) plus one 24-bit multiply-add per frame slot, pixel quads are read where
// they lie (a quad over the end of the tensor reads zeros, over the end of a plane the next plane's pixels -- rows that
// are never written), and the prefetch is unconditional and pinned in front of the MFMAs as in conv_igemm.hip.  The
// fp32 MFMA shares its lanes with the VALU: the ~90 64-bit address / select / shift instructions of the pointer form
// per K-tile were matrix time.
// (Rounds 1 - 4 carried a two-plane bf16x3 form, a producer / consumer form and a "B direct" form of this kernel as template branches;
//  they were retired with the bf16x3 mode in round 5.  What is left is the fp32 fallback of pw_tdiff16_kernel: weights bound in the
//  caller's tensors (offk_bind_weight) or a tensor past 2^31 bytes.)
template <int NT, int LEAN>
__global__ __launch_bounds__(256, 2) void pw_tdiff_kernel(PtParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];     // (PT_BM + PT_BN) * LDS_K * 4 bytes
  float* As = reinterpret_cast<float*>(lds);                    // fp32: [224][LDS_K] then [160][LDS_K]
  float* Bs = As + PT_BM * LDS_K;

  PtSite S;
  int nblk_site;
#define OFFK_PT_PICK(i)                                                                                \
  S.w = p.s[i].w; S.bias = p.s[i].bias; S.D = p.s[i].D; S.M = p.s[i].M; S.m_cs = p.s[i].m_cs;           \
  S.w_down = p.s[i].w_down; S.bias_down = p.s[i].bias_down;                                              \
  S.m_coff = p.s[i].m_coff; S.C = p.s[i].C; S.HW = p.s[i].HW; S.chunks = p.s[i].chunks;                \
  S.nrem = p.s[i].nrem; S.rsh = p.s[i].rsh;                                                            \
  S.blk_begin = p.s[i].blk_begin; S.nparts = p.s[i].nparts;                                           \
  nblk_site = (i + 1 < p.nsites ? p.s[i + 1].blk_begin : p.total_blocks) - p.s[i].blk_begin;          \
  S.xp[0] = p.s[i].xp[0]; S.xp[1] = p.s[i].xp[1]; S.xp[2] = p.s[i].xp[2]; S.xp[3] = p.s[i].xp[3];     \
  S.cp[0] = p.s[i].cp[0]; S.cp[1] = p.s[i].cp[1]; S.cp[2] = p.s[i].cp[2]; S.cp[3] = p.s[i].cp[3];
  OFFK_PT_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.x >= p.s[i].blk_begin) { OFFK_PT_PICK(i) }
#undef OFFK_PT_PICK
  const int C = S.C, HW = S.HW, L = p.L;
  int local = xcd_contiguous((int)blockIdx.x - S.blk_begin, nblk_site);   // neighbouring pixel chunks of a clip on one XCD
  const int tg = local % p.tgroups; local /= p.tgroups;
  // (clip, chunk) blocks first, then the leftover blocks (PtSite): a leftover block covers clips b .. b + (32 >> rsh) - 1
  const int nfull = p.B * S.chunks;
  const bool leftover = local >= nfull;
  const int rsh = leftover ? S.rsh : 5, rmask = (1 << rsh) - 1;
  const int b = leftover ? (local - nfull) << (5 - rsh) : local / S.chunks;
  const int q0 = (leftover ? S.chunks : local - b * S.chunks) * 32;
  const int t0 = tg * (PT_FT - 1);
  const int nf = min(PT_FT, L - t0);          // frames of this block (>= 2)
  const bool last_group = tg == p.tgroups - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = threadIdx.x & 255;          // index within the four loading waves

  // ---- loader: thread = (k quad kq, pixel quad pq, frame select fs): frames fs and fs + 4 --------------------
  const int kq = tid & 7, pq = (tid >> 3) & 7, fs = tid >> 6;
  const int cq = (4 * pq) >> rsh;                                // clip of this thread's pixel quad within the block (0 unless packed)
  const int k0px = q0 + ((4 * pq) & rmask), kkpx = min(k0px, HW - 4);   // a quad that straddles the plane end is read from HW-4
  const int sh = !LEAN && k0px < HW ? k0px - kkpx : 0;           // and shifted into place (HW % 4 != 0 only)
  const bool px_ok = k0px < HW && b + cq < p.B;
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  auto locate = [&](int k0, const float*& xb, int& cpart, int& kl) {
    xb = S.xp[0]; cpart = S.cp[0]; kl = k0;
    if (S.nparts > 1 && kl >= S.cp[0]) {
      kl -= S.cp[0]; xb = S.xp[1]; cpart = S.cp[1];
      if (S.nparts > 2 && kl >= S.cp[1]) {
        kl -= S.cp[1]; xb = S.xp[2]; cpart = S.cp[2];
        if (S.nparts > 3 && kl >= S.cp[2]) { kl -= S.cp[2]; xb = S.xp[3]; cpart = S.cp[3]; }
      }
    }
  };
  constexpr int NRG = 8 + 5;          // A: frame fs (4 k-rows), frame fs+4 (4 k-rows); B: 5 weight rows
  float4 rg[NRG];
  const float* wbase = S.w + (size_t)(tid >> 3) * C + 4 * (tid & 7);
  const float* wdbase = S.w_down + (size_t)(tid >> 3) * C + 4 * (tid & 7);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t wrs, wdrs;
  int jA[2], vA[2], woff = 0;
  if constexpr (LEAN) {
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.w), 0, kGenCh * C * 4, 0x00020000);
    wdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.w_down), 0, kDownCh * C * 4, 0x00020000);
    woff = ((tid >> 3) * C + 4 * (tid & 7)) * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = fs + 4 * half;
      const bool ok = px_ok && j < nf;
      jA[half] = ok ? j + cq * L : 0;                                 // frame slot past the group / pixel quad past the
      vA[half] = ok ? (4 * kq * HW + k0px) * 4 : (int)0x80000000;     // plane: an offset past every descriptor -> zeros
    }
  }
  auto load_tile = [&](const int set, int k0) {
    const float* xb; int cpart, kl;
    locate(k0, xb, cpart, kl);
    if constexpr (LEAN) {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.B * L * cpart * HW * 4, 0x00020000);
      const int fstride = cpart * HW * 4;                                   // bytes per frame of this part (scalar)
      const int s0 = (((b * L + t0) * cpart + kl) * HW) * 4;                // scalar: frame t0, channel kl
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int voff = jA[half] * fstride + vA[half];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, s0 + i * HW * 4, 0);
          rg[set + 4 * half + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, (32 * r * C + k0) * 4, 0);
        rg[set + 8 + r] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wdrs, woff, k0 * 4, 0);
      rg[set + 12] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = fs + 4 * half;
      const bool ok = px_ok && j < nf;
      const float* base = xb + ((size_t)(b * L + t0 + (ok ? j : 0)) * cpart + kl + 4 * kq) * HW + kkpx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f4u* src = reinterpret_cast<const f4u*>(ok ? base + (size_t)i * HW : p.zeros);
        const f4u v = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
        rg[set + 4 * half + i] = make_float4(v.x, v.y, v.z, v.w);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rg[set + 8 + r] = *reinterpret_cast<const float4*>(wbase + (size_t)32 * r * C + k0);
    rg[set + 12] = *reinterpret_cast<const float4*>(wdbase + k0);
  };
  auto shifted = [&](float4 v) {
    if (LEAN || sh == 0) return v;
    return sh == 1 ? make_float4(v.y, v.z, v.w, 0.f) : (sh == 2 ? make_float4(v.z, v.w, 0.f, 0.f) : make_float4(v.w, 0.f, 0.f, 0.f));
  };
  auto store_tile = [&](const int set) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = fs + 4 * half;
      if (j < PT_FT) {
        const float4 a0 = shifted(rg[set + 4 * half]), a1 = shifted(rg[set + 4 * half + 1]), a2 = shifted(rg[set + 4 * half + 2]), a3 = shifted(rg[set + 4 * half + 3]);
        const int row = j * 32 + 4 * pq;
        float* dst = As + row * LDS_K + 4 * kq;
        *reinterpret_cast<float4*>(dst) = make_float4(a0.x, a1.x, a2.x, a3.x);
        *reinterpret_cast<float4*>(dst + LDS_K) = make_float4(a0.y, a1.y, a2.y, a3.y);
        *reinterpret_cast<float4*>(dst + 2 * LDS_K) = make_float4(a0.z, a1.z, a2.z, a3.z);
        *reinterpret_cast<float4*>(dst + 3 * LDS_K) = make_float4(a0.w, a1.w, a2.w, a3.w);
      }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
      *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[set + 8 + r];
  };

  const int nkt = C / BK;
  f32x16 acc[PT_FT + 2];   // 0..6: gen tile of frame j; 7, 8: down tiles of frames wave and wave + 4
#pragma unroll
  for (int t = 0; t < PT_FT + 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int r32 = lane & 31, h = lane >> 5;
  const bool d1 = wave + 4 < PT_FT;          // waves 0-2 own a second down tile
  load_tile(0, 0);
#ifdef OFFK_PT_TIMING
  unsigned long long t_st = 0, t_s1 = 0, t_ld = 0, t_mm = 0, t_s2 = 0, c0 = __builtin_readcyclecounter(), c1;
#define OFFK_TICK(var) c1 = __builtin_readcyclecounter(); var += c1 - c0; c0 = c1;
#else
#define OFFK_TICK(var)
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    store_tile(0);
    OFFK_TICK(t_st)
    __syncthreads();
    OFFK_TICK(t_s1)
    if constexpr (LEAN) {
      load_tile(0, min(kt + 1, nkt - 1) * BK);      // unconditional (no phi copies of in-flight loads), and kept in
      __builtin_amdgcn_sched_barrier(0);            // front of the MFMAs
    } else {
      if (kt + 1 < nkt) load_tile(0, (kt + 1) * BK);
    }
    OFFK_TICK(t_ld)
    {
#pragma unroll 1
      for (int g = 0; g < BK / 8; ++g) {
        const float4 bw = *reinterpret_cast<const float4*>(Bs + (wave * 32 + r32) * LDS_K + 8 * g + 4 * h);
        const float4 dw = *reinterpret_cast<const float4*>(Bs + (kGenCh + r32) * LDS_K + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < PT_FT; ++j) {
          const float4 a = *reinterpret_cast<const float4*>(As + (j * 32 + r32) * LDS_K + 8 * g + 4 * h);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.x, a.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.y, a.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.z, a.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.w, a.w, acc[j], 0, 0, 0);
        }
        {
          const float4 a = *reinterpret_cast<const float4*>(As + (wave * 32 + r32) * LDS_K + 8 * g + 4 * h);
          acc[PT_FT] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.x, a.x, acc[PT_FT], 0, 0, 0);
          acc[PT_FT] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.y, a.y, acc[PT_FT], 0, 0, 0);
          acc[PT_FT] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.z, a.z, acc[PT_FT], 0, 0, 0);
          acc[PT_FT] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.w, a.w, acc[PT_FT], 0, 0, 0);
        }
        if (d1) {
          const float4 a = *reinterpret_cast<const float4*>(As + ((wave + 4) * 32 + r32) * LDS_K + 8 * g + 4 * h);
          acc[PT_FT + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.x, a.x, acc[PT_FT + 1], 0, 0, 0);
          acc[PT_FT + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.y, a.y, acc[PT_FT + 1], 0, 0, 0);
          acc[PT_FT + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.z, a.z, acc[PT_FT + 1], 0, 0, 0);
          acc[PT_FT + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dw.w, a.w, acc[PT_FT + 1], 0, 0, 0);
        }
      }
    }
    OFFK_TICK(t_mm)
    __syncthreads();
    OFFK_TICK(t_s2)
  }
#ifdef OFFK_PT_TIMING
  if (p.dbg && threadIdx.x == 0) {
    atomicAdd(p.dbg + 0, t_st); atomicAdd(p.dbg + 1, t_s1); atomicAdd(p.dbg + 2, t_ld); atomicAdd(p.dbg + 3, t_mm);
    atomicAdd(p.dbg + 4, t_s2); atomicAdd(p.dbg + 5, (unsigned long long)nkt); atomicAdd(p.dbg + 6, 1ull);
  }
  const unsigned long long e0 = __builtin_readcyclecounter();
#endif
#undef OFFK_TICK

  // ---- epilogue: G = relu(acc + bias) in registers, T = G[j+1] - G[j] -> M; D -> HBM -------------------------
  // accumulator tile: register (reg) <-> channel acc_row(reg, h) of the wave's 32-channel slab, lane r32 <-> pixel:
  // registers 4g .. 4g+3 of a lane are four consecutive channels, one 16-byte store
  float4 bg4[4], bd4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bg4[g] = *reinterpret_cast<const float4*>(S.bias + wave * 32 + 8 * g + 4 * h);
    bd4[g] = *reinterpret_cast<const float4*>(S.bias_down + 8 * g + 4 * h);
  }
#pragma unroll
  for (int j = 0; j < PT_FT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(fmaxf(acc[j][4 * g] + bg4[g].x, 0.f), fmaxf(acc[j][4 * g + 1] + bg4[g].y, 0.f),
                             fmaxf(acc[j][4 * g + 2] + bg4[g].z, 0.f), fmaxf(acc[j][4 * g + 3] + bg4[g].w, 0.f));
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
      acc[j][4 * g] = v.x; acc[j][4 * g + 1] = v.y; acc[j][4 * g + 2] = v.z; acc[j][4 * g + 3] = v.w;
    }
#pragma unroll
  for (int t = PT_FT; t < PT_FT + 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(acc[t][4 * g] + bd4[g].x, acc[t][4 * g + 1] + bd4[g].y, acc[t][4 * g + 2] + bd4[g].z, acc[t][4 * g + 3] + bd4[g].w);
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
      acc[t][4 * g] = v.x; acc[t][4 * g + 1] = v.y; acc[t][4 * g + 2] = v.z; acc[t][4 * g + 3] = v.w;
    }
  const int bl = b + (r32 >> rsh), pixl = q0 + (r32 & rmask);      // this lane's clip and pixel
  const size_t pair0 = (size_t)bl * (L - 1) + t0;
  const bool pix_ok = pixl < HW && bl < p.B;
#pragma unroll
  for (int j = 0; j + 1 < PT_FT; ++j) {
    if (j + 1 < nf && pix_ok) {
      float* mrow = S.M + ((pair0 + j) * HW + pixl) * S.m_cs + S.m_coff + kDownCh + wave * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4v tv = {acc[j + 1][4 * g] - acc[j][4 * g], acc[j + 1][4 * g + 1] - acc[j][4 * g + 1],
                        acc[j + 1][4 * g + 2] - acc[j][4 * g + 2], acc[j + 1][4 * g + 3] - acc[j][4 * g + 3]};
        if (NT & 2) __builtin_nontemporal_store(tv, reinterpret_cast<f4v*>(mrow + 8 * g));
        else *reinterpret_cast<f4v*>(mrow + 8 * g) = tv;
      }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int j = wave + 4 * half;
    // the frame shared with the next temporal group belongs to that group
    if (j < nf && (last_group || j < PT_FT - 1) && pix_ok) {
      const int dr = pt_down_row(bl, t0 + j, L, p.P, p.slice_mode);
      if (dr >= 0) {
        float* drow = S.D + ((size_t)dr * HW + pixl) * kDownCh + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f4v dv = {acc[PT_FT + half][4 * g], acc[PT_FT + half][4 * g + 1], acc[PT_FT + half][4 * g + 2], acc[PT_FT + half][4 * g + 3]};
          if (NT & 2) __builtin_nontemporal_store(dv, reinterpret_cast<f4v*>(drow + 8 * g));
          else *reinterpret_cast<f4v*>(drow + 8 * g) = dv;
        }
      }
    }
  }
#ifdef OFFK_PT_TIMING
  if (p.dbg && threadIdx.x == 0) atomicAdd(p.dbg + 7, __builtin_readcyclecounter() - e0);
#endif
}

// (Round 3 also built a 32-pixel LDS-DMA form of this kernel -- pw_tdiff_dma_kernel, 32x32x2 tiles, two blocks per CU: 1.318 ms at
//  B = 64 against 1.246 ms for the 16-pixel form below.  It was retired from the product build in round 4; DESIGN.md section 16.1
//  keeps its measurements, the git history its text.)
// ---------------------------------------------------------------------------------------------------------------
// 16-pixel form of the LDS-DMA kernel (round 3): v_mfma_f32_16x16x4_f32 tiles, a block owns (clip, 16-pixel chunk) x all
// frames.  Why: pw_tdiff_dma_kernel keeps 9 accumulator tiles of 32x32 per wave (144 registers, 202 in all), so two blocks per
// CU is all that fits, and a block alone on its CU -- its partner in a prologue, an epilogue or at a barrier -- runs at 83 % of
// the matrix pipe (profiles/r03/k1t_experiments.txt).  With 16x16 tiles a wave holds 14 gen tiles + up to 4 down tiles of FOUR
// registers each (72), the kernel fits three waves per SIMD and 29 KB of LDS per block: three blocks per CU cover each other.
// Same MFMA cycles per pixel (2048 FLOP per 32 cycles), twice the weight bytes per pixel from L2 (a block streams the whole
// [160][32] weight tile per K-tile for 16 pixels instead of 32) and 64-byte feature-map runs instead of 128.
//   wave w: gen channels [32w, 32w + 32) = channel tiles 2w, 2w + 1, all seven frames; down tiles (both channel tiles) of
//   frames w and w + 4.  MFMA (weights = A, pixels = B): D rows = channels, lanes = pixels -> a lane holds four consecutive
//   channels of one pixel (16-byte stores).  x operand: lane (pixel li, kq) of step s reads X[k = 4s + kq][pixel] -- one dword,
//   conflict-free -- two units ahead; unit = (k half, frame): 8 MFMAs (16 with the frame's down tiles), alternating between
//   the two channel tiles so that consecutive MFMAs never share an accumulator.
//   queue (per wave, as pw_tdiff_dma_kernel): 4 DMA instructions per step -- the 14 of a tile (7 frames x two 16-row halves)
//   over four waves, the two missing ones land in a dummy KB -- behind the first four units; the 4 weight loads of a k half
//   behind that half's last unit.  Waits: half 0: all but the 4 newest; half 1: all but 8; end of step (DMAs): all but 8.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void pw_tdiff16_kernel(PtParams p) {
  constexpr int FRAME_B = 32 * 64;               // one frame's K-tile image [32 k][16 pixels] fp32
  constexpr int STAGE_B = PT_FT * FRAME_B;       // 14 KB; two stages, then 1 KB that the surplus DMA slots write
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));

  // the block's site: nine compares on blk_begin, then ONE site's fields (the unrolled pick-by-copy of the other kernels loads
  // every field of all nine sites first: ~7 k of this kernel's 18 k prologue cycles, and a block is only ~200 k cycles long)
  int si = 0;
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.x >= p.s[i].blk_begin) si = i;
  si = __builtin_amdgcn_readfirstlane(si);
  const PtSite& S0 = p.s[si];
  PtSite S;
  S.bias = S0.bias; S.D = S0.D; S.M = S0.M; S.m_cs = S0.m_cs; S.bias_down = S0.bias_down; S.wt = S0.wt16;
  S.m_coff = S0.m_coff; S.C = S0.C; S.HW = S0.HW; S.chunks = S0.chunks; S.nrem = S0.nrem; S.rsh = S0.rsh;
  S.blk_begin = S0.blk_begin; S.nparts = S0.nparts; S.qpc = S0.qpc;
#pragma unroll
  for (int q = 0; q < 4; ++q) { S.xp[q] = S0.xp[q]; S.cp[q] = S0.cp[q]; }
  const int nblk_site = (si + 1 < p.nsites ? p.s[si + 1].blk_begin : p.total_blocks) - S.blk_begin;
  const int C = S.C, HW = S.HW, L = p.L;
  int local = xcd_contiguous((int)blockIdx.x - S.blk_begin, nblk_site);
  const int tg = local % p.tgroups; local /= p.tgroups;
  // (clip, 16-pixel chunk) blocks first, then the leftover blocks: HW % 16 pixels of 16 >> rsh clips each (rsh = 4: one clip)
  const int nfull = p.B * S.chunks;
  const bool leftover = local >= nfull;
  const int rsh = leftover ? S.rsh : 4, rmask = (1 << rsh) - 1;
  // quad stream (S.qpc > 0, every block is a "leftover" block): quads 4 local .. + 3 of the sequence (clip, quad in clip); four
  // consecutive quads cross at most one clip boundary (qpc >= 4): quad pq is quad qr0 + pq of clip b, or that minus qpc of clip b + 1
  const int qpc = S.qpc;
  const int b = qpc ? (4 * local) / qpc : leftover ? (local - nfull) << (4 - rsh) : local / S.chunks;
  const int qr0 = qpc ? 4 * local - b * qpc : 0;
  const int q0 = (leftover ? S.chunks : local - b * S.chunks) * 16;
  const int t0 = tg * (PT_FT - 1);
  const int nf = min(PT_FT, L - t0);
  const bool last_group = tg == p.tgroups - 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, kq = lane >> 4;

  // ---- DMA: instruction slot q = wave + 4 i (i = 0..3) lands frame q >> 1, k rows 16 (q & 1) + (lane >> 2), pixel quad lane & 3;
  //      slots 14, 15 (waves 2, 3, i = 3) go to the dummy KB.  q & 1 == wave & 1 for every i: ONE per-lane offset register serves
  //      the four slots, the frame rides on the scalar offset (round 3 kept four 64-bit offset pairs: eight registers) ----
  const int pq = lane & 3;
  const bool qnext = qpc && qr0 + pq >= qpc;
  const int cq = qpc ? (int)qnext : (4 * pq) >> rsh;               // clip of the quad within the block (0 unless packed)
  const int k0px = qpc ? 4 * (qr0 + pq - (qnext ? qpc : 0)) : q0 + ((4 * pq) & rmask);
  const bool px_ok = k0px < HW && b + cq < p.B;
  // byte offset of (k row, pixel quad) inside a frame's K-tile; bit 31 (= past every descriptor) where the quad lies outside
  const int vrow0 = px_ok ? ((16 * (wave & 1) + (lane >> 2)) * HW + k0px) * 4 : (int)0x80000000;
  const int vclip = cq * L;                                        // frames between the block's first clip and the lane's
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
  auto dma16 = [&](const i32x4& desc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory", "m0");
  };
  i32x4 dm_desc = {0, 0, 0, 0};
  int dm_fstride = 0, dm_s0 = 0, dm_voff = 0;
  auto dma_prep = [&](int kt) {
    const float* xb = S.xp[0]; int cpart = S.cp[0], kl = kt * BK;
    if (S.nparts > 1 && kl >= S.cp[0]) {
      kl -= S.cp[0]; xb = S.xp[1]; cpart = S.cp[1];
      if (S.nparts > 2 && kl >= S.cp[1]) {
        kl -= S.cp[1]; xb = S.xp[2]; cpart = S.cp[2];
        if (S.nparts > 3 && kl >= S.cp[2]) { kl -= S.cp[2]; xb = S.xp[3]; cpart = S.cp[3]; }
      }
    }
    const unsigned long long xa = reinterpret_cast<unsigned long long>(xb);
    dm_desc = i32x4{(int)(unsigned)xa, (int)(unsigned)(xa >> 32) & 0xffff, p.B * L * cpart * HW * 4, 0x00020000};
    dm_fstride = cpart * HW * 4;
    dm_s0 = (((b * L + t0) * cpart + kl) * HW) * 4;
    dm_voff = vrow0 + (int)__umul24((unsigned)vclip, (unsigned)dm_fstride);   // (< 2^24 each; one v_mad_u32_u24 -- a full 32-bit multiply
                                                                               //  became a 64-bit mad with a spilled addend pair; an invalid vrow0 stays >= 2^31)
  };
  auto dma_issue = [&](const int i, const int stage) {
    const int q = wave + 4 * i, fr = q >> 1;                        // scalar
    const int voff = fr < nf ? dm_voff : (int)0x80000000;
    const unsigned dst = q < 2 * PT_FT ? lds_base + stage * STAGE_B + fr * FRAME_B + (q & 1) * 1024 : lds_base + 2 * STAGE_B;
    dma16(dm_desc, dst, voff, dm_s0 + fr * dm_fstride);
  };

  // ---- weight operand: asm loads (hand-counted waits), one k QUARTER (8 k) at a time.  wg[set][ct]: gen channel tile ct of this
  //      wave, the lane's W[ch][kt * 32 + 8 qd + 4 e + kq], e = 0, 1; wd likewise for the two down channel tiles.  Two register
  //      sets: quarter qd multiplies out of set qd & 1 while the loads of quarter qd + 1 are in flight in the other.  (Round 3 kept
  //      both k HALVES of a tile resident -- 32 registers for weights and 16 for x, 152 in all: three blocks per CU.  Quarters make it
  //      16 + 8 and the kernel fits 128 registers: FOUR blocks per CU, 4 x 29 KB of LDS.) ----
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 wg[2][2], wd[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { wg[i][0] = f32x2{0.f, 0.f}; wg[i][1] = f32x2{0.f, 0.f}; wd[i][0] = f32x2{0.f, 0.f}; wd[i][1] = f32x2{0.f, 0.f}; }
  i32x4 wdesc;
  {
    const unsigned long long wa = reinterpret_cast<unsigned long long>(S.wt);
    wdesc = i32x4{(int)(unsigned)wa, (int)(unsigned)(wa >> 32) & 0xffff, kUnitCh * C * 4, 0x00020000};
  }
  const int wlane = lane * 8;
  // image: [kt][slab (4 gen + 1 down)][ct][quarter][lane] float2 (pw_pack_direct16_kernel)
  auto load_w = [&](const int set, const int qd, int kt) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int so_g = (((kt * 5 + wave) * 2 + ct) * 4 + qd) * 512, so_d = (((kt * 5 + 4) * 2 + ct) * 4 + qd) * 512;
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(wg[set][ct]) : "v"(wlane), "s"(wdesc), "s"(so_g));
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(wd[set][ct]) : "v"(wlane), "s"(wdesc), "s"(so_d));
    }
  };
#define OFFK_WAIT16(N, set) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wg[set][0]), "+v"(wg[set][1]), "+v"(wd[set][0]), "+v"(wd[set][1]))

  f32x4 ag[PT_FT][2], ad[2][2];          // gen tiles [frame][ct]; down tiles [frame wave / wave + 4][ct]
#pragma unroll
  for (int j = 0; j < PT_FT; ++j) { ag[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 2; ++i) { ad[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ad[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nkt = C / BK;
  // nine frame slots per k quarter: slots 0..6 = the gen tiles of frame s, slot 7 / 8 = the down tiles of frames `wave` / wave + 4
  // (wave 3 has no second down frame: it multiplies frame 6 again and never stores the result -- no branch in the MFMA stream:
  // per-unit tests of a runtime wave index made hipcc copy accumulators at every merge).  Units of two slots x two channel tiles =
  // four accumulator tiles in rotation, two k steps each; x reads one unit ahead.
  // Queue per wave and quarter, in issue order: ONE DMA of the next tile behind the quarter's first unit, the 4 weight loads of
  // quarter qd + 2 (same set) behind its last unit.  In front of quarter qd its weights must be there = all but the 5 newest
  // operations (the DMA and the 4 loads issued during quarter qd - 1); at the end of the step the tile's DMAs = all but the 4 newest.
  const char* const xl = lds + kq * 64 + li * 4;         // + frame * 2048 + quarter * 512 + e * 256
  const int xoffA = wave * FRAME_B, xoffB = min(wave + 4, PT_FT - 1) * FRAME_B;
  auto rdx = [&](float (&x)[2], const int st, const int qd, const int slot) {
    const char* q = xl + st * STAGE_B + qd * 512 + (slot < PT_FT ? slot * FRAME_B : (slot == PT_FT ? xoffA : xoffB));
#pragma unroll
    for (int e = 0; e < 2; ++e) x[e] = *reinterpret_cast<const float*>(q + e * 256);
  };
  auto mf = [&](f32x4& c, float a, float b) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); };
  auto mma = [&](const int st, int kt, int ktn) {
    constexpr int NS = PT_FT + 2, NU = (NS + 1) / 2;          // slots, units per k quarter
    float x[2][2][2];
    rdx(x[0][0], st, 0, 0);
    rdx(x[0][1], st, 0, 1);
#pragma unroll
    for (int u = 0; u < 4 * NU; ++u) {
      const int qd = u / NU, uu = u % NU, s0 = 2 * uu, s1 = s0 + 1, set = qd & 1;
      if (u + 1 < 4 * NU) {
        const int qn = (u + 1) / NU, un = (u + 1) % NU;
        rdx(x[(u + 1) & 1][0], st, qn, 2 * un);
        if (2 * un + 1 < NS) rdx(x[(u + 1) & 1][1], st, qn, 2 * un + 1);
      }
      if (uu == 0) { if (set == 0) OFFK_WAIT16(5, 0); else OFFK_WAIT16(5, 1); }
      __builtin_amdgcn_sched_barrier(0);
      const float (&xa)[2] = x[u & 1][0];
      const float (&xb)[2] = x[u & 1][1];
      // accumulators and weights of the two slots (literals: s0 / s1 are compile-time)
      f32x4 &a00 = s0 < PT_FT ? ag[s0][0] : ad[s0 - PT_FT][0], &a01 = s0 < PT_FT ? ag[s0][1] : ad[s0 - PT_FT][1];
      const f32x2 &w00 = s0 < PT_FT ? wg[set][0] : wd[set][0], &w01 = s0 < PT_FT ? wg[set][1] : wd[set][1];
      if (s1 < NS) {
        f32x4 &a10 = s1 < PT_FT ? ag[s1][0] : ad[s1 - PT_FT][0], &a11 = s1 < PT_FT ? ag[s1][1] : ad[s1 - PT_FT][1];
        const f32x2 &w10 = s1 < PT_FT ? wg[set][0] : wd[set][0], &w11 = s1 < PT_FT ? wg[set][1] : wd[set][1];
        mf(a00, w00.x, xa[0]); mf(a01, w01.x, xa[0]); mf(a10, w10.x, xb[0]); mf(a11, w11.x, xb[0]);
        mf(a00, w00.y, xa[1]); mf(a01, w01.y, xa[1]); mf(a10, w10.y, xb[1]); mf(a11, w11.y, xb[1]);
      } else {
        mf(a00, w00.x, xa[0]); mf(a01, w01.x, xa[0]);
        mf(a00, w00.y, xa[1]); mf(a01, w01.y, xa[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef OFFK_K16_NO_DMA
      if (uu == 0) dma_issue(qd, st ^ 1);            // every wave has left stage st ^ 1 at the last barrier
#else
      if (uu == 0) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "+v"(wg[set][0]) : "v"(wlane), "s"(wdesc));   // timing only: keeps the counts
#endif
      // the set is free: quarter qd + 2 of this tile, or quarter qd - 2 of the next one
#ifndef OFFK_K16_NO_W
      if (uu == NU - 1) { if (qd < 2) load_w(set, qd + 2, kt); else load_w(set, qd - 2, ktn); }
#else
      if (uu == NU - 1) load_w(set, qd & 1, 0);      // timing only: the same (cached) tile every time
#endif
    }
  };
#ifdef OFFK_PT_TIMING
  unsigned long long tm_mma = 0, tm_wait = 0, tm_bar = 0;
  const unsigned long long tm_begin = __builtin_readcyclecounter();
#define OFFK_STAMP(var, stmt) { const unsigned long long q0_ = __builtin_readcyclecounter(); stmt; var += __builtin_readcyclecounter() - q0_; }
#else
#define OFFK_STAMP(var, stmt) stmt;
#endif
  auto step = [&](int kt, const int st) {
    OFFK_STAMP(tm_mma, dma_prep(min(kt + 1, nkt - 1));
    __builtin_amdgcn_sched_barrier(0);
    mma(st, kt, min(kt + 1, nkt - 1)))
    OFFK_STAMP(tm_wait, asm volatile("s_waitcnt vmcnt(4)" ::: "memory"))   // this wave's DMAs of the step have landed
    OFFK_STAMP(tm_bar, __syncthreads())
  };
  dma_prep(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_issue(i, 0);
  load_w(0, 0, 0);
  load_w(1, 1, 0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // the tile and the first quarter's weights (the loop's first wait assumes it)
  __syncthreads();
#ifdef OFFK_PT_TIMING
  const unsigned long long tm_loop = __builtin_readcyclecounter();
#endif
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    step(kt, 0);
    step(kt + 1, 1);
  }
  if (kt < nkt) step(kt, 0);
#ifdef OFFK_PT_TIMING
  const unsigned long long tm_epi = __builtin_readcyclecounter();
#endif
#undef OFFK_STAMP
  // nothing may still be landing when the LDS is handed on; the weight registers stay allocated until their last load returned
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[0][0]), "+v"(wg[0][1]), "+v"(wg[1][0]), "+v"(wg[1][1]), "+v"(wd[0][0]), "+v"(wd[0][1]), "+v"(wd[1][0]), "+v"(wd[1][1]) :: "memory");
#undef OFFK_WAIT16

  // ---- epilogue: lane = (pixel li, channels 4 kq .. + 3 of a channel tile): G = relu(acc + bias), T = G[j + 1] - G[j]; D ----
  // (li / kq are re-derived from the execution mask here: kept from the prologue they are two more registers live across the K loop)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int li_e = lane_e & 15, kq_e = lane_e >> 4;
  const int qe = qr0 + (li_e >> 2);                                 // quad stream: the lane's quad, counted from clip b's first
  const bool qn_e = qpc && qe >= qpc;
  const int bl = qpc ? b + (int)qn_e : b + (li_e >> rsh), pixl = qpc ? 4 * (qe - (qn_e ? qpc : 0)) + (li_e & 3) : q0 + (li_e & rmask);
  const size_t pair0 = (size_t)bl * (L - 1) + t0;
  const bool pix_ok = pixl < HW && bl < p.B;
  // G in place of the accumulators (both channel tiles), then T pair by pair with the two channel tiles' stores back to back: a pixel's
  // 32 channels of this wave are one 128-byte line, written as two 64-byte halves -- issued next to each other they merge in L2
  // (channel tile by channel tile the second half came six stores later)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x4 bg = *reinterpret_cast<const f32x4*>(S.bias + wave * 32 + 16 * ct + 4 * kq_e);
#pragma unroll
    for (int j = 0; j < PT_FT; ++j) {
      const f32x4 v = ag[j][ct] + bg;
      ag[j][ct] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    }
  }
#pragma unroll
  for (int j = 0; j + 1 < PT_FT; ++j)
    if (j + 1 < nf && pix_ok) {
      float* const trow = S.M + ((pair0 + j) * HW + pixl) * S.m_cs + S.m_coff + kDownCh + wave * 32 + 4 * kq_e;
      *reinterpret_cast<f32x4*>(trow) = ag[j + 1][0] - ag[j][0];
      *reinterpret_cast<f32x4*>(trow + 16) = ag[j + 1][1] - ag[j][1];
    }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int j = wave + 4 * half;                 // (wave 3, half 1: j = 7 >= nf -- its duplicate tile is dropped here)
    if (j < nf && (last_group || j < PT_FT - 1) && pix_ok) {
      const int dr = pt_down_row(bl, t0 + j, L, p.P, p.slice_mode);
      if (dr >= 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 bd = *reinterpret_cast<const f32x4*>(S.bias_down + 16 * ct + 4 * kq_e);
          *reinterpret_cast<f32x4*>(S.D + ((size_t)dr * HW + pixl) * kDownCh + 16 * ct + 4 * kq_e) = ad[half][ct] + bd;
        }
      }
    }
  }
#ifdef OFFK_PT_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.dbg && threadIdx.x == 0) {
    const unsigned long long tm_end = __builtin_readcyclecounter();
    atomicAdd(p.dbg + 0, tm_loop - tm_begin); atomicAdd(p.dbg + 1, tm_epi - tm_loop); atomicAdd(p.dbg + 2, tm_end - tm_epi);
    atomicAdd(p.dbg + 3, tm_mma); atomicAdd(p.dbg + 4, tm_wait); atomicAdd(p.dbg + 5, tm_bar);
    atomicAdd(p.dbg + 6, (unsigned long long)nkt); atomicAdd(p.dbg + 7, 1ull);
  }
#endif
}

// Operand-order image for pw_tdiff16_kernel: one 8-byte item per (K-tile, slab of 32 rows, channel tile ct, k quarter qd, lane):
//   W[slab * 32 + 16 ct + li][kt * 32 + 8 qd + 4 e + kq], e = 0, 1   (li = lane & 15, kq = lane >> 4)
__global__ void pw_pack_direct16_kernel(const float* __restrict__ w, int C, float2* __restrict__ out) {
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int nitems = (C / BK) * 5 * 2 * 4 * 64;
  if (item >= nitems) return;
  const int lane = item & 63, qd = (item >> 6) & 3, ct = (item >> 8) & 1, slab = (item >> 9) % 5, kt = (item >> 9) / 5;
  const int li = lane & 15, kq = lane >> 4;
  const float* row = w + (size_t)(slab * 32 + 16 * ct + li) * C + kt * BK + 8 * qd + kq;
  out[item] = make_float2(row[0], row[4]);
}
hipError_t pw_pack_direct16_launch(const float* w160, int C, float* out, hipStream_t st) {
  const int nitems = (C / BK) * 5 * 2 * 4 * 64;
  hipLaunchKernelGGL(pw_pack_direct16_kernel, dim3((nitems + 255) / 256), dim3(256), 0, st, w160, C, reinterpret_cast<float2*>(out));
  return hipGetLastError();
}

hipError_t pw_tdiff_launch(const PtParams& p_in, hipStream_t st) {
  PtParams p = p_in;
  if (p.nsites <= 0 || p.B <= 0) return hipSuccess;
  // (Sites in descending order of C -- longest blocks first, so that the launch does not drain on the 32-tile blocks of the two 7x7
  //  sites -- measured no different from the network order: 1.270 / 1.270 ms, profiles/r04/ab_k1t_quad_stream.txt.)
  bool qstream = true;
#ifdef OFFK_TUNING_KNOBS
  { const char* se = getenv("OFFK_PW_STREAM"); if (se && *se == '0') qstream = false; }
#endif
#ifdef OFFK_PT_TIMING
  {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc(reinterpret_cast<void**>(&dbg), 256); (void)hipMemset(dbg, 0, 256); }
    p.dbg = dbg;
    if (getenv("OFFK_PT_TIMING_DUMP")) {
      unsigned long long hb[32];
      (void)hipMemcpy(hb, dbg, 256, hipMemcpyDeviceToHost);
      fprintf(stderr, "[pt timing] store %llu sync1 %llu load-issue %llu mma %llu sync2 %llu | ktiles %llu blocks %llu epilogue %llu\n",
              hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7]);
      fprintf(stderr, "[pt timing, dma form, wave 0 cycles] prologue %llu loop %llu epilogue+drain %llu | in loop: dma-issue+mma %llu dma-wait %llu barrier %llu | ktiles %llu blocks %llu\n",
              hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7]);
      if (hb[18])
        fprintf(stderr, "[pt timing, split form, wave 0 cycles] prologue %llu loop %llu epilogue+drain %llu | per step: dma-issue %llu wait-Wg %llu gen pass %llu wait-Wd %llu "
                "down pass + cut %llu barrier %llu | ktiles %llu blocks %llu\n", hb[8], hb[9], hb[10], hb[11], hb[12], hb[13], hb[14], hb[15], hb[16], hb[17], hb[18]);
      if (hb[24])
        fprintf(stderr, "[pt timing, producer / consumer form, cycles of producer 0 / consumer 0 of a block] producer: wait for loads %llu cut + issue %llu barrier %llu | "
                "consumer: steps %llu barrier %llu item boundary (locate + epilogue) %llu | tiles %llu blocks %llu items %llu\n",
                hb[20], hb[21], hb[22], hb[25], hb[26], hb[27], hb[23], hb[24], hb[28]);
      (void)hipMemset(dbg, 0, 256);
    }
  }
#endif
  constexpr size_t kStage32 = (size_t)(PT_BM + PT_BN) * LDS_K * 4;
  constexpr int kNT = 0;     // product default (tools/sweep_pw.py, profiles/r02)
  // exact fp32 + the library's own weight image: always the 16-pixel LDS-DMA form
#define OFFK_PT_LAUNCH_BD                                                                                            \
  {                                                                                                                  \
    constexpr int k16Lds = 2 * PT_FT * 32 * 64 + 1024;                                                               \
    hipError_t er = lds_attr_once(reinterpret_cast<const void*>(pw_tdiff16_kernel), k16Lds);                         \
    if (er != hipSuccess) return er;                                                                                 \
    hipLaunchKernelGGL(pw_tdiff16_kernel, dim3(p.total_blocks), dim3(256), k16Lds, st, p);                           \
  }
  // buffer addressing needs every byte offset below 2^31: each feature-map part and the weight tables
  bool lean = true;
  for (int i = 0; i < p.nsites; ++i) {
    for (int q = 0; q < p.s[i].nparts; ++q)
      if ((unsigned long long)p.B * p.L * p.s[i].cp[q] * p.s[i].HW * 4ull >= 0x7fffff00ull) lean = false;
    if ((unsigned long long)kGenCh * p.s[i].C * 4ull >= 0x7fffff00ull) lean = false;
  }
  bool bd = lean && p.bdirect;
  bool pack = lean;          // leftover pixels of several clips in one block: per-thread clip offsets need the buffer loader
#ifdef OFFK_TUNING_KNOBS
  { const char* pe = getenv("OFFK_PW_PACK"); if (pe && *pe == '0') pack = false; }
  { const char* be = getenv("OFFK_PW_BDIRECT"); if (be && *be == '0') bd = false; }
  { const char* le = getenv("OFFK_PW_LEAN"); if (le && !(atoi(le) & 1)) lean = bd = pack = false; }
#endif
  const bool form16 = bd;      // the library's own weight image: the 16-pixel forms (pw_tdiff16_kernel / pw_tdiff_split_kernel)
  auto layout = [&]() {      // block layout of every site (PtSite)
    if (form16) {
      int blk = 0;
      for (int i = 0; i < p.nsites; ++i) {
        PtSite& o = p.s[i];
        const int rem = o.HW % 16;
        const bool packed = rem == 4 || rem == 8;        // 14x14: four clips' four leftover pixels per block
        const int qpc = (o.HW + 3) / 4;
        // 7x7: 49 = 3 x 16 + 1 -- a fourth chunk per clip would multiply one pixel; as a stream of quads 64 clips are 208 blocks, not 256
        const bool stream = qstream && !packed && rem != 0 && qpc >= 4 && (long long)p.B * qpc < (1 << 22);
        o.qpc = stream ? qpc : 0;
        o.chunks = stream ? 0 : packed ? o.HW / 16 : (o.HW + 15) / 16;
        o.rsh = packed ? (rem == 4 ? 2 : 3) : 4;
        o.nrem = stream ? (p.B * qpc + 3) / 4 : packed ? (p.B + (16 >> o.rsh) - 1) / (16 >> o.rsh) : 0;
        o.blk_begin = blk;
        blk += (p.B * o.chunks + o.nrem) * p.tgroups;
      }
      p.total_blocks = blk;
      return;
    }
    int blk = 0;
    for (int i = 0; i < p.nsites; ++i) {
      PtSite& o = p.s[i];
      const int rem = o.HW % 32;
      // 16-byte pieces (4 leftover pixels per clip) cost a whole sector each: worth it where the kernel is MFMA-bound (1.45 -> 1.36 ms)
      const bool packed = pack && (rem == 16 || rem == 4 || rem == 8);
      o.chunks = packed ? o.HW / 32 : (o.HW + 31) / 32;
      o.rsh = packed ? (rem == 4 ? 2 : rem == 8 ? 3 : 4) : 5;
      o.nrem = packed ? (p.B + (32 >> o.rsh) - 1) / (32 >> o.rsh) : 0;
      o.qpc = 0;
      o.blk_begin = blk;
      blk += (p.B * o.chunks + o.nrem) * p.tgroups;
    }
    p.total_blocks = blk;
  };
  layout();
  // (not form16: the fallback kernel -- buffer-addressed loader where every tensor is below 2^31 bytes, pointer loader otherwise)
#ifdef OFFK_TUNING_KNOBS
  const char* e = getenv("OFFK_PW_NT");
  const int nt = e ? atoi(e) : kNT;
#define OFFK_PT_LAUNCH(LDS)                                                                                       \
  if (bd) OFFK_PT_LAUNCH_BD                                                                                       \
  else if (lean) hipLaunchKernelGGL((pw_tdiff_kernel<0, 1>), dim3(p.total_blocks), dim3(256), LDS, st, p);        \
  else switch (nt & 3) {                                                                                          \
    case 0: hipLaunchKernelGGL((pw_tdiff_kernel<0, 0>), dim3(p.total_blocks), dim3(256), LDS, st, p); break;      \
    case 1: hipLaunchKernelGGL((pw_tdiff_kernel<1, 0>), dim3(p.total_blocks), dim3(256), LDS, st, p); break;      \
    case 2: hipLaunchKernelGGL((pw_tdiff_kernel<2, 0>), dim3(p.total_blocks), dim3(256), LDS, st, p); break;      \
    default: hipLaunchKernelGGL((pw_tdiff_kernel<3, 0>), dim3(p.total_blocks), dim3(256), LDS, st, p); break;     \
  }
#else
#define OFFK_PT_LAUNCH(LDS)                                                                                       \
  if (bd) OFFK_PT_LAUNCH_BD                                                                                       \
  else if (lean) hipLaunchKernelGGL((pw_tdiff_kernel<kNT, 1>), dim3(p.total_blocks), dim3(256), LDS, st, p);      \
  else hipLaunchKernelGGL((pw_tdiff_kernel<kNT, 0>), dim3(p.total_blocks), dim3(256), LDS, st, p);
#endif
  if (form16 && p.f32split) {
    return pw_tdiff_split_launch(p, st);
  }
  OFFK_PT_LAUNCH(kStage32)
#undef OFFK_PT_LAUNCH
#undef OFFK_PT_LAUNCH_BD
  return hipGetLastError();
}

}  // namespace offk
