// K1: the OFF units' two 1x1 "reduce" convolutions, stacked and grouped.
//
// Stands for, per tap site s (reference RGB_OFF.py, site 3a lines; the other eight are
// identical up to names):
//   G = relu(motion_conv_gen_s(X))            :597-598  on all N = B*L frames
//   D = motion_spatial_down_s(X[:B*(L-1)])    :609-610  on the sliced frames (quirk Q1)
// Both are C -> {128, 32} pointwise contractions of the same feature map, so their
// weights are stacked into one [160][C] matrix and X is read from HBM once.  All nine
// sites run in ONE grouped launch (a block looks its site up in a 9-entry table), so the
// 7x7 sites (172 blocks at B=64) fill the chip together with the 28x28 ones.
//
// GEMM view: rows m = (frame, pixel) flattened, K = C, N = 160.  X is NCHW at the
// boundary ([frame][c][pixel]: for one k the rows are contiguous), so the loader reads
// 4 k-rows x 4 pixels per thread with 16-B loads and transposes in registers into the
// [row][k] LDS image the MFMA core wants.  Outputs are channels-last:
//   G [N*HW][128] (post-ReLU), D [P*HW][32] (row = output pair index).
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

constexpr int PW_BM = 128, PW_BN = 160, PW_TN = 5;

int pw_blocks_for(int M) { return (M + PW_BM - 1) / PW_BM; }

// output-pair row of frame f for the spatial branch, or -1 (see spatial_frames in the oracle)
__device__ __forceinline__ int down_row(int f, int L, int P, int slice_mode) {
  if (slice_mode == 0) return f < P ? f : -1;
  int b = f / L, t = f - b * L;
  return t < L - 1 ? b * (L - 1) + t : -1;
}

template <int NT>
__device__ __forceinline__ void pw_mma(f32x16 (&acc)[PW_TN], const float* As, const float* Bs, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll 1
  for (int g = 0; g < BK / 8; ++g) {
    float4 a = *reinterpret_cast<const float4*>(As + r * LDS_K + 8 * g + 4 * h);
    float4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + (t * 32 + r) * LDS_K + 8 * g + 4 * h);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
  }
}

typedef float f4v __attribute__((ext_vector_type(4)));
// NT bit 0: the feature map is read exactly once -> non-temporal loads (weights keep the default policy: every block
// re-reads them); bit 1: G / D are not read again before K1 ends -> non-temporal stores.
template <int NT>
__device__ __forceinline__ float4 ldx4(const float* q) {
  if (NT & 1) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(q);
}
template <int NT>
__device__ __forceinline__ void stf(float* q, float v) {
  if (NT & 2) __builtin_nontemporal_store(v, q);
  else *q = v;
}

// LEAN: buffer-descriptor addressing of the weights and of the NCHW quad loader (mode 0), unconditional prefetch pinned
// in front of the MFMAs -- as pw_tdiff.hip; needs every feature-map part and the weight tables below 2^31 bytes.
template <int NT, int LEAN>
__global__ __launch_bounds__(256, 2) void pw_reduce_kernel(PwParams p) {
  constexpr int LDS_BYTES = (PW_BM + PW_BN) * LDS_K * 4;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  float* As = reinterpret_cast<float*>(lds);                    // fp32: [128][LDS_K] then [160][LDS_K]
  float* Bs = As + PW_BM * LDS_K;

  // block -> site: field-wise scalar select chain over the (few) table entries; indexing
  // the by-value kernarg array with a runtime index (or copying a whole entry) goes
  // through scratch memory
  PwSite S;
  int nblk_site;
#define OFFK_PW_PICK(i)                                                                              \
  S.w = p.s[i].w; S.bias = p.s[i].bias; S.G = p.s[i].G; S.D = p.s[i].D;                              \
  S.w_down = p.s[i].w_down; S.bias_down = p.s[i].bias_down;                                          \
  S.C = p.s[i].C; S.HW = p.s[i].HW; S.M = p.s[i].M; S.blk_begin = p.s[i].blk_begin;                  \
  nblk_site = (i + 1 < p.nsites ? p.s[i + 1].blk_begin : p.total_blocks) - p.s[i].blk_begin;         \
  S.nparts = p.s[i].nparts;                                                                          \
  S.xp[0] = p.s[i].xp[0]; S.xp[1] = p.s[i].xp[1]; S.xp[2] = p.s[i].xp[2]; S.xp[3] = p.s[i].xp[3];    \
  S.cp[0] = p.s[i].cp[0]; S.cp[1] = p.s[i].cp[1]; S.cp[2] = p.s[i].cp[2]; S.cp[3] = p.s[i].cp[3];
  OFFK_PW_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.x >= p.s[i].blk_begin) { OFFK_PW_PICK(i) }
#undef OFFK_PW_PICK
  const int C = S.C, HW = S.HW, M = S.M;
  const int m0 = xcd_contiguous((int)blockIdx.x - S.blk_begin, nblk_site) * PW_BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // is any row of this block a source frame of the spatial branch?
  bool down_active;
  {
    int f_first = m0 / HW;
    int m_last = min(m0 + PW_BM, M) - 1;
    int f_last = m_last / HW;
    down_active = false;
    for (int f = f_first; f <= f_last; ++f) down_active |= down_row(f, p.L, p.P, p.slice_mode) >= 0;
  }

  // ---- loader setup -----------------------------------------------------------
  // mode 0: NCHW, HW % 4 == 0 : thread = (k quad kq = tid&7, pixel quad pq = tid>>3): the lanes of a 16-lane LDS
  //         store group cover the whole 32-k span of two rows (conflict-free); with pq on the low lane bits the
  //         8-byte bf16 stores of a group fell into 4 bank positions (60 % of the LDS cycles were conflicts)
  // mode 1: NCHW, any HW      : thread = (pixel i = tid&127, k quads (tid>>7) + 2r)
  // mode 2: channels-last     : thread = (rows (tid>>3)+32r, k quad tid&7)
  const int mode = p.nhwc ? 2 : ((HW & 3) == 0 ? 0 : 1);
  int fr = 0, pix = 0;       // this thread's frame and pixel (modes 0 / 1)
  bool row_ok = true;
  if (mode == 0) {
    int m = m0 + 4 * (tid >> 3);
    row_ok = m < M;                       // M % 4 == 0 here, so the quad is all-in or all-out
    int mm = row_ok ? m : 0;
    fr = mm / HW; pix = mm - fr * HW;
  } else if (mode == 1) {
    int m = m0 + (tid & 127);
    row_ok = m < M;
    int mm = row_ok ? m : 0;
    fr = mm / HW; pix = mm - fr * HW;
  }
  const int koff = mode == 0 ? 4 * (tid & 7) : (mode == 1 ? 4 * (tid >> 7) : 4 * (tid & 7));
  // channel k0 (a multiple of 32) -> (part, channel within the part); parts are whole multiples of 32
  // channels, so a K-tile never straddles two of them.  Uniform across the block: scalar selects.
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

  float4 rg[4 + PW_TN];   // prefetch registers: 4 x A, then 5 x B (one array: two arrays end up in scratch)
  // weight rows (tid>>3) + 32 r: r = 0..3 gen rows, r = 4 the down rows (their own pointer: bound parameters need not be adjacent)
  const float* wbase = S.w + (size_t)(tid >> 3) * C + 4 * (tid & 7);
  const float* wdbase = S.w_down + (size_t)(tid >> 3) * C + 4 * (tid & 7);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t wrs, wdrs;
  int woff = 0, vA = 0;
  if constexpr (LEAN) {
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.w), 0, kGenCh * C * 4, 0x00020000);
    wdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.w_down), 0, kDownCh * C * 4, 0x00020000);
    woff = ((tid >> 3) * C + 4 * (tid & 7)) * 4;
    vA = row_ok ? (koff * HW + pix) * 4 : (int)0x80000000;     // rows past M: an offset past every descriptor -> zeros
    if (!row_ok) fr = 0;
  }
  auto ld_b = [&](const __amdgpu_buffer_rsrc_t& rs, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  auto load_tile = [&](int k0) {
    const float* xb; int cpart, kl;
    locate(k0, xb, cpart, kl);
#ifdef OFFK_TUNING_KNOBS
    if (p.ablate & 1) {
    } else
#endif
    if (LEAN && mode == 0) {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (M / HW) * cpart * HW * 4, 0x00020000);
      const int voff = fr * (cpart * HW * 4) + vA;
#pragma unroll
      for (int j = 0; j < 4; ++j) rg[j] = ld_b(xrs, voff, (kl + j) * HW * 4);
    } else if (mode == 0) {
      // (the branch-free form -- masked rows reading a zero page through a selected pointer -- measured 5 % SLOWER here,
      // same box A/B in round 2: row_ok is false only in the last tile of a site and the branch is cheap)
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* base = xb + ((size_t)fr * cpart + kl + koff) * HW + pix;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rg[j] = z;
        if (row_ok) rg[j] = ldx4<NT>(base + (size_t)j * HW);
      }
    } else if (mode == 1) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* base = xb + ((size_t)fr * cpart + kl + koff) * HW + pix;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* q = base + (size_t)(8 * r) * HW;
        rg[r] = z;
        if (row_ok) rg[r] = make_float4(q[0], q[HW], q[2 * (size_t)HW], q[3 * (size_t)HW]);
      }
    } else {
      const float* base = xb + (size_t)(m0 + (tid >> 3)) * cpart + kl + koff;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rg[r] = z;
        if (m0 + (tid >> 3) + 32 * r < M) rg[r] = *reinterpret_cast<const float4*>(base + (size_t)32 * r * cpart);
      }
    }
#ifdef OFFK_TUNING_KNOBS
    if (p.ablate & 2) return;
#endif
    if constexpr (LEAN) {
#pragma unroll
      for (int r = 0; r < PW_TN - 1; ++r) rg[4 + r] = ld_b(wrs, woff, (32 * r * C + k0) * 4);
      rg[4 + PW_TN - 1] = ld_b(wdrs, woff, k0 * 4);
      return;
    }
#pragma unroll
    for (int r = 0; r < PW_TN - 1; ++r)
      rg[4 + r] = *reinterpret_cast<const float4*>(wbase + (size_t)32 * r * C + k0);
    rg[4 + PW_TN - 1] = *reinterpret_cast<const float4*>(wdbase + k0);
  };
  auto store_tile = [&]() {
#ifdef OFFK_TUNING_KNOBS
    if (p.ablate & 4) return;
#endif
    if (mode == 0) {
      float* dst = As + 4 * (tid >> 3) * LDS_K + 4 * (tid & 7);
      *reinterpret_cast<float4*>(dst) = make_float4(rg[0].x, rg[1].x, rg[2].x, rg[3].x);
      *reinterpret_cast<float4*>(dst + LDS_K) = make_float4(rg[0].y, rg[1].y, rg[2].y, rg[3].y);
      *reinterpret_cast<float4*>(dst + 2 * LDS_K) = make_float4(rg[0].z, rg[1].z, rg[2].z, rg[3].z);
      *reinterpret_cast<float4*>(dst + 3 * LDS_K) = make_float4(rg[0].w, rg[1].w, rg[2].w, rg[3].w);
    } else if (mode == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(As + (tid & 127) * LDS_K + 4 * ((tid >> 7) + 2 * r)) = rg[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[r];
    }
#pragma unroll
    for (int r = 0; r < PW_TN; ++r)
      *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[4 + r];
  };

  f32x16 acc[PW_TN];
#pragma unroll
  for (int t = 0; t < PW_TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nkt = C / BK;
  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
    store_tile();
    __syncthreads();
    if constexpr (LEAN) {
      load_tile(min(kt + 1, nkt - 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (kt + 1 < nkt) load_tile((kt + 1) * BK);
    }
#ifdef OFFK_TUNING_KNOBS
    if (p.ablate & 8) {
    } else
#endif
    {
      if (down_active) pw_mma<5>(acc, As + wave * 32 * LDS_K, Bs, lane);
      else pw_mma<4>(acc, As + wave * 32 * LDS_K, Bs, lane);
    }
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------
  const int r32 = lane & 31, h = lane >> 5;
  float bv[PW_TN];
#pragma unroll
  for (int t = 0; t < PW_TN - 1; ++t) bv[t] = S.bias[t * 32 + r32];
  bv[PW_TN - 1] = S.bias_down[r32];
  // bias / ReLU in straight-line code first: a loaded value (the bias) used inside the `if (row in range)` blocks
  // makes every block start with an s_waitcnt that also throttles the stores of the blocks before it
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = fmaxf(acc[t][reg] + bv[t], 0.f);
      asm volatile("" : "+v"(v));
      acc[t][reg] = v;
    }
    float d = acc[4][reg] + bv[4];
    asm volatile("" : "+v"(d));
    acc[4][reg] = d;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + wave * 32 + acc_row(reg, h);
    if (m < M) {
      float* g = S.G + (size_t)m * kGenCh + r32;
#pragma unroll
      for (int t = 0; t < 4; ++t) stf<NT>(g + t * 32, acc[t][reg]);
      if (down_active) {
        int f = m / HW, pix = m - f * HW;
        int dr = down_row(f, p.L, p.P, p.slice_mode);
        if (dr >= 0) stf<NT>(S.D + ((size_t)dr * HW + pix) * kDownCh + r32, acc[4][reg]);
      }
    }
  }
}

hipError_t pw_reduce_launch(const PwParams& p_in, hipStream_t st) {
  PwParams p = p_in;
#ifdef OFFK_TUNING_KNOBS
  { const char* e = getenv("OFFK_PW_ABLATE"); p.ablate = e ? atoi(e) : 0; }
#endif
  if (p.total_blocks <= 0) return hipSuccess;
  constexpr int kNT = 0;     // product default (tools/sweep_pw.py, profiles/r02)
  bool lean = true;          // buffer addressing: every byte offset below 2^31
  for (int i = 0; i < p.nsites; ++i) {
    for (int q = 0; q < p.s[i].nparts; ++q)
      if ((unsigned long long)p.s[i].M * p.s[i].cp[q] * 4ull >= 0x7fffff00ull) lean = false;
    if ((unsigned long long)kGenCh * p.s[i].C * 4ull >= 0x7fffff00ull) lean = false;
  }
#ifdef OFFK_TUNING_KNOBS
  const char* e = getenv("OFFK_PW_NT");
  const int nt = e ? atoi(e) : kNT;
  { const char* le = getenv("OFFK_PW_LEAN"); if (le && !(atoi(le) & 1)) lean = false; }
  if (lean) hipLaunchKernelGGL((pw_reduce_kernel<0, 1>), dim3(p.total_blocks), dim3(256), 0, st, p);
  else switch (nt & 3) {
    case 0: hipLaunchKernelGGL((pw_reduce_kernel<0, 0>), dim3(p.total_blocks), dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((pw_reduce_kernel<1, 0>), dim3(p.total_blocks), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((pw_reduce_kernel<2, 0>), dim3(p.total_blocks), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((pw_reduce_kernel<3, 0>), dim3(p.total_blocks), dim3(256), 0, st, p); break;
  }
#else
  if (lean) hipLaunchKernelGGL((pw_reduce_kernel<kNT, 1>), dim3(p.total_blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((pw_reduce_kernel<kNT, 0>), dim3(p.total_blocks), dim3(256), 0, st, p);
#endif
  return hipGetLastError();
}

}  // namespace offk
