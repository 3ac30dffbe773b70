// K1T in split-fp32 arithmetic (round 5): the fused units kernel of pw_tdiff.hip (1x1 reduces + ReLU + temporal difference + the
// down rows, reference RGB_OFF.py:597-610) with every fp32 operand cut into THREE bf16 planes and multiplied on the bf16 matrix pipe.
//
// Arithmetic.  x = x_h + x_m + x_l exactly (8 + 8 + 8 significand bits: x_h = the upper 16 bits of x, the remainder is exact in
// fp32, cut again, and again), the same for w; every product of two planes is exact in fp32.  Of the nine plane products the six
// above 2^-24 of the leading one are formed -- w_l x_h, w_h x_l, w_m x_m, w_m x_h, w_h x_m, w_h x_h, smallest first -- on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The six products of a 32-k step are summed in a scratch tile that starts at
// ZERO and are added to the running accumulator once per step: the running sum is rounded once per 32 k (the fp32 pipe's
// v_mfma_f32_16x16x4_f32 rounds it eight times per 32 k), the small products are rounded against a 32-k partial sum instead of
// against the whole accumulator.  Measured against fp64 on MI355X (tools/probe_split_mfma.hip, profiles/r05/probe_split_mfma.txt):
// max and rms error and the backward-error constant of |err| <= c 2^-24 sum|w x| are 2 - 2.5 x SMALLER than the fp32 pipe's on
// every distribution the parity tests use (synthetic, full mantissa, heavy tail, cancellation).  One instruction sums the eight
// products of a lane group wide and adds the four lane groups to the accumulator one after the other (same probe, part 2).
//
// Why: the fp32 pipe peaks at 157 TF and pw_tdiff16_kernel holds 0.89 of it; the bf16 pipe sustains ~2.1 PF on these operands
// (16x16x32 form, clock under load included) -> six products = ~2.3 x the fp32 rate.
//
// Structure (one block = (clip, 16-pixel chunk) x seven frames x 160 channels, as pw_tdiff16_kernel; two blocks per CU):
//   * wave w DMAs its pieces of K-tile k + 2 (a piece = one frame's 16 k rows x 16 pixels fp32, 1 KB) into its PRIVATE raw
//     region, and in step k cuts the pieces of tile k + 1 it loaded itself into the three plane images of that tile -- no barrier
//     between DMA and cut, the cut's ~26 vector instructions per piece ride in the issue slots the MFMAs of tile k leave free;
//   * plane image per (frame, plane): [4 k groups][16 pixels] x 16 B (8 bf16 = k 8g .. 8g + 7): the MFMA's B operand is one
//     conflict-free ds_read_b128 per lane; the raw piece is laid out by the DMA's per-lane global offsets so that the cut's
//     ds_read_b32 are conflict-free too ([k & 3][k >> 2][pixel quad]);
//   * weights: the library's plane image in operand order straight from L2 into registers (pw_pack_split16_kernel), gen set
//     double-buffered a whole K-tile ahead, down set reloaded behind its last use; hand-counted waits (each step issues exactly
//     6 + 4 + 6 vector-memory operations per wave: Wg(k + 1), DMA(k + 2), ..., Wd(k + 1));
//   * one barrier per K-tile (the plane images change hands).
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
constexpr int PS_FT = 7;                              // frames per block
constexpr int PS_PIECE = 1024;                        // raw piece: 16 k rows x 16 pixels fp32
constexpr int PS_RAW_WAVE = 4 * PS_PIECE;             // a wave's four DMA slots of one K-tile
constexpr int PS_RAW_STAGE = 4 * PS_RAW_WAVE;         // 16 KB
constexpr int PS_PLANE = 1024;                        // [4 g][16 px] x 16 B
constexpr int PS_FRAME = 3 * PS_PLANE;
constexpr int PS_PL_STAGE = PS_FT * PS_FRAME;         // 21 KB
constexpr int PS_LDS = 2 * PS_RAW_STAGE + 2 * PS_PL_STAGE;      // 75776 B = 60 granules of 1280 B: two blocks per CU

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int ps_down_row(int b, int t, int L, int P, int slice_mode) {
  if (slice_mode == 0) { const int f = b * L + t; return f < P ? f : -1; }
  return t < L - 1 ? b * (L - 1) + t : -1;
}
}  // namespace

__global__ __launch_bounds__(256, 2) void pw_tdiff_split_kernel(PtParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];     // raw[2][4 waves][4 pieces] | planes[2][7 frames][3][1 KB]
  char* const planes = lds + 2 * PS_RAW_STAGE;

  // ---- the block's site and its (clip, pixels, temporal group): as pw_tdiff16_kernel ----
  int si = 0;
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.x >= p.s[i].blk_begin) si = i;
  si = __builtin_amdgcn_readfirstlane(si);
  const PtSite& S0 = p.s[si];
  PtSite S;
  S.bias = S0.bias; S.D = S0.D; S.M = S0.M; S.m_cs = S0.m_cs; S.bias_down = S0.bias_down; S.wt = static_cast<const float*>(S0.wt16s);
  S.m_coff = S0.m_coff; S.C = S0.C; S.HW = S0.HW; S.chunks = S0.chunks; S.nrem = S0.nrem; S.rsh = S0.rsh;
  S.blk_begin = S0.blk_begin; S.nparts = S0.nparts; S.qpc = S0.qpc;
#pragma unroll
  for (int q = 0; q < 4; ++q) { S.xp[q] = S0.xp[q]; S.cp[q] = S0.cp[q]; }
  const int nblk_site = (si + 1 < p.nsites ? p.s[si + 1].blk_begin : p.total_blocks) - S.blk_begin;
  const int C = S.C, HW = S.HW, L = p.L;
  int local = xcd_contiguous((int)blockIdx.x - S.blk_begin, nblk_site);
  const int tg = local % p.tgroups; local /= p.tgroups;
  const int nfull = p.B * S.chunks;
  const bool leftover = local >= nfull;
  const int rsh = leftover ? S.rsh : 4, rmask = (1 << rsh) - 1;
  const int qpc = S.qpc;
  const int b = qpc ? (4 * local) / qpc : leftover ? (local - nfull) << (4 - rsh) : local / S.chunks;
  const int qr0 = qpc ? 4 * local - b * qpc : 0;
  const int q0 = (leftover ? S.chunks : local - b * S.chunks) * 16;
  const int t0 = tg * (PS_FT - 1);
  const int nf = min(PS_FT, L - t0);
  const bool last_group = tg == p.tgroups - 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, lg = lane >> 4;

  // ---- DMA: slot q = wave + 4 i (i = 0..3) = frame q >> 1, k half q & 1 (= wave & 1); slots 14, 15 read zeros.  Lane l of a piece
  //      fetches k row 4 a + r (a = (l >> 2) & 3, r = l >> 4), pixel quad l & 3: row k sits at [k & 3][k >> 2] of the piece ----
  const int pq = lane & 3;
  const bool qnext = qpc && qr0 + pq >= qpc;
  const int cq = qpc ? (int)qnext : (4 * pq) >> rsh;
  const int k0px = qpc ? 4 * (qr0 + pq - (qnext ? qpc : 0)) : q0 + ((4 * pq) & rmask);
  const bool px_ok = k0px < HW && b + cq < p.B;
  const int vrow0 = px_ok ? ((16 * (wave & 1) + 4 * ((lane >> 2) & 3) + (lane >> 4)) * HW + k0px) * 4 : (int)0x80000000;
  const int vclip = cq * L;
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
    dm_voff = vrow0 + (int)__umul24((unsigned)vclip, (unsigned)dm_fstride);
  };
  auto dma_tile = [&](const int rs) {       // the wave's four slots of the prepared K-tile into raw stage rs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave + 4 * i, fr = q >> 1;
      const int voff = fr < nf ? dm_voff : (int)0x80000000;            // (slots 14, 15: fr = 7 >= nf)
      dma16(dm_desc, lds_base + rs * PS_RAW_STAGE + wave * PS_RAW_WAVE + i * PS_PIECE, voff, dm_s0 + fr * dm_fstride);
    }
  };

  // ---- the cut: lane (pixel li, a = lg) of piece i holds k = 16 (wave & 1) + 4 a + 0..3 of frame (wave + 4 i) >> 1 ----
  const char* const raw_rd = lds + wave * PS_RAW_WAVE + lg * 64 + li * 4;          // + stage, + piece, + r * 256
  // planes: g = 2 (wave & 1) + (a >> 1), 8-byte half a & 1 of the lane's 16 B
  char* const pl_wr = planes + (2 * (wave & 1) + (lg >> 1)) * 256 + li * 16 + (lg & 1) * 8;   // + stage, + frame, + plane
  char* const pl_dummy = lds + wave * PS_RAW_WAVE + 3 * PS_PIECE + (pl_wr - planes);     // + raw stage: the wave's own fourth piece
  auto cut_piece = [&](const int i, const int rs, const int ps) {
    const int fr = (wave + 4 * i) >> 1;      // slots 14, 15 (fr = 7: zeros) are cut like the others -- no branch in the MFMA stream (hipcc sinks
                                             // the accumulator updates across any block boundary) -- and land on the piece itself
    const char* src = raw_rd + rs * PS_RAW_STAGE + i * PS_PIECE;
    unsigned x[4], h[4], m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = *reinterpret_cast<const unsigned*>(src + r * 256);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h[r] = x[r] & 0xffff0000u;
      const float r1 = __uint_as_float(x[r]) - __uint_as_float(h[r]);
      m[r] = __float_as_uint(r1) & 0xffff0000u;
      l[r] = __float_as_uint(r1 - __uint_as_float(m[r]));               // <= 8 significant bits: its low half is zero
    }
    const bool real = i < 3 || fr < PS_FT;
    char* dst = real ? pl_wr + ps * PS_PL_STAGE + fr * PS_FRAME : pl_dummy + rs * PS_RAW_STAGE;
    const int pstride = real ? PS_PLANE : 0;
    *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302), __builtin_amdgcn_perm(h[3], h[2], 0x07060302)};
    *reinterpret_cast<u32x2*>(dst + pstride) = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302), __builtin_amdgcn_perm(m[3], m[2], 0x07060302)};
    *reinterpret_cast<u32x2*>(dst + 2 * pstride) = u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302), __builtin_amdgcn_perm(l[3], l[2], 0x07060302)};
  };

  // ---- weights: image [kt][slab (4 gen + 1 down)][ct][plane][lane] x 16 B (pw_pack_split16_kernel) ----
  u32x4 wg[2][3], wd[2][3];                // gen [ct][plane], down [ct][plane]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) { wg[c][q] = u32x4{0u, 0u, 0u, 0u}; wd[c][q] = u32x4{0u, 0u, 0u, 0u}; }
  i32x4 wdesc;
  {
    const unsigned long long wa = reinterpret_cast<unsigned long long>(S.wt);
    wdesc = i32x4{(int)(unsigned)wa, (int)(unsigned)(wa >> 32) & 0xffff, kUnitCh * C * 6, 0x00020000};
  }
  const int wlane = lane * 16;
  auto load_wg = [&](int kt) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int so = (((kt * 5 + wave) * 2 + ct) * 3 + q) * 1024;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wg[ct][q]) : "v"(wlane), "s"(wdesc), "s"(so));
      }
  };
  auto load_wd = [&](int kt) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int so = (((kt * 5 + 4) * 2 + ct) * 3 + q) * 1024;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wd[ct][q]) : "v"(wlane), "s"(wdesc), "s"(so));
      }
  };
#define OFFK_WAIT_WG(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wg[0][0]), "+v"(wg[0][1]), "+v"(wg[0][2]), \
                                     "+v"(wg[1][0]), "+v"(wg[1][1]), "+v"(wg[1][2]) :: "memory")
#define OFFK_WAIT_WD(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wd[0][0]), "+v"(wd[0][1]), "+v"(wd[0][2]), \
                                     "+v"(wd[1][0]), "+v"(wd[1][1]), "+v"(wd[1][2]) :: "memory")

  f32x4 ag[PS_FT][2], ad[2][2];
#pragma unroll
  for (int j = 0; j < PS_FT; ++j) { ag[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 2; ++i) { ad[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ad[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // B operand of frame f out of plane stage st: three ds_read_b128
  const char* const xrd = planes + lg * 256 + li * 16;
  const int xoffA = wave * PS_FRAME, xoffB = min(wave + 4, PS_FT - 1) * PS_FRAME;
  auto rdx = [&](u32x4 (&x)[3], const int st, int foff) {
#pragma unroll
    for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const u32x4*>(xrd + st * PS_PL_STAGE + foff + q * PS_PLANE);
  };
  auto mf = [&](f32x4 c, const u32x4& a, const u32x4& bb) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0);
  };
  // the six products of one (frame, two channel tiles) unit, the two tiles' chains alternating; planes: 0 = h, 1 = m, 2 = l
  auto unit = [&](f32x4& t0, f32x4& t1, const u32x4 (&w0)[3], const u32x4 (&w1)[3], const u32x4 (&x)[3]) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    t0 = mf(z, w0[2], x[0]);  t1 = mf(z, w1[2], x[0]);
    t0 = mf(t0, w0[0], x[2]); t1 = mf(t1, w1[0], x[2]);
    t0 = mf(t0, w0[1], x[1]); t1 = mf(t1, w1[1], x[1]);
    t0 = mf(t0, w0[1], x[0]); t1 = mf(t1, w1[1], x[0]);
    t0 = mf(t0, w0[0], x[1]); t1 = mf(t1, w1[0], x[1]);
    t0 = mf(t0, w0[0], x[0]); t1 = mf(t1, w1[0], x[0]);
  };

  const int nkt = C / BK;
  // one step: tile kt out of plane stage ST; the cut of tile kt + 1 from raw stage ST ^ 1 into plane stage ST ^ 1; DMA of tile kt + 2
  // into raw stage ST (its tile kt was cut during the last step).  Vector-memory operations per wave and step, in issue order:
  // DMA(kt + 2) [4], Wg(kt + 1) [6] behind the gen pass, Wd(kt + 1) [6] behind the down pass.
  auto step = [&](int kt, const int ST) {
    const int kn = min(kt + 1, nkt - 1);
    dma_prep(min(kt + 2, nkt - 1));
    dma_tile(ST);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 x[2][3];
    f32x4 t[2][2];
    rdx(x[0], ST, 0);
    OFFK_WAIT_WG(10);                         // Wg(kt): all but Wd(kt) [6] and this step's DMAs [4]
    // unit j: the B operand of unit j + 1 is read first; the scratch tiles of unit j - 1 are added to their accumulators beside
    // unit j's MFMAs (an MFMA's result is ~8 issue slots away)
#pragma unroll
    for (int j = 0; j < PS_FT; ++j) {
      if (j + 1 < PS_FT) rdx(x[(j + 1) & 1], ST, (j + 1) * PS_FRAME);
      else rdx(x[(j + 1) & 1], ST, xoffA);
      __builtin_amdgcn_sched_barrier(0);
      unit(t[j & 1][0], t[j & 1][1], wg[0], wg[1], x[j & 1]);
      if (j > 0) {
        ag[j - 1][0] += t[(j - 1) & 1][0]; ag[j - 1][1] += t[(j - 1) & 1][1];
        asm volatile("" : "+v"(ag[j - 1][0]), "+v"(ag[j - 1][1]));       // the update stays here
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    load_wg(kn);
    // the down tiles: frames `wave` (x[1]) and wave + 4 (wave 3: frame 6 again, never stored); beside them the cut of tile kt + 1
    OFFK_WAIT_WD(10);                         // Wd(kt): all but this step's DMAs [4] and Wg(kt + 1) [6]; older: DMA(kt + 1) -- landed too
    rdx(x[0], ST, xoffB);
    __builtin_amdgcn_sched_barrier(0);
    unit(t[1][0], t[1][1], wd[0], wd[1], x[1]);
    ag[PS_FT - 1][0] += t[0][0]; ag[PS_FT - 1][1] += t[0][1];
    asm volatile("" : "+v"(ag[PS_FT - 1][0]), "+v"(ag[PS_FT - 1][1]));
    cut_piece(0, ST ^ 1, ST ^ 1);
    cut_piece(1, ST ^ 1, ST ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    unit(t[0][0], t[0][1], wd[0], wd[1], x[0]);
    ad[0][0] += t[1][0]; ad[0][1] += t[1][1];
    asm volatile("" : "+v"(ad[0][0]), "+v"(ad[0][1]));
    cut_piece(2, ST ^ 1, ST ^ 1);
    cut_piece(3, ST ^ 1, ST ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    ad[1][0] += t[0][0]; ad[1][1] += t[0][1];
    asm volatile("" : "+v"(ad[1][0]), "+v"(ad[1][1]));
    load_wd(kn);
    __syncthreads();                         // (the compiler's lgkmcnt(0) in front of it covers the plane writes)
  };

  // ---- prologue: DMA(0), DMA(1), Wg(0), Wd(0) -- the order the steps' wait counts assume ----
  dma_prep(0);
  dma_tile(0);
  dma_prep(min(1, nkt - 1));
  dma_tile(1);
  load_wg(0);
  load_wd(0);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // DMA(0)
#pragma unroll
  for (int i = 0; i < 4; ++i) cut_piece(i, 0, 0);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    step(kt, 0);
    step(kt + 1, 1);
  }
  if (kt < nkt) step(kt, 0);
  // nothing may still be landing when the LDS is handed on; the weight registers stay allocated until their last load returned
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[0][0]), "+v"(wg[0][1]), "+v"(wg[0][2]), "+v"(wg[1][0]), "+v"(wg[1][1]), "+v"(wg[1][2]),
               "+v"(wd[0][0]), "+v"(wd[0][1]), "+v"(wd[0][2]), "+v"(wd[1][0]), "+v"(wd[1][1]), "+v"(wd[1][2]) :: "memory");
#undef OFFK_WAIT_WG
#undef OFFK_WAIT_WD

  // ---- epilogue (as pw_tdiff16_kernel): lane = (pixel li, channels 4 kq .. + 3 of a channel tile) ----
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int li_e = lane_e & 15, kq_e = lane_e >> 4;
  const int qe = qr0 + (li_e >> 2);
  const bool qn_e = qpc && qe >= qpc;
  const int bl = qpc ? b + (int)qn_e : b + (li_e >> rsh), pixl = qpc ? 4 * (qe - (qn_e ? qpc : 0)) + (li_e & 3) : q0 + (li_e & rmask);
  const size_t pair0 = (size_t)bl * (L - 1) + t0;
  const bool pix_ok = pixl < HW && bl < p.B;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x4 bg = *reinterpret_cast<const f32x4*>(S.bias + wave * 32 + 16 * ct + 4 * kq_e);
#pragma unroll
    for (int j = 0; j < PS_FT; ++j) {
      const f32x4 v = ag[j][ct] + bg;
      ag[j][ct] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    }
  }
#pragma unroll
  for (int j = 0; j + 1 < PS_FT; ++j)
    if (j + 1 < nf && pix_ok) {
      float* const trow = S.M + ((pair0 + j) * HW + pixl) * S.m_cs + S.m_coff + kDownCh + wave * 32 + 4 * kq_e;
      *reinterpret_cast<f32x4*>(trow) = ag[j + 1][0] - ag[j][0];
      *reinterpret_cast<f32x4*>(trow + 16) = ag[j + 1][1] - ag[j][1];
    }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int j = wave + 4 * half;
    if (j < nf && (last_group || j < PS_FT - 1) && pix_ok) {
      const int dr = ps_down_row(bl, t0 + j, L, p.P, p.slice_mode);
      if (dr >= 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 bd = *reinterpret_cast<const f32x4*>(S.bias_down + 16 * ct + 4 * kq_e);
          *reinterpret_cast<f32x4*>(S.D + ((size_t)dr * HW + pixl) * kDownCh + 16 * ct + 4 * kq_e) = ad[half][ct] + bd;
        }
      }
    }
  }
}

// Plane image of a site's 160 weight rows for pw_tdiff_split_kernel: one 16-byte item per (K-tile, slab of 32 rows, channel tile ct,
// plane q, lane): the bf16 plane q (0 = h, 1 = m, 2 = l) of W[slab * 32 + 16 ct + li][kt * 32 + 8 g + 0..7], li = lane & 15, g = lane >> 4.
__global__ void pw_pack_split16_kernel(const float* __restrict__ w, int C, uint4* __restrict__ out) {
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int nitems = (C / BK) * 5 * 2 * 64;
  if (item >= nitems) return;
  const int lane = item & 63, ct = (item >> 6) & 1, slab = (item >> 7) % 5, kt = (item >> 7) / 5;
  const int li = lane & 15, g = lane >> 4;
  const float* row = w + (size_t)(slab * 32 + 16 * ct + li) * C + kt * BK + 8 * g;
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned x = __float_as_uint(row[e]);
    h[e] = x & 0xffff0000u;
    const float r1 = row[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xffff0000u;
    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
  }
  uint4* dst = out + (size_t)((kt * 5 + slab) * 2 + ct) * 3 * 64 + lane;
  dst[0] = make_uint4((h[0] >> 16) | h[1], (h[2] >> 16) | h[3], (h[4] >> 16) | h[5], (h[6] >> 16) | h[7]);
  dst[64] = make_uint4((m[0] >> 16) | m[1], (m[2] >> 16) | m[3], (m[4] >> 16) | m[5], (m[6] >> 16) | m[7]);
  dst[128] = make_uint4((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u), (l[4] >> 16) | (l[5] & 0xffff0000u),
                        (l[6] >> 16) | (l[7] & 0xffff0000u));
}
hipError_t pw_pack_split16_launch(const float* w160, int C, void* out, hipStream_t st) {
  const int nitems = (C / BK) * 5 * 2 * 64;
  hipLaunchKernelGGL(pw_pack_split16_kernel, dim3((nitems + 255) / 256), dim3(256), 0, st, w160, C, reinterpret_cast<uint4*>(out));
  return hipGetLastError();
}

// p: block layout filled by pw_tdiff_launch (the 16-pixel form's)
hipError_t pw_tdiff_split_launch(const PtParams& p, hipStream_t st) {
  hipError_t er = lds_attr_once(reinterpret_cast<const void*>(pw_tdiff_split_kernel), PS_LDS);
  if (er != hipSuccess) return er;
  hipLaunchKernelGGL(pw_tdiff_split_kernel, dim3(p.total_blocks), dim3(256), PS_LDS, st, p);
  return hipGetLastError();
}

}  // namespace offk
