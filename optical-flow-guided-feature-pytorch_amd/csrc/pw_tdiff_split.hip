// K1T in split-fp32 arithmetic (rounds 5 - 6): the fused units kernel of pw_tdiff.hip (1x1 reduces + ReLU + temporal difference + the
// down rows, reference RGB_OFF.py:597-610) with every fp32 operand cut into THREE bf16 planes and multiplied on the bf16 matrix pipe.
//
// Arithmetic.  x = x_h + x_m + x_l exactly (8 + 8 + 8 significand bits: x_h = the upper 16 bits of x, the remainder is exact in
// fp32, cut again, and again), the same for w; every product of two planes is exact in fp32.  Of the nine plane products the six
// above 2^-24 of the leading one are formed on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, into TWO running accumulators per
// output tile (round 6): A1 takes the leading product w_h x_h, A2 the five small ones (w_l x_h, w_h x_l, w_m x_m, w_m x_h, w_h x_m --
// all below 2^-8 of the leading one, so A2's own roundings are 2^-8 of an fp32 chain's); out = A1 + A2 once, in the epilogue.
// Measured against fp64 on MI355X (tools/probe_split_mfma.hip mode 6, "6 products, 2 acc (hh | rest)", profiles/r05/probe_split_mfma.txt):
// max / rms error and the backward-error constant are 2 - 2.5 x SMALLER than the fp32 pipe's (v_mfma_f32_16x16x4_f32 chain) on every
// distribution the parity tests use -- level with round 5's form (all six products summed from zero per 32-k step and folded into one
// accumulator once per step), which cost ~88 v_add_f32 per wave and K-tile beside 108 MFMAs: with two waves per SIMD the vector issue
// port was as loaded as the matrix pipe (2 x (108 x 8 + 216 x 4) = 3456 cycles per step and SIMD either way) and the kernel ran at the
// sum of its floors.  One instruction sums the eight products of a lane group wide and adds the four lane groups to the accumulator
// one after the other (same probe, part 2).
//
// Structure (one block = (clip, 16-pixel chunk) x seven frames x 160 channels, as pw_tdiff16_kernel; two blocks per CU):
//   * feature map: wave w DMAs (buffer_load_dwordx4 ... lds) its four pieces of a K-tile (a piece = one frame's 16 k rows x 16 pixels
//     fp32, 1 KB) into its PRIVATE raw region -- two raw stages; in step k the wave cuts the pieces of tile k + 1 it loaded itself
//     into the three plane images of that tile (no barrier between DMA and cut: its own vmcnt covers them) and re-issues the DMA of
//     tile k + 3 into each piece's slot as soon as the cut has read it: every piece is two steps in flight;
//   * plane image per (frame, plane): [4 k groups][16 pixels] x 16 B (8 bf16 = k 8g .. 8g + 7): the MFMA's B operand is one
//     conflict-free ds_read_b128 per lane, the cut writes 8 B (a lane's four k of one pixel) per plane: ds_write_b64, conflict-free;
//     the raw piece is laid out by the DMA's per-lane global offsets so that the cut's reads are conflict-free too ([k & 3][k >> 2][pixel quad]);
//   * the cut: 22 vector instructions per piece and lane (and, sub, and, sub per value + six v_perm), in twelve stages of two placed by
//     hand behind individual MFMAs;
//   * weights: the library's plane image in operand order straight from L2 into registers (pw_pack_split16_kernel), gen set
//     double-buffered a whole K-tile ahead, down set reloaded behind its last use; hand-counted waits (each step issues exactly
//     6 + 4 + 3 vector-memory operations per wave: Wg(k + 1), DMA(k + 3), Wd(k + 1));
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
constexpr int PS_GS = 256;                            // one k group of a plane: 16 pixels x 16 B (8 bf16 = k 8g .. 8g + 7)
constexpr int PS_PLANE = 4 * PS_GS;                   // 32 k
constexpr int PS_FRAME = 3 * PS_PLANE;                // planes h, m, l
constexpr int PS_STAGE = (PS_FT + 1) * PS_FRAME;      // seven frames + frame slot 7: the two pieces of waves 2, 3 that have no frame (zeros)
constexpr int PS_LDS = 2 * PS_RAW_STAGE + 2 * PS_STAGE;      // 81920 B = 64 granules of 1280 B: two blocks per CU (163840 B)

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
  extern __shared__ __attribute__((aligned(16))) char lds[];     // raw[2 stages][4 waves][4 pieces][1 KB] | planes[2 stages][8 frame slots][3 planes][4 k groups][256 B]
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

  // ---- DMA: piece q = wave + 4 i (i = 0..3) = frame q >> 1, k half q & 1 (= wave & 1); pieces 14, 15 (frame slot 7) read zeros.  Lane l of
  //      a piece fetches k row 4 a + r (r = l >> 4, a = (l >> 2) & 3), pixel quad l & 3, and the DMA puts lane l's 16 B at 16 l: row k of
  //      the piece sits at [k & 3][k >> 2][pixel] ----
  const int pq = lane & 3;
  const bool qnext = qpc && qr0 + pq >= qpc;
  const int cq = qpc ? (int)qnext : (4 * pq) >> rsh;
  const int k0px = qpc ? 4 * (qr0 + pq - (qnext ? qpc : 0)) : q0 + ((4 * pq) & rmask);
  const bool px_ok = k0px < HW && b + cq < p.B;
  const int vrow0 = px_ok ? ((16 * (wave & 1) + 4 * ((lane >> 2) & 3) + (lane >> 4)) * HW + k0px) * 4 : (int)0x80000000;   // bit 31: past every descriptor -> zeros
  const int vclip = cq * L;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
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
  auto dma_piece = [&](const int i, const int rs) {       // piece i of the wave, of the prepared K-tile, into raw stage rs
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 32)     /* timing experiment: no feature-map loads */
    return;
#endif
    const int fr = (wave + 4 * i) >> 1;
    const int voff = fr < nf ? dm_voff : (int)0x80000000;              // (pieces 14, 15: fr = 7 >= nf)
    const unsigned lds_addr = lds_base + rs * PS_RAW_STAGE + wave * PS_RAW_WAVE + i * PS_PIECE;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(dm_desc), "s"(dm_s0 + fr * dm_fstride) : "memory", "m0");
  };

  // ---- the cut: lane (pixel li, a = lg) of piece i holds k = 16 (wave & 1) + 4 a + r, r = 0..3, of frame (wave + 4 i) >> 1: four values ->
  //      8 B (k 4 (a & 1) .. + 3 of k group 2 (wave & 1) + (a >> 1)) of each plane.  Twelve stages of two vector instructions, operands in
  //      registers between stages: a stage rides behind one MFMA (a v_mfma_f32_16x16x32_bf16 holds the SIMD's issue port for 8 of its 16
  //      cycles -- two 4-cycle instructions fit; left to the compiler a unit came out as twelve MFMAs back to back with the vector work
  //      behind them at its full cost, while the partner wave of the SIMD -- the CU's other block, in step with this one -- was at the
  //      same place) ----
  const char* const raw_rd = lds + wave * PS_RAW_WAVE + lg * 64 + li * 4;          // + stage, + piece, + r * 256
  char* const pl_wr = planes + (2 * (wave & 1) + (lg >> 1)) * PS_GS + li * 16 + (lg & 1) * 8;   // + stage, + frame, + plane
  unsigned cv[2][4];                          // raw values of the piece being cut / of the next piece (read a piece ahead of their use)
  unsigned ch[4], cm[4];
  float cr[4], cl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { cv[0][r] = cv[1][r] = 0u; ch[r] = cm[r] = 0u; cr[r] = cl[r] = 0.f; }
  auto cut_stage = [&](const int st, const int i, const int rs, const int ps) {
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 4)      /* timing experiment: no cut */
    if (i < 8) return;
#endif
    const int fr = (wave + 4 * i) >> 1;
    unsigned (&v)[4] = cv[i & 1];
    if (st == 0) {
      const char* src = raw_rd + rs * PS_RAW_STAGE + i * PS_PIECE;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const unsigned*>(src + r * 256);
    }
    if (st == 1) { ch[0] = v[0] & 0xffff0000u; ch[1] = v[1] & 0xffff0000u; }
    if (st == 2) { ch[2] = v[2] & 0xffff0000u; ch[3] = v[3] & 0xffff0000u; }
    if (st == 3) { cr[0] = __uint_as_float(v[0]) - __uint_as_float(ch[0]); cr[1] = __uint_as_float(v[1]) - __uint_as_float(ch[1]); }
    if (st == 4) { cr[2] = __uint_as_float(v[2]) - __uint_as_float(ch[2]); cr[3] = __uint_as_float(v[3]) - __uint_as_float(ch[3]); }
    char* dst = pl_wr + ps * PS_STAGE + fr * PS_FRAME;
    if (st == 5)
      *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_amdgcn_perm(ch[1], ch[0], 0x07060302), __builtin_amdgcn_perm(ch[3], ch[2], 0x07060302)};
    if (st == 6) { cm[0] = __float_as_uint(cr[0]) & 0xffff0000u; cm[1] = __float_as_uint(cr[1]) & 0xffff0000u; }
    if (st == 7) { cm[2] = __float_as_uint(cr[2]) & 0xffff0000u; cm[3] = __float_as_uint(cr[3]) & 0xffff0000u; }
    if (st == 8) { cl[0] = cr[0] - __uint_as_float(cm[0]); cl[1] = cr[1] - __uint_as_float(cm[1]); }      // <= 8 significant bits: the low halves are zero
    if (st == 9) { cl[2] = cr[2] - __uint_as_float(cm[2]); cl[3] = cr[3] - __uint_as_float(cm[3]); }
    if (st == 10)
      *reinterpret_cast<u32x2*>(dst + PS_PLANE) = u32x2{__builtin_amdgcn_perm(cm[1], cm[0], 0x07060302), __builtin_amdgcn_perm(cm[3], cm[2], 0x07060302)};
    if (st == 11)
      *reinterpret_cast<u32x2*>(dst + 2 * PS_PLANE) = u32x2{__builtin_amdgcn_perm(__float_as_uint(cl[1]), __float_as_uint(cl[0]), 0x07060302),
                                                            __builtin_amdgcn_perm(__float_as_uint(cl[3]), __float_as_uint(cl[2]), 0x07060302)};
  };

  // ---- weights: image [kt][slab (4 gen + 1 down)][ct][plane][lane] x 16 B (pw_pack_split16_kernel) ----
  u32x4 wg[2][2][3], wd[3];                // gen [set][ct][plane] (set = K-tile parity), down [plane] (the wave's ONE down channel tile)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    wd[q] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 2; ++c) { wg[0][c][q] = u32x4{0u, 0u, 0u, 0u}; wg[1][c][q] = u32x4{0u, 0u, 0u, 0u}; }
  }
  i32x4 wdesc;
  {
    const unsigned long long wa = reinterpret_cast<unsigned long long>(S.wt);
    wdesc = i32x4{(int)(unsigned)wa, (int)(unsigned)(wa >> 32) & 0xffff, kUnitCh * C * 6, 0x00020000};
  }
  const int wlane = lane * 16;
  // down tiles: wave w owns down channel tile w & 1 of frames (w >> 1) + 2 i, i = 0..3 (waves 2, 3: i = 3 is frame slot 7: zeros, never
  // stored) -- every wave loading BOTH down tiles made the weight stream 48 KB per block and K-tile, 24 KB of it the down rows four times
  const int ctd = wave & 1, fd0 = wave >> 1;
  auto load_wg1 = [&](const int set, const int n, int kt) {       // n = ct * 3 + plane
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 16)     /* timing experiment: no weight loads inside the loop */
    if (kt > 0) return;
#endif
    const int so = ((kt * 5 + wave) * 6 + n) * 1024;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wg[set][n / 3][n % 3]) : "v"(wlane), "s"(wdesc), "s"(so));
  };
  auto load_wd = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int so = (((kt * 5 + 4) * 2 + ctd) * 3 + q) * 1024;
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 16)
      if (kt > 0) continue;
#endif
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wd[q]) : "v"(wlane), "s"(wdesc), "s"(so));
    }
  };
  // "all but my N newest vector-memory operations have completed", tied to the registers the operations before that write ("memory": the
  // raw pieces the DMAs before that wrote)
#define OFFK_WAIT_STEP(N, set) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wg[set][0][0]), "+v"(wg[set][0][1]), "+v"(wg[set][0][2]), \
                                            "+v"(wg[set][1][0]), "+v"(wg[set][1][1]), "+v"(wg[set][1][2]) :: "memory")
#define OFFK_WAIT_WD(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]) :: "memory")

  f32x4 a1[PS_FT][2], a2[PS_FT][2], d1[4], d2[4];       // [frame][channel tile]: A1 = sum w_h x_h, A2 = the five small products
#pragma unroll
  for (int j = 0; j < PS_FT; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) { a1[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { d1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; d2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // B operand of a frame out of plane stage st: three ds_read_b128
  const char* const xrd = planes + lg * PS_GS + li * 16;
  const int xoffD = fd0 * PS_FRAME;                     // down frames: fd0 + 2 i
  auto rdx = [&](u32x4 (&x)[3], const int st, int foff) {
#pragma unroll
    for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const u32x4*>(xrd + st * PS_STAGE + foff + q * PS_PLANE);
  };
  auto mf = [&](f32x4& c, const u32x4& a, const u32x4& bb) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0);
  };

  const int nkt = C / BK;
#define OFFK_SB __builtin_amdgcn_sched_barrier(0)
  // One step: tile kt out of plane stage ST with the gen weights of set ST; the cut of tile kt + 1 out of raw stage ST ^ 1 into plane
  // stage ST ^ 1; that raw stage's pieces re-loaded with tile kt + 3 as the cut has read them; the gen weights of tile kt + 1 into set
  // ST ^ 1.  What rides behind MFMA m = 0 .. 107 of the step (`side`): piece i = m / 24 is cut behind MFMAs 24 i + {0, 4, 6, ..., 22, 23}
  // (its raw reads behind the first, two idle slots for their latency); the step's vector-memory instructions go out one at a time
  // (all at the top of the step they cost ~1200 cycles of issue: eight waves' 1-KB requests against the CU's 64 B / clk), in this order:
  // Wg(kt + 1) [6] behind MFMAs 2, 5, 14, 17, 26, 29; DMA(kt + 3) [4] behind 40, 54, 68, 92; Wd(kt + 1) [3] at the end -- 13 operations per
  // wave and step, the waits below count on it.
  auto step = [&](int kt, const int ST) {
    const int kn = min(kt + 1, nkt - 1);
    dma_prep(min(kt + 3, nkt - 1));
    OFFK_SB;
    auto side = [&](const int m) {
      const int i = m / 24, t = m % 24;
      if (t == 0) cut_stage(0, i, ST ^ 1, ST ^ 1);
      else if (t == 23) cut_stage(11, i, ST ^ 1, ST ^ 1);
      else if (t >= 4 && (t & 1) == 0) cut_stage(t / 2 - 1, i, ST ^ 1, ST ^ 1);
      if (m == 2) load_wg1(ST ^ 1, 0, kn);
      if (m == 5) load_wg1(ST ^ 1, 1, kn);
      if (m == 14) load_wg1(ST ^ 1, 2, kn);
      if (m == 17) load_wg1(ST ^ 1, 3, kn);
      if (m == 26) load_wg1(ST ^ 1, 4, kn);
      if (m == 29) load_wg1(ST ^ 1, 5, kn);
      if (m == 40) dma_piece(0, ST ^ 1);
      if (m == 54) dma_piece(1, ST ^ 1);
      if (m == 68) dma_piece(2, ST ^ 1);
      if (m == 92) dma_piece(3, ST ^ 1);
      OFFK_SB;
    };
    u32x4 x[2][3];
    rdx(x[0], ST, 0);
    // all but DMA(kt + 2) [4] and Wd(kt) [3]: the gen weights of tile kt and the raw pieces of tile kt + 1 (a step older)
    if (ST == 0) OFFK_WAIT_STEP(7, 0); else OFFK_WAIT_STEP(7, 1);
    // gen unit j: twelve MFMAs, the two channel tiles' chains alternating (planes 0 = h, 1 = m, 2 = l; the small products first)
#pragma unroll
    for (int j = 0; j < PS_FT; ++j) {
      if (j + 1 < PS_FT) rdx(x[(j + 1) & 1], ST, (j + 1) * PS_FRAME);
      else rdx(x[(j + 1) & 1], ST, xoffD);
      OFFK_SB;
      const u32x4 (&w0)[3] = wg[ST][0];
      const u32x4 (&w1)[3] = wg[ST][1];
      const u32x4 (&xx)[3] = x[j & 1];
      const int m0 = 12 * j;
      mf(a2[j][0], w0[2], xx[0]); side(m0 + 0);
      mf(a2[j][1], w1[2], xx[0]); side(m0 + 1);
      mf(a2[j][0], w0[0], xx[2]); side(m0 + 2);
      mf(a2[j][1], w1[0], xx[2]); side(m0 + 3);
      mf(a2[j][0], w0[1], xx[1]); side(m0 + 4);
      mf(a2[j][1], w1[1], xx[1]); side(m0 + 5);
      mf(a2[j][0], w0[1], xx[0]); side(m0 + 6);
      mf(a2[j][1], w1[1], xx[0]); side(m0 + 7);
      mf(a2[j][0], w0[0], xx[1]); side(m0 + 8);
      mf(a2[j][1], w1[0], xx[1]); side(m0 + 9);
      mf(a1[j][0], w0[0], xx[0]); side(m0 + 10);
      mf(a1[j][1], w1[0], xx[0]); side(m0 + 11);
    }
    // the wave's down tiles: frames fd0 + 2 i (x[1] holds the first): six MFMAs each
    OFFK_WAIT_WD(9);                          // Wd(kt): all but Wg(kt + 1) [6] and DMA(kt + 3) pieces 0 - 2 [3]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i + 1 < 4) rdx(x[i & 1], ST, xoffD + 2 * (i + 1) * PS_FRAME);
      OFFK_SB;
      const u32x4 (&xx)[3] = x[(i + 1) & 1];
      const int m0 = 12 * PS_FT + 6 * i;
      mf(d2[i], wd[2], xx[0]); side(m0 + 0);
      mf(d2[i], wd[0], xx[2]); side(m0 + 1);
      mf(d1[i], wd[0], xx[0]); side(m0 + 2);
      mf(d2[i], wd[1], xx[1]); side(m0 + 3);
      mf(d2[i], wd[1], xx[0]); side(m0 + 4);
      mf(d2[i], wd[0], xx[1]); side(m0 + 5);
    }
    load_wd(kn);
    __syncthreads();                         // (the compiler's lgkmcnt(0) in front of it covers the plane writes)
  };

  // ---- prologue: DMA(0) [4], DMA(1) [4], Wg(0) [6]; cut tile 0; DMA(2) [4], Wd(0) [3] -- the order the steps' wait counts assume ----
  dma_prep(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(i, 0);
  dma_prep(min(1, nkt - 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(i, 1);
#pragma unroll
  for (int n = 0; n < 6; ++n) load_wg1(0, n, 0);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // DMA(0)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int st = 0; st < 12; ++st) cut_stage(st, i, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the cut's raw reads have returned before the DMAs below overwrite the pieces
  dma_prep(min(2, nkt - 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(i, 0);
  load_wd(0);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    step(kt, 0);
    step(kt + 1, 1);
  }
  if (kt < nkt) step(kt, 0);
#undef OFFK_SB
  // every load has returned before its registers die (and every DMA before the block's LDS is handed on)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[0][0][0]), "+v"(wg[0][0][1]), "+v"(wg[0][0][2]), "+v"(wg[0][1][0]), "+v"(wg[0][1][1]), "+v"(wg[0][1][2]),
               "+v"(wg[1][0][0]), "+v"(wg[1][0][1]), "+v"(wg[1][0][2]), "+v"(wg[1][1][0]), "+v"(wg[1][1][1]), "+v"(wg[1][1][2]) :: "memory");
  asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]));
#undef OFFK_WAIT_STEP
#undef OFFK_WAIT_WD

  // ---- epilogue (as pw_tdiff16_kernel): lane = (pixel li, channels 4 kq .. + 3 of a channel tile) ----
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int li_e = lane_e & 15, kq_e = lane_e >> 4;
  const int qe = qr0 + (li_e >> 2);
  const bool qn_e = qpc && qe >= qpc;
  const int bl = qpc ? b + (int)qn_e : b + (li_e >> rsh), pixl = qpc ? 4 * (qe - (qn_e ? qpc : 0)) + (li_e & 3) : q0 + (li_e & rmask);
  const size_t pair0 = (size_t)bl * (L - 1) + t0;
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 64)     /* timing experiment: (almost) no stores */
  const bool pix_ok = pixl < HW && bl < p.B && lane_e == 0 && blockIdx.x == 0;
#else
  const bool pix_ok = pixl < HW && bl < p.B;
#endif
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const f32x4 bg = *reinterpret_cast<const f32x4*>(S.bias + wave * 32 + 16 * ct + 4 * kq_e);
#pragma unroll
    for (int j = 0; j < PS_FT; ++j) {
      const f32x4 v = (a1[j][ct] + a2[j][ct]) + bg;
      a1[j][ct] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    }
  }
#pragma unroll
  for (int j = 0; j + 1 < PS_FT; ++j)
    if (j + 1 < nf && pix_ok) {
      float* const trow = S.M + ((pair0 + j) * HW + pixl) * S.m_cs + S.m_coff + kDownCh + wave * 32 + 4 * kq_e;
      *reinterpret_cast<f32x4*>(trow) = a1[j + 1][0] - a1[j][0];
      *reinterpret_cast<f32x4*>(trow + 16) = a1[j + 1][1] - a1[j][1];
    }
  {
    const int ctd_e = wave & 1, fd0_e = wave >> 1;
    const f32x4 bd = *reinterpret_cast<const f32x4*>(S.bias_down + 16 * ctd_e + 4 * kq_e);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = fd0_e + 2 * i;                 // (waves 2, 3, i = 3: j = 7 >= nf -- the zero tile of frame slot 7 is dropped here)
      // the frame shared with the next temporal group belongs to that group
      if (j < nf && (last_group || j < PS_FT - 1) && pix_ok) {
        const int dr = ps_down_row(bl, t0 + j, L, p.P, p.slice_mode);
        if (dr >= 0) *reinterpret_cast<f32x4*>(S.D + ((size_t)dr * HW + pixl) * kDownCh + 16 * ctd_e + 4 * kq_e) = (d1[i] + d2[i]) + bd;
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
