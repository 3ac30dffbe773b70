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
// one k group of a plane: 16 pixel slots x 16 B (8 bf16 = k 8g .. 8g + 7), pixel px of k group g in slot px ^ 2 (g & 1).  ds_read_b128
// is served in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) against 64 banks: with 256-B k groups every group
// covers a whole bank row; the slot swizzle keeps the cut's ds_write_b32 (32 banks, 32-lane groups) at two lanes per bank, which is free
// (272-byte k groups without the swizzle, the first layout: every ds_read_b128 group met one two-way conflict -- SQ_LDS_BANK_CONFLICT
// was half of SQ_LDS_IDX_ACTIVE, profiles/r05/pmc_per_kernel_f32split.csv; the LDS is not this kernel's limiter: -0.7 %)
constexpr int PS_GS = 256;
constexpr int PS_PLANE = 4 * PS_GS;                   // 32 k
constexpr int PS_FRAME = 3 * PS_PLANE;                // planes h, m, l
constexpr int PS_STAGE = (PS_FT + 1) * PS_FRAME;      // seven frames + the slot wave 3's idle loader half cuts its zeros into
constexpr int PS_LDS = 2 * PS_STAGE;                  // 49152 B = 39 granules of 1280 B

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ps_down_row(int b, int t, int L, int P, int slice_mode) {
  if (slice_mode == 0) { const int f = b * L + t; return f < P ? f : -1; }
  return t < L - 1 ? b * (L - 1) + t : -1;
}
}  // namespace

__global__ __launch_bounds__(256, 2) void pw_tdiff_split_kernel(PtParams p) {
  extern __shared__ __attribute__((aligned(16))) char planes[];     // [2 stages][8 frame slots][3 planes][4 k groups][256 B]

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

  // ---- feature-map loader: wave w loads frames w and w + 4 (slot fs = 0, 1; wave 3's second slot reads zeros); lane (pixel quad pq =
  //      lane & 3, k pair kp = lane >> 2) holds X[k = 2 kp + e][4 pq .. + 3] of the K-tile, e = 0, 1: two 16-byte loads per frame ----
  const int pq = lane & 3, kp = lane >> 2;
  const bool qnext = qpc && qr0 + pq >= qpc;
  const int cq = qpc ? (int)qnext : (4 * pq) >> rsh;
  const int k0px = qpc ? 4 * (qr0 + pq - (qnext ? qpc : 0)) : q0 + ((4 * pq) & rmask);
  const bool px_ok = k0px < HW && b + cq < p.B;
  const int vrow0 = px_ok ? (2 * kp * HW + k0px) * 4 : (int)0x80000000;        // bit 31: past every descriptor -> zeros
  const int vclip = cq * L;
  u32x4 xr[2][2][2];                          // [register set = K-tile parity][frame slot][e]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int f = 0; f < 2; ++f) { xr[a][f][0] = u32x4{0u, 0u, 0u, 0u}; xr[a][f][1] = u32x4{0u, 0u, 0u, 0u}; }
  i32x4 xd_desc = {0, 0, 0, 0};
  int xd_fstride = 0, xd_s0 = 0, xd_voff = 0;
  auto x_prep = [&](int kt) {
    const float* xb = S.xp[0]; int cpart = S.cp[0], kl = kt * BK;
    if (S.nparts > 1 && kl >= S.cp[0]) {
      kl -= S.cp[0]; xb = S.xp[1]; cpart = S.cp[1];
      if (S.nparts > 2 && kl >= S.cp[1]) {
        kl -= S.cp[1]; xb = S.xp[2]; cpart = S.cp[2];
        if (S.nparts > 3 && kl >= S.cp[2]) { kl -= S.cp[2]; xb = S.xp[3]; cpart = S.cp[3]; }
      }
    }
    const unsigned long long xa = reinterpret_cast<unsigned long long>(xb);
    xd_desc = i32x4{(int)(unsigned)xa, (int)(unsigned)(xa >> 32) & 0xffff, p.B * L * cpart * HW * 4, 0x00020000};
    xd_fstride = cpart * HW * 4;
    xd_s0 = (((b * L + t0) * cpart + kl) * HW) * 4;
    xd_voff = vrow0 + (int)__umul24((unsigned)vclip, (unsigned)xd_fstride);
  };
  auto load_x1 = [&](const int set, const int fs, const int e) {      // row e of frame slot fs of the prepared K-tile into register set `set`
    const int fr = wave + 4 * fs;
    const int voff = fr < nf ? xd_voff : (int)0x80000000;
    const int so = xd_s0 + fr * xd_fstride + e * HW * 4;
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 32)     /* timing experiment: no feature-map loads */
    return;
#endif
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(xr[set][fs][e]) : "v"(voff), "s"(xd_desc), "s"(so) : "memory");
  };
  auto load_x = [&](const int set, const int fs) { load_x1(set, fs, 0); load_x1(set, fs, 1); };

  // ---- the cut.  Atom a = (frame slot a >> 2, pixel i = a & 3 of the lane's quad): the lane's two k of that pixel -> one dword (k pair)
  //      of each plane.  In stages of two vector instructions, operands in registers between stages: a stage rides behind one MFMA
  //      (a v_mfma_f32_16x16x32_bf16 holds the SIMD's issue port for 8 of its 16 cycles -- two 4-cycle instructions fit; left to the
  //      compiler a unit came out as twelve MFMAs back to back with the vector work behind them at its full cost, while the partner wave
  //      of the SIMD -- the CU's other block, in step with this one -- was at the same place) ----
  char* const pl_wr = planes + (kp >> 2) * PS_GS + 4 * pq * 16 + (kp & 3) * 4;       // + stage, + frame, + plane, + pixel i * 16
  const int wr_swz = (kp & 4) ? 32 : 0;      // pixel i of the quad goes to slot i ^ 2 in the odd k groups
  unsigned ch0 = 0, ch1 = 0, cm0 = 0, cm1 = 0;
  float cr0 = 0.f, cr1 = 0.f, cl0 = 0.f, cl1 = 0.f;
  auto atom_stage = [&](const int st, const int a, const int set, const int ps) {
#if defined(OFFK_PS_EXP) && (OFFK_PS_EXP & 4)      /* timing experiment: no cut */
    if (a < 8) return;
#endif
    const int fs = a >> 2, i = a & 3;
    const unsigned xa = xr[set][fs][0][i], xb = xr[set][fs][1][i];
    if (st == 0) { ch0 = xa & 0xffff0000u; ch1 = xb & 0xffff0000u; }
    if (st == 1) { cr0 = __uint_as_float(xa) - __uint_as_float(ch0); cr1 = __uint_as_float(xb) - __uint_as_float(ch1); }
    if (st == 2) { cm0 = __float_as_uint(cr0) & 0xffff0000u; cm1 = __float_as_uint(cr1) & 0xffff0000u; }
    if (st == 3) { cl0 = cr0 - __uint_as_float(cm0); cl1 = cr1 - __uint_as_float(cm1); }      // <= 8 significant bits: the low halves are zero
    char* dst = pl_wr + ps * PS_STAGE + (wave + 4 * fs) * PS_FRAME + i * 16 + (i < 2 ? wr_swz : -wr_swz);      // (wave 3, slot 1: frame slot 7 -- never read)
    if (st == 4) {
      *reinterpret_cast<unsigned*>(dst) = __builtin_amdgcn_perm(ch1, ch0, 0x07060302);
      *reinterpret_cast<unsigned*>(dst + PS_PLANE) = __builtin_amdgcn_perm(cm1, cm0, 0x07060302);
    }
    if (st == 5) *reinterpret_cast<unsigned*>(dst + 2 * PS_PLANE) = __builtin_amdgcn_perm(__float_as_uint(cl1), __float_as_uint(cl0), 0x07060302);
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
  // down tiles: wave w owns down channel tile w & 1 of frames (w >> 1) + 2 i, i = 0..3 (waves 2, 3: i = 3 is frame 6 again, never
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
  // "all but my N newest vector-memory operations have completed", tied to the registers the operations before that write
#define OFFK_WAIT_STEP(N, set) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wg[set][0][0]), "+v"(wg[set][0][1]), "+v"(wg[set][0][2]), \
                                            "+v"(wg[set][1][0]), "+v"(wg[set][1][1]), "+v"(wg[set][1][2]), "+v"(xr[set ^ 1][0][0]), \
                                            "+v"(xr[set ^ 1][0][1]), "+v"(xr[set ^ 1][1][0]), "+v"(xr[set ^ 1][1][1]) :: "memory")
#define OFFK_WAIT_WD(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]) :: "memory")

  f32x4 ag[PS_FT][2], ad[4];
#pragma unroll
  for (int j = 0; j < PS_FT; ++j) { ag[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 4; ++i) ad[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B operand of a frame out of plane stage st: three ds_read_b128
  const char* const xrd = planes + lg * PS_GS + (li ^ (2 * (lg & 1))) * 16;
  const int xoffD = fd0 * PS_FRAME, xoffD3 = min(fd0 + 6, PS_FT - 1) * PS_FRAME;     // down frames: xoffD + 2 i frames, the fourth clamped
  auto rdx = [&](u32x4 (&x)[3], const int st, int foff) {
#pragma unroll
    for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const u32x4*>(xrd + st * PS_STAGE + foff + q * PS_PLANE);
  };
  auto mf = [&](f32x4 c, const u32x4& a, const u32x4& bb) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0);
  };
  // half of a scratch tile into its accumulator (two v_add_f32)
  auto fold2 = [&](f32x4& acc, const f32x4& tv, const int hi) {
    if (hi) { acc.z += tv.z; acc.w += tv.w; } else { acc.x += tv.x; acc.y += tv.y; }
    asm volatile("" : "+v"(acc));           // the update stays where it is written (hipcc sinks it across any block boundary otherwise)
  };

  const int nkt = C / BK;
#ifdef OFFK_PT_TIMING
  unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tm_c = 0;
  const unsigned long long tm_begin = __builtin_readcyclecounter();
#define OFFK_TICK(i) { const unsigned long long c_ = __builtin_readcyclecounter(); tm[i] += c_ - tm_c; tm_c = c_; }
#else
#define OFFK_TICK(i)
#endif
#define OFFK_SB __builtin_amdgcn_sched_barrier(0)
  // One step: tile kt out of plane stage ST with the gen weights of set ST; the cut of tile kt + 1 out of register set ST ^ 1 into plane
  // stage ST ^ 1 (eight atoms: one beside each gen unit and the first down unit); that register set re-loaded with tile kt + 3; the gen
  // weights of tile kt + 1 into set ST ^ 1.  The step's vector-memory instructions go out one at a time between the MFMAs (all at the
  // top of the step they cost ~1200 cycles of issue: eight waves' 1-KB requests against the CU's 64 B / clk), in this order:
  // Wg(kt + 1) [6], X(kt + 3) [2 + 2], Wd(kt + 1) [3] -- 13 operations per wave and step, the waits below count on it.
  auto step = [&](int kt, const int ST) {
    const int kn = min(kt + 1, nkt - 1);
#ifdef OFFK_PT_TIMING
    tm_c = __builtin_readcyclecounter();
#endif
    x_prep(min(kt + 3, nkt - 1));
    OFFK_SB;
    OFFK_TICK(0)
    u32x4 x[2][3];
    f32x4 t[2][2];
    rdx(x[0], ST, 0);
    // all but X(kt + 2) [4] and Wd(kt) [3]: the gen weights of tile kt and the feature-map registers of tile kt + 1 (a step older)
    if (ST == 0) OFFK_WAIT_STEP(7, 0); else OFFK_WAIT_STEP(7, 1);
    OFFK_TICK(1)
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // gen unit j: twelve MFMAs (the two channel tiles' chains alternating; planes 0 = h, 1 = m, 2 = l, smallest products first); behind
    // MFMA n: n = 0..3 the scratch tiles of unit j - 1 into their accumulators, n = 4..9 the stages of cut atom j
#pragma unroll
    for (int j = 0; j < PS_FT; ++j) {
      if (j + 1 < PS_FT) rdx(x[(j + 1) & 1], ST, (j + 1) * PS_FRAME);
      else rdx(x[(j + 1) & 1], ST, xoffD);
      OFFK_SB;
      const u32x4 (&w0)[3] = wg[ST][0];
      const u32x4 (&w1)[3] = wg[ST][1];
      const u32x4 (&xx)[3] = x[j & 1];
      f32x4 &t0 = t[j & 1][0], &t1 = t[j & 1][1];
      const f32x4 &p0 = t[(j - 1) & 1][0], &p1 = t[(j - 1) & 1][1];
      const int jp = j > 0 ? j - 1 : 0;
      t0 = mf(z, w0[2], xx[0]);  if (j > 0) fold2(ag[jp][0], p0, 0); OFFK_SB;
      t1 = mf(z, w1[2], xx[0]);  if (j > 0) fold2(ag[jp][0], p0, 1); OFFK_SB;
      t0 = mf(t0, w0[0], xx[2]); if (j > 0) fold2(ag[jp][1], p1, 0); OFFK_SB;
      t1 = mf(t1, w1[0], xx[2]); if (j > 0) fold2(ag[jp][1], p1, 1); OFFK_SB;
      t0 = mf(t0, w0[1], xx[1]); atom_stage(0, j, ST ^ 1, ST ^ 1); OFFK_SB;
      t1 = mf(t1, w1[1], xx[1]); atom_stage(1, j, ST ^ 1, ST ^ 1); OFFK_SB;
      if (j < 3) { load_wg1(ST ^ 1, 2 * j, kn); OFFK_SB; }
      t0 = mf(t0, w0[1], xx[0]); atom_stage(2, j, ST ^ 1, ST ^ 1); OFFK_SB;
      t1 = mf(t1, w1[1], xx[0]); atom_stage(3, j, ST ^ 1, ST ^ 1); OFFK_SB;
      t0 = mf(t0, w0[0], xx[1]); atom_stage(4, j, ST ^ 1, ST ^ 1); OFFK_SB;
      t1 = mf(t1, w1[0], xx[1]); atom_stage(5, j, ST ^ 1, ST ^ 1); OFFK_SB;
      t0 = mf(t0, w0[0], xx[0]); OFFK_SB;
      t1 = mf(t1, w1[0], xx[0]); OFFK_SB;
      if (j < 3) { load_wg1(ST ^ 1, 2 * j + 1, kn); OFFK_SB; }
      if (j == 4) { load_x1(ST ^ 1, 0, 0); OFFK_SB; }      // atoms 0..3 have read slot 0 of the set; one load per unit: eight waves' KBs in
      if (j == 5) { load_x1(ST ^ 1, 0, 1); OFFK_SB; }      // one burst fill the CU's queue and the next instruction of every wave waits
    }
    OFFK_TICK(2)
    // the wave's down tiles: frames fd0 + 2 i (x[1] holds the first): one chain of six MFMAs each; beside them the scratch tiles of gen
    // unit 6, cut atom 7, and the down tiles' own scratch tiles
    OFFK_WAIT_WD(8);                          // Wd(kt): all but Wg(kt + 1) [6] and X(kt + 3) slot 0 [2]
    OFFK_TICK(3)
    f32x4 td[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i + 1 < 4) rdx(x[i & 1], ST, i + 1 < 3 ? xoffD + 2 * (i + 1) * PS_FRAME : xoffD3);
      OFFK_SB;
      if (i == 1) { load_x1(ST ^ 1, 1, 0); OFFK_SB; }      // atom 7 has read slot 1 (its stages 0, 1 ran beside the first down unit)
      if (i == 2) { load_x1(ST ^ 1, 1, 1); OFFK_SB; }
      const u32x4 (&xx)[3] = x[(i + 1) & 1];
      f32x4 &t0 = td[i & 1];
      const f32x4 &pd = td[(i - 1) & 1];
      const int ip = i > 0 ? i - 1 : 0;
      t0 = mf(z, wd[2], xx[0]);
      if (i == 0) fold2(ag[PS_FT - 1][0], t[0][0], 0); else fold2(ad[ip], pd, 0);
      OFFK_SB;
      t0 = mf(t0, wd[0], xx[2]);
      if (i == 0) fold2(ag[PS_FT - 1][0], t[0][0], 1); else fold2(ad[ip], pd, 1);
      OFFK_SB;
      t0 = mf(t0, wd[1], xx[1]);
      if (i == 0) fold2(ag[PS_FT - 1][1], t[0][1], 0); else if (i == 1) atom_stage(2, 7, ST ^ 1, ST ^ 1);
      OFFK_SB;
      t0 = mf(t0, wd[1], xx[0]);
      if (i == 0) fold2(ag[PS_FT - 1][1], t[0][1], 1); else if (i == 1) atom_stage(3, 7, ST ^ 1, ST ^ 1);
      OFFK_SB;
      t0 = mf(t0, wd[0], xx[1]);
      if (i == 0) atom_stage(0, 7, ST ^ 1, ST ^ 1); else if (i == 1) atom_stage(4, 7, ST ^ 1, ST ^ 1);
      OFFK_SB;
      t0 = mf(t0, wd[0], xx[0]);
      if (i == 0) atom_stage(1, 7, ST ^ 1, ST ^ 1); else if (i == 1) atom_stage(5, 7, ST ^ 1, ST ^ 1);
      OFFK_SB;
    }
    load_wd(kn);
    fold2(ad[3], td[1], 0);
    fold2(ad[3], td[1], 1);
    OFFK_TICK(4)
    __syncthreads();                         // (the compiler's lgkmcnt(0) in front of it covers the plane writes)
    OFFK_TICK(5)
  };

  // ---- prologue: X(0) [4], X(1) [4], Wg(0) [6]; cut tile 0; X(2) [4], Wd(0) [3] -- the order the steps' wait counts assume ----
  x_prep(0);
  load_x(0, 0); load_x(0, 1);
  x_prep(min(1, nkt - 1));
  load_x(1, 0); load_x(1, 1);
#pragma unroll
  for (int n = 0; n < 6; ++n) load_wg1(0, n, 0);
  asm volatile("s_waitcnt vmcnt(10)" : "+v"(xr[0][0][0]), "+v"(xr[0][0][1]), "+v"(xr[0][1][0]), "+v"(xr[0][1][1]) :: "memory");      // X(0)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int st = 0; st < 6; ++st) atom_stage(st, a, 0, 0);
  x_prep(min(2, nkt - 1));
  load_x(0, 0); load_x(0, 1);
  load_wd(0);
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
#undef OFFK_SB
  // every load has returned before its registers die
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[0][0][0]), "+v"(wg[0][0][1]), "+v"(wg[0][0][2]), "+v"(wg[0][1][0]), "+v"(wg[0][1][1]), "+v"(wg[0][1][2]),
               "+v"(wg[1][0][0]), "+v"(wg[1][0][1]), "+v"(wg[1][0][2]), "+v"(wg[1][1][0]), "+v"(wg[1][1][1]), "+v"(wg[1][1][2]) :: "memory");
  asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]), "+v"(xr[0][0][0]), "+v"(xr[0][0][1]), "+v"(xr[0][1][0]), "+v"(xr[0][1][1]),
               "+v"(xr[1][0][0]), "+v"(xr[1][0][1]), "+v"(xr[1][1][0]), "+v"(xr[1][1][1]));
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
  {
    const int ctd_e = wave & 1, fd0_e = wave >> 1;
    const f32x4 bd = *reinterpret_cast<const f32x4*>(S.bias_down + 16 * ctd_e + 4 * kq_e);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = fd0_e + 2 * i;                 // (waves 2, 3, i = 3: j = 7 >= nf -- the duplicate tile is dropped here)
      // the frame shared with the next temporal group belongs to that group
      if (j < nf && (last_group || j < PS_FT - 1) && pix_ok) {
        const int dr = ps_down_row(bl, t0 + j, L, p.P, p.slice_mode);
        if (dr >= 0) *reinterpret_cast<f32x4*>(S.D + ((size_t)dr * HW + pixl) * kDownCh + 16 * ctd_e + 4 * kq_e) = ad[i] + bd;
      }
    }
  }
#ifdef OFFK_PT_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.dbg && threadIdx.x == 0) {
    const unsigned long long tm_end = __builtin_readcyclecounter();
    atomicAdd(p.dbg + 8, tm_loop - tm_begin); atomicAdd(p.dbg + 9, tm_epi - tm_loop); atomicAdd(p.dbg + 10, tm_end - tm_epi);
    for (int i = 0; i < 6; ++i) atomicAdd(p.dbg + 11 + i, tm[i]);
    atomicAdd(p.dbg + 17, (unsigned long long)nkt); atomicAdd(p.dbg + 18, 1ull);
  }
#endif
#undef OFFK_TICK
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
