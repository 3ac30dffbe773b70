// K1T in split-fp32 arithmetic, producer / consumer form (round 5): the arithmetic and the tile of pw_tdiff_split.hip -- a (clip,
// 16-pixel chunk) x seven frames x 160 channels, three bf16 planes per fp32 operand, six plane products per 32-k step summed from zero
// and added to the fp32 accumulator once per step (reference RGB_OFF.py:597-610) -- as ONE persistent block of eight waves per CU:
//
//   waves 4..7, producers: stream the feature map from HBM into registers two K-tiles ahead, cut it into the three plane images of a
//       K-tile and write those into a ring of three LDS stages -- nothing else.  A producer that waits for HBM blocks nobody.
//   waves 0..3, consumers: one per SIMD, never touch the feature map in global memory: B operands out of the ring (ds_read_b128), weights
//       straight from L2 into registers (the library's plane image, a cached 1-KB load costs ~10 cycles beside MFMAs), MFMAs, the
//       scratch-tile adds, the epilogue.
//
// Why (profiles/r05/split_units_cycle_split.txt, probe_vmem_issue.txt): in pw_tdiff_split_kernel every wave loads, cuts and multiplies;
// two blocks per CU run in step, so a wave that stalls on a feature-map load (the kernel moves 3 GB per launch: ~0.5 ms of HBM time,
// as long as its MFMAs take) stalls beside its SIMD partner, and the kernel's time came out as the SUM of its MFMA time and its
// memory time (0.90 ms; 0.59 ms with the loads compiled out).  With the roles split a consumer's MFMA stream stays dense while the
// producers wait (probe: 188 -> 196 cycles per 12 MFMAs with four producer waves streaming beside them).
//
// One wave per SIMD multiplies, so a consumer interleaves THREE chains (a dependent v_mfma_f32_16x16x32_bf16 is ~44 cycles away, two
// chains leave the pipe idle a quarter of the time): a step = six units of three chains x six products = 108 MFMAs,
//   units 0..3: gen frame u, channel tiles 0 and 1, and the wave's down tile of frame fd0 + 2 u
//   unit 4: frame 4 tiles 0, 1 and frame 5 tile 0;   unit 5: frame 5 tile 1 and frame 6 tiles 0, 1.
// A chain's scratch tile is added to its accumulator right in front of the first MFMA of the chain that re-uses its registers (one unit
// later: the result is long there).
//
// Ring protocol (one s_barrier of all eight waves per slot): in slot s the producers cut K-tile s of the block's tile stream into
// stage s % 3 while the consumers multiply tile s - 2 out of stage (s - 2) % 3; tile s - 1 is complete and stays untouched through
// slot s + 1, so a consumer may read the first operands of its next step before the barrier.  The tile stream runs across items
// (item l = the block pw_tdiff_split_kernel would launch as block l; a persistent block takes l = blockIdx.x + n gridDim.x): no
// prologue per item, the producers are into the next item while the consumers store the last one.
#include <cstdio>
#include <cstdlib>

// NOT in the product build (round 5: measured 1.07 ms against pw_tdiff_split_kernel's 0.90 ms, profiles/r05/split_units_producer_consumer.txt).
// To build it: add this file to build.py's SOURCES (flags as pw_tdiff_split.hip) and compile pw_tdiff.hip / offk_api.hip with -DOFFK_WITH_PC;
// OFFK_SPLIT_PC=2 at offk_create selects it.
#include "../../optical-flow-guided-feature-pytorch_amd/csrc/offk_common.h"
#define OFFK_WITH_PC
#include "../../optical-flow-guided-feature-pytorch_amd/csrc/offk_internal.h"

namespace offk {

namespace {
constexpr int PC_FT = 7;
constexpr int PC_GS = 272;                            // one k group of a plane: 16 pixels x 16 B + 16 B of padding (conflict-free cut writes)
constexpr int PC_PLANE = 4 * PC_GS;
constexpr int PC_FRAME = 3 * PC_PLANE;
constexpr int PC_STAGE = (PC_FT + 1) * PC_FRAME;      // + the slot producer 3's idle loader half cuts its zeros into
constexpr int PC_NST = 3;
constexpr int PC_LDS = PC_NST * PC_STAGE;             // 78336 B

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pc_down_row(int b, int t, int L, int P, int slice_mode) {
  if (slice_mode == 0) { const int f = b * L + t; return f < P ? f : -1; }
  return t < L - 1 ? b * (L - 1) + t : -1;
}
__device__ __forceinline__ int sc(int v) { return __builtin_amdgcn_readfirstlane(v); }

// an item = what block l of pw_tdiff_split_kernel's launch works on (wave-uniform scalars)
struct PcItem {
  int si, C, HW, nkt;
  int b, q0, qr0, qpc, rsh, t0, nf, last_group;
};
__device__ __forceinline__ void pc_locate(const PtParams& p, int l, PcItem& it) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && l >= p.s[i].blk_begin) si = i;
  si = sc(si);
  const PtSite& S = p.s[si];
  const int nblk_site = (si + 1 < p.nsites ? p.s[si + 1].blk_begin : p.total_blocks) - S.blk_begin;
  int local = xcd_contiguous(l - S.blk_begin, nblk_site);
  const int tg = local % p.tgroups; local /= p.tgroups;
  const int nfull = p.B * S.chunks;
  const bool leftover = local >= nfull;
  const int qpc = S.qpc;
  const int b = qpc ? (4 * local) / qpc : leftover ? (local - nfull) << (4 - S.rsh) : local / max(S.chunks, 1);
  it.si = si; it.C = S.C; it.HW = S.HW; it.nkt = S.C / BK;
  it.rsh = sc(leftover ? S.rsh : 4);
  it.qpc = qpc;
  it.b = sc(b);
  it.qr0 = sc(qpc ? 4 * local - b * qpc : 0);
  it.q0 = sc((leftover ? S.chunks : local - b * S.chunks) * 16);
  it.t0 = sc(tg * (PC_FT - 1));
  it.nf = sc(min(PC_FT, p.L - tg * (PC_FT - 1)));
  it.last_group = sc(tg == p.tgroups - 1);
}
}  // namespace

__global__ __launch_bounds__(512, 1) void pw_tdiff_pc_kernel(PtParams p) {
  extern __shared__ __attribute__((aligned(16))) char planes[];     // [3 stages][8 frame slots][3 planes][4 k groups][272 B]
  const int lane = threadIdx.x & 63;
  const int wave = sc((int)(threadIdx.x >> 6));
  const int G = (int)gridDim.x, L = p.L;
  // tiles of this block's stream
  int T = 0;
  for (int l = (int)blockIdx.x; l < p.total_blocks; l += G) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < kNumSites; ++i)
      if (i < p.nsites && l >= p.s[i].blk_begin) si = i;
    T += p.s[sc(si)].C / BK;
  }
  T = sc(T);

  if (wave >= 4) {
    // =============================================== producers ===============================================
    // producer pw loads frames pw and pw + 4 of every K-tile (slot fs = 0, 1; producer 3's second slot reads zeros); lane (pixel quad
    // pq = lane & 3, k pair kp = lane >> 2) holds X[k = 2 kp + e][4 pq .. + 3], e = 0, 1
    const int pw = wave - 4;
    const int pq = lane & 3, kp = lane >> 2;
    u32x4 xr[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int f = 0; f < 2; ++f) { xr[a][f][0] = u32x4{0u, 0u, 0u, 0u}; xr[a][f][1] = u32x4{0u, 0u, 0u, 0u}; }
    // load cursor
    int ld_l = (int)blockIdx.x, ld_kt = 0;
    PcItem it;
    int vrow0 = (int)0x80000000, vclip = 0;
    auto geometry = [&]() {        // the lane's pixel quad of the item under the cursor
      const int rmask = (1 << it.rsh) - 1;
      const bool qnext = it.qpc && it.qr0 + pq >= it.qpc;
      const int cq = it.qpc ? (int)qnext : (4 * pq) >> it.rsh;
      const int k0px = it.qpc ? 4 * (it.qr0 + pq - (qnext ? it.qpc : 0)) : it.q0 + ((4 * pq) & rmask);
      const bool px_ok = k0px < it.HW && it.b + cq < p.B;
      vrow0 = px_ok ? (2 * kp * it.HW + k0px) * 4 : (int)0x80000000;
      vclip = cq * L;
    };
    pc_locate(p, ld_l, it);
    geometry();
    auto load_tile = [&](const int set, const bool real) {      // the K-tile under the cursor (or four loads of zeros: the counts stay)
      const PtSite& S = p.s[it.si];
      const float* xb = S.xp[0]; int cpart = S.cp[0], kl = ld_kt * BK;
      if (S.nparts > 1 && kl >= S.cp[0]) {
        kl -= S.cp[0]; xb = S.xp[1]; cpart = S.cp[1];
        if (S.nparts > 2 && kl >= S.cp[1]) {
          kl -= S.cp[1]; xb = S.xp[2]; cpart = S.cp[2];
          if (S.nparts > 3 && kl >= S.cp[2]) { kl -= S.cp[2]; xb = S.xp[3]; cpart = S.cp[3]; }
        }
      }
      const unsigned long long xa = reinterpret_cast<unsigned long long>(xb);
      const i32x4 desc = {sc((int)(unsigned)xa), sc((int)(unsigned)(xa >> 32) & 0xffff), sc(p.B * L * cpart * it.HW * 4), 0x00020000};
      const int fstride = sc(cpart * it.HW * 4);
      const int s0 = sc((((it.b * L + it.t0) * cpart + kl) * it.HW) * 4);
      const int voff_ok = vrow0 + (int)__umul24((unsigned)vclip, (unsigned)fstride);
#pragma unroll
      for (int fs = 0; fs < 2; ++fs) {
        const int fr = pw + 4 * fs;
        const int voff = real && fr < it.nf ? voff_ok : (int)0x80000000;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int so = sc(s0 + fr * fstride + e * it.HW * 4);
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(xr[set][fs][e]) : "v"(voff), "s"(desc), "s"(so) : "memory");
        }
      }
    };
    auto advance = [&]() {
      if (++ld_kt == it.nkt) {
        ld_kt = 0;
        ld_l += G;
        if (ld_l < p.total_blocks) { pc_locate(p, ld_l, it); geometry(); }
      }
    };
    char* const pl_wr = planes + (kp >> 2) * PC_GS + 4 * pq * 16 + (kp & 3) * 4;       // + stage, + frame, + plane, + pixel i * 16
    auto cut_tile = [&](const int set, int stage) {
      char* const base = pl_wr + stage * PC_STAGE;
#pragma unroll
      for (int fs = 0; fs < 2; ++fs) {
        char* const fb = base + (pw + 4 * fs) * PC_FRAME;             // (producer 3, slot 1: frame slot 7 -- never read)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned xa = xr[set][fs][0][i], xb = xr[set][fs][1][i];
          const unsigned h0 = xa & 0xffff0000u, h1 = xb & 0xffff0000u;
          const float r0 = __uint_as_float(xa) - __uint_as_float(h0), r1 = __uint_as_float(xb) - __uint_as_float(h1);
          const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
          const float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);      // <= 8 significant bits: the low halves are zero
          *reinterpret_cast<unsigned*>(fb + i * 16) = __builtin_amdgcn_perm(h1, h0, 0x07060302);
          *reinterpret_cast<unsigned*>(fb + i * 16 + PC_PLANE) = __builtin_amdgcn_perm(m1, m0, 0x07060302);
          *reinterpret_cast<unsigned*>(fb + i * 16 + 2 * PC_PLANE) = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302);
        }
      }
    };
    // prime: tiles 0 and 1 (four loads each -- every slot below issues four more: "all but my four newest" is the tile being cut)
    load_tile(0, true);
    advance();
    load_tile(1, 1 < T);
    advance();
    // slots in pairs, register set 0 then set 1, straight-line (a set picked by a branch on the slot's parity made hipcc merge the two
    // sets' registers through COPIES -- of registers whose loads were still in flight: stale data, silently)
    int stage = 0;
#ifdef OFFK_PT_TIMING
    unsigned long long tp[3] = {0, 0, 0}, tc_ = __builtin_readcyclecounter();
#define OFFK_PTICK(i) { const unsigned long long c_ = __builtin_readcyclecounter(); tp[i] += c_ - tc_; tc_ = c_; }
#else
#define OFFK_PTICK(i)
#endif
    auto slot = [&](const int set, int s) {
      if (s < T) {
        if (set) asm volatile("s_waitcnt vmcnt(4)" : "+v"(xr[1][0][0]), "+v"(xr[1][0][1]), "+v"(xr[1][1][0]), "+v"(xr[1][1][1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(xr[0][0][0]), "+v"(xr[0][0][1]), "+v"(xr[0][1][0]), "+v"(xr[0][1][1]) :: "memory");
        OFFK_PTICK(0)
#if !(defined(OFFK_PC_EXP) && (OFFK_PC_EXP & 1))     /* timing experiment 1: the producers cut nothing */
        cut_tile(set, stage);
#endif
#if !(defined(OFFK_PC_EXP) && (OFFK_PC_EXP & 2))     /* timing experiment 2: ... and load only zeros */
        load_tile(set, s + 2 < T);
#else
        load_tile(set, false);
#endif
        if (s + 2 < T) advance();
        stage = stage == PC_NST - 1 ? 0 : stage + 1;
        OFFK_PTICK(1)
      }
      __syncthreads();
      OFFK_PTICK(2)
    };
    for (int s = 0; s < T + 2; s += 2) {
      slot(0, s);
      if (s + 1 < T + 2) slot(1, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0][0]), "+v"(xr[0][0][1]), "+v"(xr[0][1][0]), "+v"(xr[0][1][1]),
                 "+v"(xr[1][0][0]), "+v"(xr[1][0][1]), "+v"(xr[1][1][0]), "+v"(xr[1][1][1]) :: "memory");
#ifdef OFFK_PT_TIMING
    if (p.dbg && threadIdx.x == 256) { atomicAdd(p.dbg + 20, tp[0]); atomicAdd(p.dbg + 21, tp[1]); atomicAdd(p.dbg + 22, tp[2]); atomicAdd(p.dbg + 23, (unsigned long long)T); atomicAdd(p.dbg + 24, 1ull); }
#endif
    return;
  }

  // ================================================= consumers =================================================
  const int li = lane & 15, lg = lane >> 4;
  const int ctd = wave & 1, fd0 = wave >> 1;          // the wave's down channel tile and the parity of its down frames (fd0 + 2 i)
  // weights: image [kt][slab (4 gen + 1 down)][ct][plane][lane] x 16 B (pw_pack_split16_kernel); sets alternate with the tile
  u32x4 wg[2][2][3], wd[3];          // (the down weights in ONE set: their chain ends with unit 3, the next tile's are loaded behind it)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      wd[q] = u32x4{0u, 0u, 0u, 0u};
      wg[a][0][q] = u32x4{0u, 0u, 0u, 0u}; wg[a][1][q] = u32x4{0u, 0u, 0u, 0u};
    }
  const int wlane = lane * 16;
  // weight cursor: the tile behind the one being multiplied
  int w_l = (int)blockIdx.x, w_kt = 0, w_nkt = 0;
  i32x4 wdesc = {0, 0, 0, 0};
  auto w_site = [&]() {
    int si = 0;
#pragma unroll
    for (int i = 1; i < kNumSites; ++i)
      if (i < p.nsites && w_l >= p.s[i].blk_begin) si = i;
    const PtSite& S = p.s[sc(si)];
    const unsigned long long wa = reinterpret_cast<unsigned long long>(S.wt16s);
    wdesc = i32x4{sc((int)(unsigned)wa), sc((int)(unsigned)(wa >> 32) & 0xffff), sc(kUnitCh * S.C * 6), 0x00020000};
    w_nkt = S.C / BK;
  };
  auto w_advance = [&]() {
    if (++w_kt == w_nkt) {
      w_kt = 0;
      w_l += G;
      if (w_l < p.total_blocks) w_site();
    }
  };
  auto load_wg1 = [&](const int set, const int n) {       // n = ct * 3 + plane, of the tile under the weight cursor
    const int so = sc(((w_kt * 5 + wave) * 6 + n) * 1024);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wg[set][n / 3][n % 3]) : "v"(wlane), "s"(wdesc), "s"(so));
  };
  auto load_wd1 = [&](const int q) {
    const int so = sc((((w_kt * 5 + 4) * 2 + ctd) * 3 + q) * 1024);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wd[q]) : "v"(wlane), "s"(wdesc), "s"(so));
  };
#define OFFK_WAIT_W(set) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[set][0][0]), "+v"(wg[set][0][1]), "+v"(wg[set][0][2]), "+v"(wg[set][1][0]), \
                                      "+v"(wg[set][1][1]), "+v"(wg[set][1][2]), "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]) :: "memory")

  f32x4 ag[PC_FT][2], ad[4], t[2][6];          // t[unit parity][chain a of tile 0..2 | chain b of tile 0..2]
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < PC_FT; ++j) { ag[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) ad[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 6; ++c) { t[0][c] = f32x4{0.f, 0.f, 0.f, 0.f}; t[1][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  zero_acc();

  // B operands out of the ring: lane (pixel li, k group lg) reads 16 B of a (frame, plane)
  const char* const xrd = planes + lg * PC_GS + li * 16;
  const int xoffD = fd0 * PC_FRAME;                                     // down frame i: xoffD + 2 i frames; the fourth clamped to frame 6
  const int xoffD3 = min(fd0 + 6, PC_FT - 1) * PC_FRAME;
  auto rdx = [&](u32x4 (&x)[3], const char* sb, int foff) {
#pragma unroll
    for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const u32x4*>(sb + foff + q * PC_PLANE);
  };
  auto mf = [&](f32x4 c, const u32x4& a, const u32x4& bb) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0);
  };
  auto fold2 = [&](f32x4& acc, const f32x4& tv, const int hi) {      // half of a scratch tile into its accumulator (two v_add_f32)
    if (hi) { acc.z += tv.z; acc.w += tv.w; } else { acc.x += tv.x; acc.y += tv.y; }
    asm volatile("" : "+v"(acc));           // the update stays where it is written
  };
  auto fold = [&](f32x4& acc, const f32x4& tv) { fold2(acc, tv, 0); fold2(acc, tv, 1); };
  auto rd1 = [&](u32x4& dst, const char* sb, int off) { dst = *reinterpret_cast<const u32x4*>(sb + off); };
  u32x4 x[2][2][3];                          // [unit parity][operand A / B of the unit][plane]
#define OFFK_SB __builtin_amdgcn_sched_barrier(0)
#ifdef OFFK_PC_DBG_NOPRE
#define OFFK_PC_NEXT_OPERANDS (void)nb;
#else
#define OFFK_PC_NEXT_OPERANDS rdx(x[0][0], nb, 0); rdx(x[0][1], nb, xoffD);
#endif

  // the operands of unit 0 of the first tile become readable behind the second warm-up barrier
  w_site();
#pragma unroll
  for (int n = 0; n < 6; ++n) load_wg1(0, n);
#pragma unroll
  for (int q = 0; q < 3; ++q) load_wd1(q);
  w_advance();
  __syncthreads();
  __syncthreads();
  int stage = 0;
  rdx(x[0][0], xrd, 0);
  rdx(x[0][1], xrd, xoffD);

  // one step: tile (item, kt) out of ring stage `stage`, weights of set WS; loads the weights of the next tile into set WS ^ 1 and reads
  // the first operands of the next tile (stage + 1: complete since the barrier before this step)
  auto step = [&](const int WS) {
    const char* const sb = xrd + stage * PC_STAGE;
    const int nstage = stage == PC_NST - 1 ? 0 : stage + 1;
    const char* const nb = xrd + nstage * PC_STAGE;
    OFFK_WAIT_W(WS);
#ifdef OFFK_PC_DBG_NOPRE
    rdx(x[0][0], sb, 0);
    rdx(x[0][1], sb, xoffD);
#endif
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // unit u: three tiles (weights W, operand X).  The six products of a tile are TWO chains of three (a: w_l x_h, w_h x_l, w_m x_m;
    // b: w_m x_h, w_h x_m, w_h x_h; planes 0 = h, 1 = m, 2 = l) -- six chains per unit: a dependent v_mfma_f32_16x16x32_bf16 is ~5 issue
    // slots away (probe_vmem_issue.txt: 25.8 / 19.7 / 13.7 cycles per MFMA with 3 / 4 / 6 chains on a wave that has its SIMD to itself), and
    // ONE wave per SIMD multiplies here.  The tile's two scratch tiles are summed and added to the accumulator during the NEXT unit: the
    // running sum is still rounded once per 32 k.  Nothing covers an instruction that does not fit into the 8 issue cycles an MFMA leaves
    // free of its 16: behind MFMA n at most two vector adds (n = 0..11: the folds), one LDS read (12..17: the operands of the next unit) and
    // one weight load (16, 17).
#define OFFK_FOLD(n, F, TP, c)                                                                                   \
    if ((n) == 0) { TP[c].x += TP[c + 3].x; TP[c].y += TP[c + 3].y; }                                            \
    if ((n) == 1) { TP[c].z += TP[c + 3].z; TP[c].w += TP[c + 3].w; }                                            \
    if ((n) == 2) { F.x += TP[c].x; F.y += TP[c].y; asm volatile("" : "+v"(F)); }                                \
    if ((n) == 3) { F.z += TP[c].z; F.w += TP[c].w; asm volatile("" : "+v"(F)); }
#define OFFK_UNIT(TC, TP, W0, X0, F0, W1, X1, F1, W2, X2, F2, NA, NAOFF, NB, NBOFF, NBASE, LD0, LD1)             \
    {                                                                                                            \
      TC[0] = mf(z, W0[2], X0[0]);      OFFK_FOLD(0, F0, TP, 0) OFFK_SB;                                         \
      TC[1] = mf(z, W1[2], X1[0]);      OFFK_FOLD(1, F0, TP, 0) OFFK_SB;                                         \
      TC[2] = mf(z, W2[2], X2[0]);      OFFK_FOLD(0, F1, TP, 1) OFFK_SB;                                         \
      TC[3] = mf(z, W0[1], X0[0]);      OFFK_FOLD(1, F1, TP, 1) OFFK_SB;                                         \
      TC[4] = mf(z, W1[1], X1[0]);      OFFK_FOLD(0, F2, TP, 2) OFFK_SB;                                         \
      TC[5] = mf(z, W2[1], X2[0]);      OFFK_FOLD(1, F2, TP, 2) OFFK_SB;                                         \
      TC[0] = mf(TC[0], W0[0], X0[2]);  OFFK_FOLD(2, F0, TP, 0) OFFK_SB;                                         \
      TC[1] = mf(TC[1], W1[0], X1[2]);  OFFK_FOLD(3, F0, TP, 0) OFFK_SB;                                         \
      TC[2] = mf(TC[2], W2[0], X2[2]);  OFFK_FOLD(2, F1, TP, 1) OFFK_SB;                                         \
      TC[3] = mf(TC[3], W0[0], X0[1]);  OFFK_FOLD(3, F1, TP, 1) OFFK_SB;                                         \
      TC[4] = mf(TC[4], W1[0], X1[1]);  OFFK_FOLD(2, F2, TP, 2) OFFK_SB;                                         \
      TC[5] = mf(TC[5], W2[0], X2[1]);  OFFK_FOLD(3, F2, TP, 2) OFFK_SB;                                         \
      TC[0] = mf(TC[0], W0[1], X0[1]);  rd1(NA[0], NBASE, NAOFF); OFFK_SB;                                       \
      TC[1] = mf(TC[1], W1[1], X1[1]);  rd1(NA[1], NBASE, NAOFF + PC_PLANE); OFFK_SB;                            \
      TC[2] = mf(TC[2], W2[1], X2[1]);  rd1(NA[2], NBASE, NAOFF + 2 * PC_PLANE); OFFK_SB;                        \
      TC[3] = mf(TC[3], W0[0], X0[0]);  rd1(NB[0], NBASE, NBOFF); OFFK_SB;                                       \
      TC[4] = mf(TC[4], W1[0], X1[0]);  rd1(NB[1], NBASE, NBOFF + PC_PLANE); LD0; OFFK_SB;                       \
      TC[5] = mf(TC[5], W2[0], X2[0]);  rd1(NB[2], NBASE, NBOFF + 2 * PC_PLANE); LD1; OFFK_SB;                   \
    }
    const u32x4 (&g0)[3] = wg[WS][0];
    const u32x4 (&g1)[3] = wg[WS][1];
    const u32x4 (&dw)[3] = wd;
    // unit 0: frame 0 tiles 0, 1 + down frame 0 (folds: the chains of unit 5 of the step before: frame 5 tile 1, frame 6 tiles 0, 1)
    OFFK_UNIT(t[0], t[1], g0, x[0][0], ag[5][1], g1, x[0][0], ag[6][0], dw, x[0][1], ag[6][1],
              x[1][0], 1 * PC_FRAME, x[1][1], xoffD + 2 * PC_FRAME, sb, load_wg1(WS ^ 1, 0), load_wg1(WS ^ 1, 1))
    OFFK_UNIT(t[1], t[0], g0, x[1][0], ag[0][0], g1, x[1][0], ag[0][1], dw, x[1][1], ad[0],
              x[0][0], 2 * PC_FRAME, x[0][1], xoffD + 4 * PC_FRAME, sb, load_wg1(WS ^ 1, 2), load_wg1(WS ^ 1, 3))
    OFFK_UNIT(t[0], t[1], g0, x[0][0], ag[1][0], g1, x[0][0], ag[1][1], dw, x[0][1], ad[1],
              x[1][0], 3 * PC_FRAME, x[1][1], xoffD3, sb, load_wg1(WS ^ 1, 4), load_wg1(WS ^ 1, 5))
    OFFK_UNIT(t[1], t[0], g0, x[1][0], ag[2][0], g1, x[1][0], ag[2][1], dw, x[1][1], ad[2],
              x[0][0], 4 * PC_FRAME, x[0][1], 5 * PC_FRAME, sb, (void)0, (void)0)
    // unit 4: frame 4 tiles 0, 1, frame 5 tile 0 (the down chain is done: its weights of the next tile)
    OFFK_UNIT(t[0], t[1], g0, x[0][0], ag[3][0], g1, x[0][0], ag[3][1], g0, x[0][1], ad[3],
              x[1][0], 5 * PC_FRAME, x[1][1], 6 * PC_FRAME, sb, load_wd1(0), load_wd1(1))
    // unit 5: frame 5 tile 1, frame 6 tiles 0, 1; reads the first operands of the next tile
    OFFK_UNIT(t[1], t[0], g1, x[1][0], ag[4][0], g0, x[1][1], ag[4][1], g1, x[1][1], ag[5][0],
              x[0][0], 0, x[0][1], xoffD, nb, load_wd1(2), (void)0)
#undef OFFK_UNIT
#undef OFFK_FOLD
    stage = nstage;
  };

  // Every item starts on weight set 0: two copies of the step in sequence (set 0, set 1, ...; chosen per tile by a branch the two
  // copies met at a merge that cost hipcc ~120 spilled registers); an item with an odd number of K-tiles leaves the next item's first
  // weights in set 1 -- moved over once (24 register moves per item of 19 steps).
#ifdef OFFK_PT_TIMING
  unsigned long long tq[3] = {0, 0, 0}, tc_ = __builtin_readcyclecounter();
  int n_items = 0;
#define OFFK_CTICK(i) { const unsigned long long c_ = __builtin_readcyclecounter(); tq[i] += c_ - tc_; tc_ = c_; }
#else
#define OFFK_CTICK(i)
#endif
  int gtile = 0;
  for (int l = (int)blockIdx.x; l < p.total_blocks; l += G) {
    PcItem it;
    pc_locate(p, l, it);
    bool odd;
    OFFK_CTICK(2)
    for (int kt = 0;;) {
      step(0);
      if (++gtile < T) w_advance();
      OFFK_CTICK(0)
      __syncthreads();
      OFFK_CTICK(1)
      if (++kt == it.nkt) { odd = true; break; }
      step(1);
      if (++gtile < T) w_advance();
      OFFK_CTICK(0)
      __syncthreads();
      OFFK_CTICK(1)
      if (++kt == it.nkt) { odd = false; break; }
    }
    if (odd) {
      OFFK_WAIT_W(1);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) wg[0][c][q] = wg[1][c][q];
    }
    // the last unit's scratch tiles
    fold(ag[5][1], t[1][0]); fold(ag[6][0], t[1][1]); fold(ag[6][1], t[1][2]);
    fold(ag[5][1], t[1][3]); fold(ag[6][0], t[1][4]); fold(ag[6][1], t[1][5]);

    // ---- epilogue (as pw_tdiff_split_kernel): lane = (pixel li, channels 4 lg .. + 3 of a channel tile) ----
    const PtSite& S = p.s[it.si];
    const int rmask = (1 << it.rsh) - 1;
    const int qe = it.qr0 + (li >> 2);
    const bool qn_e = it.qpc && qe >= it.qpc;
    const int bl = it.qpc ? it.b + (int)qn_e : it.b + (li >> it.rsh);
    const int pixl = it.qpc ? 4 * (qe - (qn_e ? it.qpc : 0)) + (li & 3) : it.q0 + (li & rmask);
    const size_t pair0 = (size_t)bl * (L - 1) + it.t0;
    const bool pix_ok = pixl < it.HW && bl < p.B;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x4 bg = *reinterpret_cast<const f32x4*>(S.bias + wave * 32 + 16 * ct + 4 * lg);
#pragma unroll
      for (int j = 0; j < PC_FT; ++j) {
        const f32x4 v = ag[j][ct] + bg;
        ag[j][ct] = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      }
    }
#pragma unroll
    for (int j = 0; j + 1 < PC_FT; ++j)
      if (j + 1 < it.nf && pix_ok) {
        float* const trow = S.M + ((pair0 + j) * it.HW + pixl) * S.m_cs + S.m_coff + kDownCh + wave * 32 + 4 * lg;
        *reinterpret_cast<f32x4*>(trow) = ag[j + 1][0] - ag[j][0];
        *reinterpret_cast<f32x4*>(trow + 16) = ag[j + 1][1] - ag[j][1];
      }
    {
      const f32x4 bd = *reinterpret_cast<const f32x4*>(S.bias_down + 16 * ctd + 4 * lg);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = fd0 + 2 * i;                   // (waves 2, 3, i = 3: j = 7 >= nf -- the duplicate tile is dropped here)
        // the frame shared with the next temporal group belongs to that group
        if (j < it.nf && (it.last_group || j < PC_FT - 1) && pix_ok) {
          const int dr = pc_down_row(bl, it.t0 + j, L, p.P, p.slice_mode);
          if (dr >= 0) *reinterpret_cast<f32x4*>(S.D + ((size_t)dr * it.HW + pixl) * kDownCh + 16 * ctd + 4 * lg) = ad[i] + bd;
        }
      }
    }
    zero_acc();
#ifdef OFFK_PT_TIMING
    ++n_items;
#endif
    OFFK_CTICK(2)
  }
#ifdef OFFK_PT_TIMING
  if (p.dbg && threadIdx.x == 0) { atomicAdd(p.dbg + 25, tq[0]); atomicAdd(p.dbg + 26, tq[1]); atomicAdd(p.dbg + 27, tq[2]); atomicAdd(p.dbg + 28, (unsigned long long)n_items); }
#endif
#undef OFFK_SB
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wg[0][0][0]), "+v"(wg[0][0][1]), "+v"(wg[0][0][2]), "+v"(wg[0][1][0]), "+v"(wg[0][1][1]), "+v"(wg[0][1][2]),
               "+v"(wg[1][0][0]), "+v"(wg[1][0][1]), "+v"(wg[1][0][2]), "+v"(wg[1][1][0]), "+v"(wg[1][1][1]), "+v"(wg[1][1][2]) :: "memory");
  asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]));
#undef OFFK_WAIT_W
}

// p: block layout filled by pw_tdiff_launch (the 16-pixel form's); persistent: one block of eight waves per CU
hipError_t pw_tdiff_pc_launch(const PtParams& p, int n_cu, hipStream_t st) {
  hipError_t er = lds_attr_once(reinterpret_cast<const void*>(pw_tdiff_pc_kernel), PC_LDS);
  if (er != hipSuccess) return er;
  const int grid = p.total_blocks < n_cu ? p.total_blocks : n_cu;
  hipLaunchKernelGGL(pw_tdiff_pc_kernel, dim3(grid), dim3(512), PC_LDS, st, p);
  return hipGetLastError();
}

}  // namespace offk
