// K4gs: the batched GEMMs of a convolution on a Winograd path (wino_gemm.hip: M[point] = V[point] . U[point]^T) in split-fp32
// arithmetic on the bf16 matrix pipe -- the arithmetic of pw_tdiff_split.hip: every fp32 operand as three bf16 planes (h + m + l = the
// value, exactly), six plane products per multiply on v_mfma_f32_16x16x32_bf16 into TWO running accumulators per tile (round 6: the leading
// product w_h x_h in one, the five small ones in the other, added once per item -- tools/probe_split_mfma.hip mode 6; round 5 summed the
// six products of a 32-k step from zero in a scratch tile and folded it into one accumulator per step: the same registers, 4 NT more
// vector adds per step).  Reference arithmetic: the fp32 contractions of
// RGB_OFF.py:762, 766, 775-777, 833, 837 in their Winograd forms (winograd.hip).
//
// Structure.  One persistent launch (as wino_gemm.hip: a block works through its items (point, m-tile, n-tile) as one stream of
// K-tiles, nothing is waited for at an item boundary), 64 rows x 128 output channels per item, two blocks per CU.
//   * U: cut once by the library (wino_pack_split_kernel) into the plane image [problem][K-tile][channel tile][plane][lane] x 16 B in
//     operand order; a wave owns two channel tiles (32 channels) of all 64 rows and loads their six 1-KB pieces of the NEXT K-tile
//     straight from L2 into registers;
//   * V: thread (row = tid / 8 (+ 32), k chunk = tid % 8) loads 16 B of the K-tile TWO steps ahead into registers, cuts them one step
//     ahead (18 vector instructions per four values) and writes 8 B per plane into the plane image of the next K-tile in LDS:
//     [row tile][plane][k group g][slot = row ^ 2 g] x 16 B -- conflict-free for the cut's ds_write_b64 (banks mod 32, sixteen-lane
//     groups = two rows x eight chunks) and for the MFMA operand's ds_read_b128 (MI355X_MICROARCH.md, LDS);
//   * a step = one K-tile: 48 MFMAs per wave, product-major (eight independent chains of six), the cut / the loads / the folds in
//     slices behind single MFMAs; one barrier per step; the three cursors (V two tiles ahead, U one, the multiply) walk the same item
//     list, an item's descriptors travel from the first to the last through two sets of pending registers.
// Weights = A operand, V rows = B operand: an accumulator lane holds four consecutive output channels of one row (16-byte stores).
#include <climits>
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

#ifndef OFFK_GS_EXP
#define OFFK_GS_EXP 0      /* timing experiments (tools/build_one.py -DOFFK_GS_EXP=mask): 1 no cut, 4 no U loads, 8 no V loads, 16 U from one K-tile */
#endif

namespace offk {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int GS_BM = 64;                   // rows per item; channels per item: 64 CT (four waves x CT channel tiles of 16)
constexpr int GS_GRP = 256;                 // one k group of a plane: 16 row slots x 16 B (8 bf16 = k 8g .. 8g + 7)
constexpr int GS_PLANE = 4 * GS_GRP;        // 32 k
constexpr int GS_RT = 3 * GS_PLANE;         // a row tile (16 rows): planes h, m, l
constexpr int GS_STAGE = 4 * GS_RT;         // 64 rows
constexpr int GS_LDS = 2 * GS_STAGE;        // 24576 B
#ifndef OFFK_GS_BLOCKS
#define OFFK_GS_BLOCKS 3      /* blocks per CU the kernel is compiled for */
#endif
#ifndef OFFK_GS_BLOCKS_CT1
#define OFFK_GS_BLOCKS_CT1 3
#endif
constexpr int GS_RESIDENT = OFFK_GS_BLOCKS * 256;        // blocks the chip holds at a time = the persistent grid (CT = 2: 162 VGPRs)
constexpr int GS_RESIDENT_CT1 = OFFK_GS_BLOCKS_CT1 * 256;    // CT = 1: 120 VGPRs
constexpr int GS_OOB = (int)0x80000000;     // a per-lane offset past every descriptor: loads return zeros, stores are dropped
}  // namespace

// CT = channel tiles per wave: 2 (128 channels per item), or 1 (64: the 7x7 conv's Co = 64 -- half the MFMAs per cut value and per step)
// EPI: the epilogue of a 1x1 convolution (conv_igemm.hip's, for the launches a split-fp32 handle sends here): x rows / y rows with a
// channel stride and offset, + bias, ReLU, and the folded average pool of a head (per 32-row slab and channel the sums over the slab's
// rows of its first / second image)
template <int CT, bool EPI>
__global__ __launch_bounds__(256, OFFK_GS_BLOCKS) void wino_gemm_split_kernel(WinoGemmArgs p) {
  constexpr int NT = 4 * CT;                // the wave's accumulator tiles: (row tile i / CT, channel tile i % CT)
  constexpr int GS_BN = 64 * CT;
  extern __shared__ __attribute__((aligned(16))) char gs_planes[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  // ---- V loader / cut: thread = (row lrow (+ 32), k chunk lc) ----
  const int lrow = tid >> 3, lc = tid & 7, lgp = lc >> 1;
  char* const pl_wr = gs_planes + (lrow >> 4) * GS_RT + lgp * GS_GRP + (((lrow & 15) ^ (2 * lgp)) << 4) + (lc & 1) * 8;
  const char* const xrd = gs_planes + lg * GS_GRP + ((li ^ (2 * lg)) << 4);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_planes), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);

  // ---- items (the group tables as scalars of their own, as wino_gemm.hip) ----
  const int gx = p.gm * p.gn;
  const int nK0 = p.g_K[0], nK1 = p.g_K[1], nK2 = p.g_K[2], nK3 = p.g_K[3];
  const int nB0 = p.g_batch[0], nB1 = p.g_batch[1], nB2 = p.g_batch[2];
  const long long oX0 = p.g_x[0], oX1 = p.g_x[1], oX2 = p.g_x[2], oX3 = p.g_x[3];
  const long long oW0 = p.g_w[0], oW1 = p.g_w[1], oW2 = p.g_w[2], oW3 = p.g_w[3];
  const long long oY0 = p.g_y[0], oY1 = p.g_y[1], oY2 = p.g_y[2], oY3 = p.g_y[3];
  const int ngroups = p.ngroups, argM = p.M, argCo = p.Co, total_items = p.total_items;
  const int gm = p.gm;
  const int main_items = (total_items / gx / 8) * 8 * gx;      // the items of the problems that go to XCDs whole
  const int x_rs = p.x_rs, y_rs = EPI ? p.y_rs : p.Co;      // (EPI: x_coff / y_coff are part of the descriptors' base addresses)
  const int ukstep = (argCo >> 4) * 3072;          // bytes of one K-tile of a problem's plane image
  const int grid = (int)gridDim.x;
  auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

  // V cursor (two tiles ahead of the multiply)
  int x_l = (int)blockIdx.x, x_t = 0, x_nkt = 0, x_soff = 0, x_off[2] = {GS_OOB, GS_OOB};
  // what the V cursor found out about its item, for the U cursor (pu_*, one step behind) and the multiply (pc_*, via pd_*, two behind)
  int pu_soff = 0, pu_nkt = 0, pc_ysoff = 0, pc_m0 = 0, pc_n0 = 0;
  int pd_ysoff = 0, pd_m0 = 0, pd_n0 = 0, pd_nkt = 0;
  auto locate = [&](int l) {
    // Item order.  Block b runs on XCD b & 7 and takes items b, b + grid, ...: item l belongs to XCD l & 7, and the blocks of an XCD are
    // at positions j = l >> 3 .. + 63 of their XCD's list at any one time.  Whole problems go to one XCD (problem s = 8 (j / gx) + xcd,
    // in the groups' order: every XCD gets the same mix of K), so the six m-tiles that read the same U tile and the n-tiles that read
    // the same V rows meet in one L2 (with wino_gemm.hip's order -- a problem's items dealt over all XCDs -- every U tile was fetched by
    // six L2s: the weight loads were 69 of 162 us at K = 832).  The problems % 8 last problems are dealt item by item.
    int prob, lid;
    if (l < main_items) { const int j = l >> 3, jq = sc(j / gx); prob = jq * 8 + (l & 7); lid = j - jq * gx; }
    else { const int lr = l - main_items; const int pq = sc(lr / gx); prob = (main_items / gx) + pq; lid = lr - pq * gx; }
    prob = sc(prob);
    const int nt = sc(lid / gm), mt = lid - nt * gm;          // m-tile fastest
    int b = prob, K = nK0;
    long long ox = oX0, ow = oW0, oy = oY0;
    if (ngroups > 1 && prob >= nB0) { b = prob - nB0; K = nK1; ox = oX1; ow = oW1; oy = oY1; }
    if (ngroups > 2 && prob >= nB0 + nB1) { b = prob - nB0 - nB1; K = nK2; ox = oX2; ow = oW2; oy = oY2; }
    if (ngroups > 3 && prob >= nB0 + nB1 + nB2) { b = prob - nB0 - nB1 - nB2; K = nK3; ox = oX3; ow = oW3; oy = oY3; }
    const int m0 = mt * GS_BM, n0 = nt * GS_BN;
    x_soff = sc((int)((ox + (long long)b * argM * K) * 4));
    x_nkt = sc(K >> 5);
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int m = m0 + lrow + 32 * r2;
      x_off[r2] = m < argM ? (m * (EPI ? x_rs : K) + lc * 4) * 4 : GS_OOB;
    }
    pu_soff = sc((int)((ow + (long long)b * argCo * K) * 6) + ((n0 >> 4) + CT * wave) * 3072);
    pu_nkt = x_nkt;
    pc_ysoff = sc((int)((oy + (long long)b * argM * argCo) * 4));
    pc_m0 = m0; pc_n0 = n0;
  };
  // U cursor
  int u_l = x_l, u_t = 0, u_nkt = 0, u_soff = 0, u_voff = lane * 16;
  // the multiply
  int c_l = x_l, c_t = 0, c_nkt = 0, c_ysoff = 0, c_m0 = 0, c_n0 = 0;

  u32x4 xr[2][2];                  // V registers [set = K-tile parity][row half]
  u32x4 wr[2][CT][3];              // U planes [set][channel tile][plane]
  auto load_x = [&](const int set, const int r) {
    xr[set][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, x_off[r], x_soff + x_t * 128, 0));
  };
  auto load_u = [&](const int set, const int n) {      // n = ct * 3 + plane
#if defined(OFFK_GS_EXP) && (OFFK_GS_EXP & 16)     /* timing experiment: every step re-reads the item's first K-tile (cache-resident) */
    wr[set][n / 3][n % 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, u_voff, u_soff + n * 1024, 0));
#else
    wr[set][n / 3][n % 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, u_voff, u_soff + u_t * ukstep + n * 1024, 0));
#endif
  };
  auto adv_x = [&]() {
    ++x_t;
    if (x_t == x_nkt) {
      x_l += grid; x_t = 0;
      if (x_l < total_items) locate(x_l);
      else { x_off[0] = GS_OOB; x_off[1] = GS_OOB; x_nkt = INT_MAX; }
    }
  };
  auto adv_u = [&]() {
    ++u_t;
    if (u_t == u_nkt) {
      u_l += grid; u_t = 0;
      if (u_l < total_items) { u_soff = pu_soff; u_nkt = pu_nkt; pd_ysoff = pc_ysoff; pd_m0 = pc_m0; pd_n0 = pc_n0; pd_nkt = pu_nkt; }
      else { u_voff = GS_OOB; u_nkt = INT_MAX; }
    }
  };

  // ---- the cut of register set `set`, row half r, in seven slices (+ a store in the last three): 4 and, 2 packed subtractions, 4 and,
  //      2 packed subtractions, 3 x (2 byte permutes + ds_write_b64) -- 18 vector instructions per four values ----
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  unsigned ch[4], cm[4];
  f32x2 cr[2], cl[2];
  auto cut_slice = [&](const int s, const int set, const int r, const int st) {
    const u32x4& x = xr[set][r];
    if (s == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ch[e] = x[e] & 0xffff0000u;
    } else if (s == 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        cr[e] = f32x2{__uint_as_float(x[2 * e]), __uint_as_float(x[2 * e + 1])} - f32x2{__uint_as_float(ch[2 * e]), __uint_as_float(ch[2 * e + 1])};
    } else if (s == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cm[e] = __float_as_uint(cr[e >> 1][e & 1]) & 0xffff0000u;
    } else if (s == 3) {
#pragma unroll
      for (int e = 0; e < 2; ++e) cl[e] = cr[e] - f32x2{__uint_as_float(cm[2 * e]), __uint_as_float(cm[2 * e + 1])};     // <= 8 significant bits left
    } else {
      char* dst = pl_wr + st * GS_STAGE + r * 2 * GS_RT + (s - 4) * GS_PLANE;
      u32x2 d;
      if (s == 4) d = u32x2{__builtin_amdgcn_perm(x[1], x[0], 0x07060302), __builtin_amdgcn_perm(x[3], x[2], 0x07060302)};
      else if (s == 5) d = u32x2{__builtin_amdgcn_perm(cm[1], cm[0], 0x07060302), __builtin_amdgcn_perm(cm[3], cm[2], 0x07060302)};
      else d = u32x2{__builtin_amdgcn_perm(__float_as_uint(cl[0][1]), __float_as_uint(cl[0][0]), 0x07060302),
                     __builtin_amdgcn_perm(__float_as_uint(cl[1][1]), __float_as_uint(cl[1][0]), 0x07060302)};
      *reinterpret_cast<u32x2*>(dst) = d;
    }
  };
  constexpr int CUT_SLICES = 7;

  f32x4 acc[NT], acs[NT];          // acc: sum w_h x_h (and, in store_item, the item's result); acs: the five small products
#pragma unroll
  for (int i = 0; i < NT; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto mf = [&](f32x4 c, const u32x4& a, const u32x4& b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
#define OFFK_SB __builtin_amdgcn_sched_barrier(0)

  // the finished item: a lane holds channels 4 lg .. + 3 of row li.
  // MI355X + hipcc (ROCm 7.2): a VALU write to the FIRST data register of a buffer_store_dwordx4 in the instruction right behind it
  // reached memory in lanes 12-15 of every sixteen (run-to-run varying; hipcc places no wait state there when the store's soffset is a
  // register -- it had re-used the register for the next tile's address): each store is followed by s_nop 1, fenced.
  const int st_voff = (li * y_rs + CT * wave * 16 + 4 * lg) * 4;
  auto store_item = [&]() {
    const int ybase = c_ysoff + (c_m0 * y_rs + c_n0) * 4;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] += acs[i];
    if (EPI) {
      // + bias, ReLU
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + c_n0 + (CT * wave + ct) * 16 + 4 * lg);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          f32x4& a = acc[rt * CT + ct];
          a += b4;
          if (p.relu) a = f32x4{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)};
        }
      }
      if (p.pool_part) {
        // column sums per 32-row slab (row tiles 2 s, 2 s + 1), split at the image boundary inside the slab (an image has >= 32 rows: a
        // slab touches at most two); fixed order: the slab's two row tiles, then the sixteen rows of a tile by xor-shuffles 1, 2, 4, 8
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int slab = (c_m0 >> 5) + sl;
          const int b1 = ((slab * 32) / p.pool_hw + 1) * p.pool_hw;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int m = slab * 32 + h2 * 16 + li;
              const f32x4 v = acc[(2 * sl + h2) * CT + ct];
              const bool in = m < argM, first = m < b1;
              const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
              s0 += in && first ? v : z4;
              s1 += in && !first ? v : z4;
            }
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { s0[e] += __shfl_xor(s0[e], d); s1[e] += __shfl_xor(s1[e], d); }
            }
            if (li == 0 && slab * 32 < argM) {
              float* pp = p.pool_part + (size_t)slab * 2 * argCo + c_n0 + (CT * wave + ct) * 16 + 4 * lg;
              *reinterpret_cast<f32x4*>(pp) = s0;
              *reinterpret_cast<f32x4*>(pp + argCo) = s1;
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int rt = i / CT;
      const int voff = li < argM - c_m0 - rt * 16 ? st_voff : GS_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), yrs, voff, ybase + (rt * 16 * y_rs + (i % CT) * 16) * 4, 0);
      OFFK_SB;
      asm volatile("s_nop 1");
      OFFK_SB;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };

  // ---- prologue: V(0) cut into stage 0, V(1) in registers, U(0) in registers ----
  locate(x_l);
  u_soff = pu_soff; u_nkt = pu_nkt;
  c_ysoff = pc_ysoff; c_m0 = pc_m0; c_n0 = pc_n0; c_nkt = pu_nkt;
  load_x(0, 0); load_x(0, 1);
  adv_x();
  load_x(1, 0); load_x(1, 1);
  adv_x();
#pragma unroll
  for (int n = 0; n < 3 * CT; ++n) load_u(0, n);
  adv_u();
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < CUT_SLICES; ++s) cut_slice(s, 0, r, 0);
  __syncthreads();

  // One step: K-tile c_t of the multiply's item out of plane stage ST with the U registers of set ST; the cut of the next tile (register
  // set ST ^ 1) into plane stage ST ^ 1; V two tiles ahead into register set ST; U of the next tile into set ST ^ 1.
  auto step = [&](const int ST) -> bool {
    u32x4 xb[4][3];
    const char* const rd = xrd + ST * GS_STAGE;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xb[rt][0] = *reinterpret_cast<const u32x4*>(rd + rt * GS_RT);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xb[rt][2] = *reinterpret_cast<const u32x4*>(rd + rt * GS_RT + 2 * GS_PLANE);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xb[rt][1] = *reinterpret_cast<const u32x4*>(rd + rt * GS_RT + GS_PLANE);
    OFFK_SB;
    // planes 0 = h, 1 = m, 2 = l; smallest products first: w_l x_h, w_h x_l, w_m x_m, w_m x_h, w_h x_m, w_h x_h
    constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, XP[6] = {0, 2, 1, 0, 1, 0};
    // the step's side jobs -- V loads [2], cut slices [14], U loads [3 CT] -- go out behind the MFMAs of the first five products, in
    // that order, job j behind MFMA j * NS / NJ
    constexpr int NJ = 2 + 3 * CT + 2 * CUT_SLICES, NS = 5 * NT;
// (the U loads LAST, planes l, m, h: the registers of the current tile's planes come free in that order -- w_l behind the first
    //  product, w_m behind the fourth -- which keeps the kernel at 162 VGPRs = three blocks per CU; with the loads at the top of the step
    //  it needed 174, and the third block is worth 5 - 8 % per launch)
    auto job = [&](const int j) {
      if (j < 2) { if (!(OFFK_GS_EXP & 8)) load_x(ST, j); }
      else if (j < 2 + 2 * CUT_SLICES) { if (!(OFFK_GS_EXP & 1)) { const int c = j - 2; cut_slice(c % CUT_SLICES, ST ^ 1, c / CUT_SLICES, ST ^ 1); } }
      else if (!(OFFK_GS_EXP & 4)) { const int u = j - 2 - 2 * CUT_SLICES; load_u(ST ^ 1, (u % CT) * 3 + 2 - u / CT); }
    };
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = q * NT + i, rt = i / CT, ct = i % CT;
        if (q < 5) acs[i] = mf(acs[i], wr[ST][ct][WP[q]], xb[rt][XP[q]]);
        else acc[i] = mf(acc[i], wr[ST][ct][WP[q]], xb[rt][XP[q]]);
        if (n < NS) {
#pragma unroll
          for (int j = (n * NJ + NS - 1) / NS; j < ((n + 1) * NJ + NS - 1) / NS; ++j) job(j);
        }
        OFFK_SB;
      }
    }
    // cursors: the multiply, then U, then V (an item's descriptors are handed down in that order)
    ++c_t;
    bool go_on = true;
    if (c_t == c_nkt) {
      store_item();
      c_l += grid; c_t = 0;
      if (c_l < total_items) { c_ysoff = pd_ysoff; c_m0 = pd_m0; c_n0 = pd_n0; c_nkt = pd_nkt; }
      else go_on = false;
    }
    adv_u();
    adv_x();
    __syncthreads();
    return go_on;
  };
  for (;;) {
    if (!step(0)) break;
    if (!step(1)) break;
  }
}

// U -> plane image.  Thread = (problem, K-tile, channel tile, lane): 8 consecutive k of one output channel, three 16-byte pieces.
__global__ void wino_pack_split_kernel(const float* __restrict__ U, uint4* __restrict__ img, int Co, int K, int nproblems) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nct = Co >> 4, nkt = K >> 5;
  const long long per = (long long)nkt * nct * 64;
  if (item >= per * nproblems) return;
  const int prob = (int)(item / per);
  const int rem = (int)(item - prob * per);
  const int lane = rem & 63, ct = (rem >> 6) % nct, kt = (rem >> 6) / nct;
  const int li = lane & 15, g = lane >> 4;
  const float* row = U + ((size_t)prob * Co + ct * 16 + li) * K + kt * 32 + 8 * g;
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned x = __float_as_uint(row[e]);
    h[e] = x & 0xffff0000u;
    const float r1 = __uint_as_float(x) - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xffff0000u;
    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
  }
  auto pk = [](const unsigned (&v)[8]) {
    return uint4{(v[0] >> 16) | (v[1] & 0xffff0000u), (v[2] >> 16) | (v[3] & 0xffff0000u), (v[4] >> 16) | (v[5] & 0xffff0000u),
                 (v[6] >> 16) | (v[7] & 0xffff0000u)};
  };
  uint4* o = img + (((size_t)prob * nkt + kt) * nct + ct) * 3 * 64 + lane;
  o[0] = pk(h); o[64] = pk(m); o[128] = pk(l);
}

// the plane image of the U of one group of problems (Co x K each): 6 bytes per element, same element offsets as U
hipError_t wino_pack_split_launch(const float* U, void* img, int Co, int K, int nproblems, hipStream_t st) {
  if (Co % 16 || K % 32 || nproblems <= 0) return hipErrorInvalidValue;
  const long long items = (long long)nproblems * (K / 32) * (Co / 16) * 64;
  hipLaunchKernelGGL(wino_pack_split_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, U, reinterpret_cast<uint4*>(img), Co, K,
                     nproblems);
  return hipGetLastError();
}

bool wino_gemm_split_supported(const WinoGemmArgs& a) {
  if (!a.w_planes || a.M <= 0 || a.Co % 64 || a.ngroups < 1 || a.ngroups > 4) return false;
  long long problems = 0, xe = 0, we = 0, ye = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    if (a.g_K[g] % 32 || a.g_K[g] < 64 || a.g_batch[g] <= 0) return false;
    problems += a.g_batch[g];
    xe = std::max(xe, a.g_x[g] + (long long)a.g_batch[g] * a.M * a.g_K[g]);
    we = std::max(we, a.g_w[g] + (long long)a.g_batch[g] * a.Co * a.g_K[g]);
    ye = std::max(ye, a.g_y[g] + (long long)a.g_batch[g] * a.M * a.Co);
    if ((long long)a.M * a.g_K[g] * 4 >= 0x7fffff00ll) return false;
  }
  if (xe * 4 >= 0x7fffff00ll || we * 6 >= 0x7fffff00ll || ye * 4 >= 0x7fffff00ll) return false;
  const long long total = problems * ((a.M + GS_BM - 1) / GS_BM) * (a.Co / 64);
  return total > 0 && total < (1ll << 30);
}

hipError_t wino_gemm_split_launch(const WinoGemmArgs& a_in, hipStream_t st) {
  WinoGemmArgs a = a_in;
  if (!wino_gemm_split_supported(a)) return hipErrorInvalidValue;
  long long problems = 0, xe = 0, we = 0, ye = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    problems += a.g_batch[g];
    xe = std::max(xe, a.g_x[g] + (long long)a.g_batch[g] * a.M * a.g_K[g]);
    we = std::max(we, a.g_w[g] + (long long)a.g_batch[g] * a.Co * a.g_K[g]);
    ye = std::max(ye, a.g_y[g] + (long long)a.g_batch[g] * a.M * a.Co);
  }
  a.x_bytes = xe * 4; a.w_bytes = we * 6; a.y_bytes = ye * 4;
  a.gm = (a.M + GS_BM - 1) / GS_BM;
  const bool wide = a.Co % 128 == 0;         // 128 channels per item where Co allows it
  a.gn = a.Co / (wide ? 128 : 64);
  a.total_items = (int)(problems * a.gm * a.gn);
  const bool epi = a.epilogue != 0;
  if (epi) {
    // one problem; the rows of x / y are x_rs / y_rs floats apart and start x_coff / y_coff floats into them
    if (problems != 1 || a.x_rs < a.g_K[0] || a.y_rs < a.Co || ((a.x_rs | a.y_rs | a.x_coff | a.y_coff) & 3) || a.x_coff < 0 || a.y_coff < 0 ||
        (a.pool_part && a.pool_hw < 32))
      return hipErrorInvalidValue;
    a.x += a.x_coff; a.y += a.y_coff;
    a.x_bytes = ((long long)(a.M - 1) * a.x_rs + a.g_K[0]) * 4;
    a.y_bytes = ((long long)(a.M - 1) * a.y_rs + a.Co) * 4;
    if (a.x_bytes >= 0x7fffff00ll || a.y_bytes >= 0x7fffff00ll) return hipErrorInvalidValue;
  }
  const void* fn = wide ? (epi ? reinterpret_cast<const void*>(wino_gemm_split_kernel<2, true>) : reinterpret_cast<const void*>(wino_gemm_split_kernel<2, false>))
                        : (epi ? reinterpret_cast<const void*>(wino_gemm_split_kernel<1, true>) : reinterpret_cast<const void*>(wino_gemm_split_kernel<1, false>));
  hipError_t e = lds_attr_once(fn, GS_LDS);
  if (e != hipSuccess) return e;
  const int resident = wide ? GS_RESIDENT : GS_RESIDENT_CT1;
  const int grid = a.total_items < resident ? a.total_items : resident;
  if (wide && epi) hipLaunchKernelGGL((wino_gemm_split_kernel<2, true>), dim3(grid), dim3(256), GS_LDS, st, a);
  else if (wide) hipLaunchKernelGGL((wino_gemm_split_kernel<2, false>), dim3(grid), dim3(256), GS_LDS, st, a);
  else if (epi) hipLaunchKernelGGL((wino_gemm_split_kernel<1, true>), dim3(grid), dim3(256), GS_LDS, st, a);
  else hipLaunchKernelGGL((wino_gemm_split_kernel<1, false>), dim3(grid), dim3(256), GS_LDS, st, a);
  return hipGetLastError();
}

}  // namespace offk
