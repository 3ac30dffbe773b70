// K4g: the batched GEMMs of a convolution on a Winograd path (winograd.hip / winograd7.hip:  M[point] = V[point] . U[point]^T  for
// 64 ... 121 points, [rows x K] . [Co x K]^T) as ONE persistent launch.
//
// Why a kernel of its own (round 4, profiles/r04/conv_block_cycle_split.txt).  conv_igemm_kernel runs these as gm x gn x points blocks
// of one 64 x 64 tile each.  A CU holds four of them (32 KB of LDS each = 26 allocation granules of 1280 B); they are equally long,
// start together, share the matrix pipe equally and therefore stay IN STEP -- a launch takes ceil(blocks / 1024) block lives, and the
// per-block set-up (5 k cycles), the wait for the first tile from HBM (10 - 25 k) and the epilogue (8 k) are cycles in which no block of
// the CU multiplies: 0.74 of the pipe in a full round at K = 832, under 0.5 at K = 128 ... 256.
// Here a block is resident for the whole launch and works through its items (point, m-tile, n-tile) as ONE stream of K-tiles:
//   * the tile behind an item's last tile is the NEXT item's first one -- it is in flight while the last tile multiplies, so an item
//     never waits for a cold tile; the next item's addresses are scalar work beside the MFMAs;
//   * an item's epilogue is sixteen accumulator reads and sixteen buffer stores issued in front of the step's wait, which lets the
//     newest sixteen operations (the stores) stay in flight (s_waitcnt vmcnt(16)): they drain under the next item's first step.
// (Also built: the A operand in THREE stages, two tiles ahead -- four blocks per CU may use 40 KB each, 32 granules -- correct, 1 - 2 %
//  slower on every launch: the wait per K-tile is not an HBM latency one more tile of distance would cover.  Git history.)
// Same tile, same LDS image, same MFMA sequence and k order as conv_igemm_kernel's LDS-DMA form (PREC 4): results are bit-identical
// (tests/test_gpu_paths.py); item l of the launch is the tile block l of the generic launch worked on (same XCD: the grid is a
// multiple of 8).
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WG_BM = 64, WG_BN = 64;
constexpr int WG_STAGE = (WG_BM + WG_BN) * 128;       // one stage: A rows then B rows, 128 B (32 k) each
constexpr int WG_LDS = 2 * WG_STAGE;                   // 32 KB: four blocks per CU
constexpr int WG_RESIDENT = 4 * 256;                   // blocks the chip holds at a time = the persistent grid
}  // namespace

__global__ __launch_bounds__(256) void wino_gemm_kernel(WinoGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, hh = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)wg_lds);

  // ---- loader: thread = (row (tid >> 3) + 32 r, LDS slot tid & 7); the slot holds source chunk slot ^ ((row >> 1) & 7) ----
  const int lrow = tid >> 3, kpos = 4 * ((tid & 7) ^ ((tid >> 4) & 7));
  const unsigned dma_dst = lds_base + (unsigned)wave * (8 * 128);          // + stage * WG_STAGE + (A: 0 | B: 64 rows) + r * 32 rows
  auto dma16 = [&](const i32x4& desc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory", "m0");
  };
  // ---- reader: row r32 of the wave's slab, chunk (2 g + hh) ^ swizzle(row) ----
  const int sw = hh ^ ((r32 >> 1) & 7);
  int aoff[4], boff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    aoff[g] = (wm * 32 + r32) * 128 + ((sw ^ (2 * g)) << 4);
    boff[g] = WG_BM * 128 + (wn * 32 + r32) * 128 + ((sw ^ (2 * g)) << 4);
  }

  // ---- items ----
  const int gx = p.gm * p.gn;
  // (the group tables as scalars of their own: indexed through the argument struct they were copied to scratch memory, whose loads
  //  count in vmcnt like the DMAs this kernel counts by hand)
  const int nK0 = p.g_K[0], nK1 = p.g_K[1], nK2 = p.g_K[2], nK3 = p.g_K[3];
  const int nB0 = p.g_batch[0], nB1 = p.g_batch[1], nB2 = p.g_batch[2];
  const long long oX0 = p.g_x[0], oX1 = p.g_x[1], oX2 = p.g_x[2], oX3 = p.g_x[3];
  const long long oW0 = p.g_w[0], oW1 = p.g_w[1], oW2 = p.g_w[2], oW3 = p.g_w[3];
  const long long oY0 = p.g_y[0], oY1 = p.g_y[1], oY2 = p.g_y[2], oY3 = p.g_y[3];
  const int ngroups = p.ngroups, argM = p.M, argCo = p.Co, gn = p.gn, total_items = p.total_items;
  const float* const argx = p.x; const float* const argw = p.w; float* const argy = p.y;
  struct Item {
    i32x4 adesc, bdesc;
    int a_off[2], b_off[2];      // per-thread byte offsets of its two A rows / two B rows (k chunk included); bit 31: row past M
    int nkt;
    float* y;                    // the problem's output [M][Co]
    int m0, n0;
  };
  auto locate = [&](int l, Item& it) {
    // (integer divisions of uniform values are computed in vector registers: pinned back to scalars, the DMA's descriptor and scalar
    //  offset operands must be)
    const int prob = __builtin_amdgcn_readfirstlane(l / gx), bx = l - prob * gx;
    const int xcd = bx & 7, q = gx >> 3, r = gx & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bx >> 3);      // as conv_igemm_kernel: n-tile fastest
    const int mt = __builtin_amdgcn_readfirstlane(lid / gn), nt = lid - mt * gn;
    int b = prob, K = nK0;
    long long ox = oX0, ow = oW0, oy = oY0;
    if (ngroups > 1 && prob >= nB0) { b = prob - nB0; K = nK1; ox = oX1; ow = oW1; oy = oY1; }
    if (ngroups > 2 && prob >= nB0 + nB1) { b = prob - nB0 - nB1; K = nK2; ox = oX2; ow = oW2; oy = oY2; }
    if (ngroups > 3 && prob >= nB0 + nB1 + nB2) { b = prob - nB0 - nB1 - nB2; K = nK3; ox = oX3; ow = oW3; oy = oY3; }
    const float* xa = argx + ox + (long long)b * argM * K;
    const float* wa = argw + ow + (long long)b * argCo * K;
    it.y = argy + oy + (long long)b * argM * argCo;
    const unsigned long long xu = reinterpret_cast<unsigned long long>(xa), wu = reinterpret_cast<unsigned long long>(wa);
    auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    it.adesc = i32x4{sc((int)(unsigned)xu), sc((int)(unsigned)(xu >> 32) & 0xffff), sc(argM * K * 4), 0x00020000};
    it.bdesc = i32x4{sc((int)(unsigned)wu), sc((int)(unsigned)(wu >> 32) & 0xffff), sc(argCo * K * 4), 0x00020000};
    it.m0 = mt * WG_BM; it.n0 = nt * WG_BN; it.nkt = sc(K >> 5);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = it.m0 + lrow + 32 * r;
      it.a_off[r] = m < argM ? (m * K + kpos) * 4 : (int)0x80000000;
      it.b_off[r] = ((it.n0 + lrow + 32 * r) * K + kpos) * 4;
    }
  };
  // piece i (0, 1: A rows; 2, 3: B rows) of K-tile t of item `it` into stage st
  auto dma_piece = [&](const Item& it, const int i, int t, const int st) {
    const unsigned dst = dma_dst + st * WG_STAGE + (i < 2 ? i * 32 * 128 : WG_BM * 128 + (i - 2) * 32 * 128);
    const int soff = __builtin_amdgcn_readfirstlane(t * 128);
#ifdef OFFK_WG_ABLATE      /* timing experiments only: 1 = no A pieces, 2 = no B pieces, 3 = neither (the offsets point past the descriptors) */
    if (i < 2) dma16(it.adesc, dst, (OFFK_WG_ABLATE & 1) ? (int)0x80000000 : it.a_off[i], soff);
    else dma16(it.bdesc, dst, (OFFK_WG_ABLATE & 2) ? (int)0x80000000 : it.b_off[i - 2], soff);
#else
    if (i < 2) dma16(it.adesc, dst, it.a_off[i], soff);
    else dma16(it.bdesc, dst, it.b_off[i - 2], soff);
#endif
  };

  f32x16 acc;
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    asm volatile("" : "+a"(acc));
  };
  // one K-tile out of stage st; `issue`: the four pieces of the next tile (item ld, tile ld_t) go out one behind each k-group's MFMAs
  auto mma = [&](const int st, const bool issue, const Item& ld, int ld_t) {
    const char* const base = wg_lds + st * WG_STAGE;
    float4 a[2], b[2];
    a[0] = *reinterpret_cast<const float4*>(base + aoff[0]);
    b[0] = *reinterpret_cast<const float4*>(base + boff[0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) {
        a[(g + 1) & 1] = *reinterpret_cast<const float4*>(base + aoff[g + 1]);
        b[(g + 1) & 1] = *reinterpret_cast<const float4*>(base + boff[g + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (issue && g < 2) {        // an A and a B piece of the next tile in front of each of the first two k-groups (behind all four: the
        dma_piece(ld, g, ld_t, st ^ 1);          // last piece went out right in front of the wait for it; -1 .. -3 % per launch)
        dma_piece(ld, g + 2, ld_t, st ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1].x, b[g & 1].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1].y, b[g & 1].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1].z, b[g & 1].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1].w, b[g & 1].w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+a"(acc));      // the tile lives in accumulator registers across the loop (hipcc parked it in vector registers at
                                       // the back edge: sixteen v_accvgpr_write in front of every K-tile's MFMAs, sixteen reads behind)
  };
  // the finished tile: rows past M fall outside the descriptor (the row goes into the per-lane offset: only that one is compared with
  // the descriptor's size).  Exactly sixteen store instructions per wave -- the wait behind them is counted.
  auto store_tile = [&](const Item& it) {
    const unsigned long long yu = reinterpret_cast<unsigned long long>(it.y);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(yu), 0, argM * argCo * 4, 0x00020000);
    const int co4 = argCo * 4, mb = it.m0 + wm * 32;
    const int voff = (4 * hh * argCo + it.n0 + wn * 32 + r32) * 4;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[reg]), yrs, voff + (mb + (reg & 3) + 8 * (reg >> 2)) * co4, 0, 0);
  };

  int l = (int)blockIdx.x;                 // < total_items (the grid is min(total_items, resident))
  Item cp, ld;                             // the item being multiplied / the item the load unit is working on (the same, or the next)
  locate(l, ld);
  cp = ld;
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(ld, i, 0, 0);
  int ld_t = 1, cp_t = 0;                  // next tile to load of ld / tile being multiplied of cp
  bool more = true;                        // the load unit still has a tile to fetch
  bool ld_next = false;                    // ld has moved on to the item behind cp
  zero();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one step = one K-tile of cp out of stage ST; returns false behind the block's last tile
  auto step = [&](const int ST) -> bool {
    if (more && ld_t == ld.nkt) {          // the tile behind cp's last one is the first tile of the block's next item
      l += (int)gridDim.x;
      if (l < total_items) { locate(l, ld); ld_t = 0; ld_next = true; } else more = false;
    }
    mma(ST, more, ld, ld_t);
    if (more) ++ld_t;
    ++cp_t;
    const bool item_done = cp_t == cp.nkt;
    if (item_done) {
      store_tile(cp);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // the next tile has landed; the sixteen stores stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (item_done) {
      if (!ld_next) return false;          // cp was the block's last item
      cp.y = ld.y; cp.m0 = ld.m0; cp.n0 = ld.n0; cp.nkt = ld.nkt; cp_t = 0; ld_next = false;
      zero();
    }
    return true;
  };
  for (;;) {
    if (!step(0)) break;
    if (!step(1)) break;
  }
}

// shapes the persistent kernel takes (everything else goes to the generic kernel -- wino_gemms_launch in offk_api.hip); a HIP error
// of the launch itself is never read as "unsupported" 
bool wino_gemm_supported(const WinoGemmArgs& a) {
  if (a.M <= 0 || a.Co % WG_BN || a.ngroups < 1 || a.ngroups > 4) return false;
  long long problems = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    if (a.g_K[g] % 32 || a.g_K[g] <= 0 || (long long)a.M * a.g_K[g] * 4 >= 0x7fffff00ll || (long long)a.Co * a.g_K[g] * 4 >= 0x7fffff00ll) return false;
    problems += a.g_batch[g];
  }
  if ((long long)a.M * a.Co * 4 >= 0x7f000000ll) return false;
  const long long total = problems * ((a.M + WG_BM - 1) / WG_BM) * (a.Co / WG_BN);
  return total > 0 && total < (1ll << 30);
}

hipError_t wino_gemm_launch(const WinoGemmArgs& a_in, hipStream_t st) {
  WinoGemmArgs a = a_in;
  if (!wino_gemm_supported(a)) return hipErrorInvalidValue;
  int problems = 0;
  for (int g = 0; g < a.ngroups; ++g) problems += a.g_batch[g];
  a.gm = (a.M + WG_BM - 1) / WG_BM;
  a.gn = a.Co / WG_BN;
  a.total_items = (int)((long long)problems * a.gm * a.gn);
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(wino_gemm_kernel), WG_LDS);
  if (e != hipSuccess) return e;
  const int grid = a.total_items < WG_RESIDENT ? a.total_items : WG_RESIDENT;
  hipLaunchKernelGGL(wino_gemm_kernel, dim3(grid), dim3(256), WG_LDS, st, a);
  return hipGetLastError();
}

}  // namespace offk
