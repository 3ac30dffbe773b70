// K4gr: the batched GEMMs of a convolution on a Winograd path (wino_gemm.hip: M[point] = V[point] . U[point]^T), fp32 matrix pipe, with
// the units kernel's operand scheme (pw_tdiff.hip: pw_tdiff16_kernel) -- VERDICT r04 #2: the weight operand U straight from L2 into
// registers, only V through LDS, 64 rows x 128 output channels per item (a wave owns all 64 rows x 32 channels: eight 16x16 accumulator
// tiles), v_mfma_f32_16x16x4_f32.  Reference arithmetic: RGB_OFF.py:762, 766, 775-777, 833, 837 in their Winograd forms.
//
//   * U: the library's operand-order image [problem][K-tile][channel tile][k half h][lane] x 16 B (wino_pack_r_kernel): lane (channel i,
//     kq) holds U[ch][32 kt + 16 h + 4 kq .. + 3]; a wave loads its two channel tiles x two halves = four 1-KB pieces of the NEXT K-tile
//     into registers;
//   * V: the K-tile [64 rows][32 k] by LDS-DMA into swizzled 128-byte rows (wino_gemm.hip's image), two stages; lane (row i, kq) reads
//     the 16-byte chunk kq + 4 h of its row -- four consecutive k -- so MFMA (h, j) contracts k = 16 h + 4 kq + j on both operands;
//   * persistent items in wino_gemm_split.hip's order (whole problems per XCD, m-tile fastest), the next item's first tile in flight
//     during the last step of the current one, the epilogue's eight 16-byte stores issued behind the step's wait (they stay in flight).
// Not bit-identical to wino_gemm_kernel (another k order inside a K-tile); the same fp32 products, sums within a few ulps.
#include <climits>
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int GR_BM = 64, GR_BN = 128;
constexpr int GR_STAGE = GR_BM * 128;        // one V K-tile: 64 rows x 128 B
constexpr int GR_LDS = 2 * GR_STAGE;         // 16 KB
#ifndef OFFK_GR_BLOCKS
#define OFFK_GR_BLOCKS 4
#endif
constexpr int GR_RESIDENT = OFFK_GR_BLOCKS * 256;
constexpr int GR_OOB = (int)0x80000000;
}  // namespace

__global__ __launch_bounds__(256, OFFK_GR_BLOCKS) void wino_gemm_r_kernel(WinoGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gr_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)gr_lds);

  // ---- V loader (wino_gemm.hip's): thread = (row (tid >> 3) + 32 r, LDS slot tid & 7); the slot holds source chunk slot ^ ((row >> 1) & 7) ----
  const int lrow = tid >> 3, kpos = 4 * ((tid & 7) ^ ((tid >> 4) & 7));
  const unsigned dma_dst = lds_base + (unsigned)wave * (8 * 128);          // + stage * GR_STAGE + r * 32 rows
  // ---- V reader: row rt * 16 + li, chunk (kq + 4 h) ^ ((row >> 1) & 7) ----
  int boff[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rt * 16 + li;
      boff[rt][h] = row * 128 + (((kq + 4 * h) ^ ((row >> 1) & 7)) << 4);
    }

  const int gx = p.gm * p.gn;
  const int nK0 = p.g_K[0], nK1 = p.g_K[1], nK2 = p.g_K[2], nK3 = p.g_K[3];
  const int nB0 = p.g_batch[0], nB1 = p.g_batch[1], nB2 = p.g_batch[2];
  const long long oX0 = p.g_x[0], oX1 = p.g_x[1], oX2 = p.g_x[2], oX3 = p.g_x[3];
  const long long oW0 = p.g_w[0], oW1 = p.g_w[1], oW2 = p.g_w[2], oW3 = p.g_w[3];
  const long long oY0 = p.g_y[0], oY1 = p.g_y[1], oY2 = p.g_y[2], oY3 = p.g_y[3];
  const int ngroups = p.ngroups, argM = p.M, argCo = p.Co, total_items = p.total_items, gm = p.gm;
  const int main_items = (total_items / gx / 8) * 8 * gx;
  const int ukstep = (argCo >> 4) * 2048;          // bytes of one K-tile of a problem's image: channel tiles x 2 KB
  const int grid = (int)gridDim.x;
  auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  i32x4 xdesc, wdesc;
  {
    const unsigned long long xa = reinterpret_cast<unsigned long long>(p.x), wa = reinterpret_cast<unsigned long long>(p.w_planes);
    xdesc = i32x4{(int)(unsigned)xa, (int)(unsigned)(xa >> 32) & 0xffff, (int)p.x_bytes, 0x00020000};
    wdesc = i32x4{(int)(unsigned)wa, (int)(unsigned)(wa >> 32) & 0xffff, (int)p.w_bytes, 0x00020000};
  }
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);

  // the load cursor (one tile ahead of the multiply): V offsets of the thread's two rows, the item's scalar offsets
  int l_l = (int)blockIdx.x, l_t = 0, l_nkt = 0, l_xsoff = 0, l_usoff = 0, l_off[2] = {GR_OOB, GR_OOB}, l_uvoff = lane * 16;
  int l_ysoff = 0, l_m0 = 0, l_n0 = 0;
  auto locate = [&](int l) {
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
    l_m0 = mt * GR_BM; l_n0 = nt * GR_BN;
    l_xsoff = sc((int)((ox + (long long)b * argM * K) * 4));
    l_nkt = sc(K >> 5);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = l_m0 + lrow + 32 * r;
      l_off[r] = m < argM ? (m * K + kpos) * 4 : GR_OOB;
    }
    l_usoff = sc((int)((ow + (long long)b * argCo * K) * 4) + ((l_n0 >> 4) + 2 * wave) * 2048);
    l_ysoff = sc((int)((oy + (long long)b * argM * argCo) * 4));
  };
  // the multiply
  int c_l = l_l, c_t = 0, c_nkt = 0, c_ysoff = 0, c_m0 = 0, c_n0 = 0;

  auto dma_v = [&](const int r, const int st) {
    const unsigned dst = dma_dst + st * GR_STAGE + r * 32 * 128;
    const int soff = __builtin_amdgcn_readfirstlane(l_xsoff + l_t * 128);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(dst), "v"(l_off[r]), "s"(xdesc), "s"(soff) : "memory", "m0");
  };
  u32x4 wr[2][2][2];               // U [set][channel tile][k half]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) { wr[a][c][0] = u32x4{0u, 0u, 0u, 0u}; wr[a][c][1] = u32x4{0u, 0u, 0u, 0u}; }
  auto load_u = [&](const int set, const int n) {      // n = ct * 2 + h
    const int soff = __builtin_amdgcn_readfirstlane(l_usoff + l_t * ukstep + n * 1024);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wr[set][n >> 1][n & 1]) : "v"(l_uvoff), "s"(wdesc), "s"(soff) : "memory");
  };
  bool more = true, l_next = false;      // the load cursor still has a tile to fetch / it has moved on to the item behind the multiply's
  auto adv_l = [&]() {
    ++l_t;
    if (l_t == l_nkt) {
      l_l += grid; l_t = 0;
      if (l_l < total_items) { locate(l_l); l_next = true; }
      else { more = false; l_off[0] = GR_OOB; l_off[1] = GR_OOB; l_uvoff = GR_OOB; l_nkt = INT_MAX; }
    }
  };

  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#define OFFK_SB __builtin_amdgcn_sched_barrier(0)
  const int st_voff = (li * argCo + 2 * wave * 16 + 4 * kq) * 4;
  auto store_item = [&]() {      // (each store followed by a fenced s_nop 1: the store-data hazard, tools/probe_store_hazard.hip)
    const int ybase = c_ysoff + (c_m0 * argCo + c_n0) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rt = i >> 1;
      const int voff = li < argM - c_m0 - rt * 16 ? st_voff : GR_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), yrs, voff, ybase + (rt * 16 * argCo + (i & 1) * 16) * 4, 0);
      OFFK_SB;
      asm volatile("s_nop 1");
      OFFK_SB;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: tile 0 of the first item in stage 0 and register set 0 ----
  locate(l_l);
  c_ysoff = l_ysoff; c_m0 = l_m0; c_n0 = l_n0; c_nkt = l_nkt;
  dma_v(0, 0); dma_v(1, 0);
#pragma unroll
  for (int n = 0; n < 4; ++n) load_u(0, n);
  adv_l();
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[0][0][0]), "+v"(wr[0][0][1]), "+v"(wr[0][1][0]), "+v"(wr[0][1][1]) :: "memory");
  __syncthreads();

  // one step = one K-tile out of stage ST / register set ST; the next tile (the load cursor's) goes into stage / set ST ^ 1
  auto step = [&](const int ST) -> bool {
    const char* const rd = gr_lds + ST * GR_STAGE;
    f32x4 xb[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) xb[rt] = *reinterpret_cast<const f32x4*>(rd + boff[rt][h]);
      OFFK_SB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rt = i >> 1, ct = i & 1;
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wr[ST][ct][h][j]), xb[rt][j], acc[i], 0, 0, 0);
          // the next tile's two DMAs and four U loads one at a time behind the first MFMAs of the step
          if (h == 0 && j == 0) {
            if (i == 0) dma_v(0, ST ^ 1);
            else if (i == 1) dma_v(1, ST ^ 1);
            else if (i < 6) load_u(ST ^ 1, i - 2);
          }
          OFFK_SB;
        }
      }
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    if (more) adv_l();
    ++c_t;
    const bool item_done = c_t == c_nkt;
    // the next tile has landed (V in LDS, U in set ST ^ 1).  ONE wait statement in front of the item's stores: they are issued behind it and
    // stay in flight across the barrier.  (Two statements, vmcnt(8) behind the stores and vmcnt(0) otherwise, made hipcc merge the two
    // paths through COPIES of the U registers placed in front of the waits -- copies of loads that had not landed.)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[ST ^ 1][0][0]), "+v"(wr[ST ^ 1][0][1]), "+v"(wr[ST ^ 1][1][0]), "+v"(wr[ST ^ 1][1][1]) :: "memory");
    if (item_done) store_item();
    __syncthreads();
    if (item_done) {
      if (!l_next) return false;          // that was the block's last item
      c_ysoff = l_ysoff; c_m0 = l_m0; c_n0 = l_n0; c_nkt = l_nkt; c_t = 0; l_next = false;
    }
    return true;
  };
  for (;;) {
    if (!step(0)) break;
    if (!step(1)) break;
  }
}

// U [problems][Co][K] fp32 -> the operand-order image: thread = (problem, K-tile, channel tile, k half, lane), one 16-byte piece
__global__ void wino_pack_r_kernel(const float* __restrict__ U, float4* __restrict__ img, int Co, int K, int nproblems) {
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nct = Co >> 4, nkt = K >> 5;
  const long long per = (long long)nkt * nct * 2 * 64;
  if (item >= per * nproblems) return;
  const int prob = (int)(item / per);
  const int rem = (int)(item - prob * per);
  const int lane = rem & 63, h = (rem >> 6) & 1, ct = (rem >> 7) % nct, kt = (rem >> 7) / nct;
  const int li = lane & 15, kq = lane >> 4;
  img[item] = *reinterpret_cast<const float4*>(U + ((size_t)prob * Co + ct * 16 + li) * K + kt * 32 + 16 * h + 4 * kq);
}
hipError_t wino_pack_r_launch(const float* U, void* img, int Co, int K, int nproblems, hipStream_t st) {
  if (Co % 16 || K % 32 || nproblems <= 0) return hipErrorInvalidValue;
  const long long items = (long long)nproblems * (K / 32) * (Co / 16) * 2 * 64;
  hipLaunchKernelGGL(wino_pack_r_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, U, reinterpret_cast<float4*>(img), Co, K, nproblems);
  return hipGetLastError();
}

static bool gr_extents(const WinoGemmArgs& a, long long* problems, long long* xe, long long* we, long long* ye) {
  *problems = 0; *xe = 0; *we = 0; *ye = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    if (a.g_K[g] % 32 || a.g_K[g] < 64 || a.g_batch[g] <= 0) return false;
    *problems += a.g_batch[g];
    *xe = std::max(*xe, a.g_x[g] + (long long)a.g_batch[g] * a.M * a.g_K[g]);
    *we = std::max(*we, a.g_w[g] + (long long)a.g_batch[g] * a.Co * a.g_K[g]);
    *ye = std::max(*ye, a.g_y[g] + (long long)a.g_batch[g] * a.M * a.Co);
    if ((long long)a.M * a.g_K[g] * 4 >= 0x7fffff00ll) return false;
  }
  return *xe * 4 < 0x7fffff00ll && *we * 4 < 0x7fffff00ll && *ye * 4 < 0x7fffff00ll;
}

// (w_planes: the image of wino_pack_r_launch, 4 bytes per element at the same element offsets as w)
bool wino_gemm_r_supported(const WinoGemmArgs& a) {
  if (!a.w_planes || a.M <= 0 || a.Co % GR_BN || a.ngroups < 1 || a.ngroups > 4 || a.epilogue) return false;
  long long problems, xe, we, ye;
  if (!gr_extents(a, &problems, &xe, &we, &ye)) return false;
  const long long total = problems * ((a.M + GR_BM - 1) / GR_BM) * (a.Co / GR_BN);
  return total > 0 && total < (1ll << 30);
}

hipError_t wino_gemm_r_launch(const WinoGemmArgs& a_in, hipStream_t st) {
  WinoGemmArgs a = a_in;
  if (!wino_gemm_r_supported(a)) return hipErrorInvalidValue;
  long long problems, xe, we, ye;
  gr_extents(a, &problems, &xe, &we, &ye);
  a.x_bytes = xe * 4; a.w_bytes = we * 4; a.y_bytes = ye * 4;
  a.gm = (a.M + GR_BM - 1) / GR_BM;
  a.gn = a.Co / GR_BN;
  a.total_items = (int)(problems * a.gm * a.gn);
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(wino_gemm_r_kernel), GR_LDS);
  if (e != hipSuccess) return e;
  const int grid = a.total_items < GR_RESIDENT ? a.total_items : GR_RESIDENT;
  hipLaunchKernelGGL(wino_gemm_r_kernel, dim3(grid), dim3(256), GR_LDS, st, a);
  return hipGetLastError();
}

}  // namespace offk
