// K4c: one bottleneck chain of the fusion stage at 14x14 in ONE kernel (exact fp32, gfx950).
//
// Stands for the 1x1 -> 3x3 -> 1x1 (+ residual) blocks of fusion@28 (reference RGB_OFF.py):
//   28a :658-667   t1 = relu(c1(relu(x0)));  t2 = relu(c2_3x3(t1));  sa = relu(c3(t2) + branch(x0))   (x0 pre-ReLU, :657)
//   28b :670-676   t1 = relu(c1(sa));        t2 = relu(c2_3x3(t1));  sb = relu(c3(t2) + sa)
//   28c :679-685   same on sb -> motion_sum_28c (the 256 channels at offset 800 of the fusion@14 buffer, :760)
// As separate launches these are ten short-K convolutions at 20-55 % of the fp32 matrix peak: a launch with nothing in it
// costs 9-13 us (dispatch of ~1200 blocks, block prologue, drain in front of the dependent successor), the epilogue another
// 3-12 us, against 4-35 us of MFMA floor, and the 64-channel intermediates make an HBM / L2 round trip each
// (profiles/r02/small_conv_ablation.txt, profiles/r03/bench_b64_start_of_round.json -> roofline_in_path).
//
// Every dependency of a chain is image-local (the 3x3 never leaves the image), so a block that owns whole image rows needs
// no other block: here a block owns HALF an image (7 of the 14 rows) and computes t1 for the 8 rows it needs (one halo row
// re-computed: c1 is 12-24 % of a chain, so 1.5-3 % more MFMAs).  Halves, not whole images: P = 384 images are 1.5 per CU,
// 768 halves are exactly three resident blocks per CU (LDS 52.5 KB, <= 168 VGPRs), i.e. one full round with three waves per
// SIMD to cover each other's barriers.
//
// Geometry: an image row is one 16-wide MFMA tile -- slot 0 and slot 15 are the zero padding of the 3x3, slots 1..14 the
// pixels -- so tap (dy, dx) of output row r reads the t1 tile of row r + dy shifted by dx slots: plain LDS addresses, no
// per-tap index arithmetic.  (14 of 16 slots used: the 12.5 % the 14x14 geometry costs on 16- or 32-row MFMA tiles.)
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand: D rows = output channels, lanes = pixels, so a lane holds four
// consecutive channels of one pixel = one 16-byte store (LDS or global).  Wave w owns output channels [16w, 16w + 16) of
// c1 / c2 and [64w, 64w + 64) of c3 and reads its weight rows straight from global (L2) into MFMA operand registers; the
// activations are shared through LDS:
//   phase 1  c1 (1x1, Cin -> 64): input tiles of 16 channels by LDS-DMA into a two-stage ring (swizzled 64-byte rows), the
//            wave's whole weight slab in registers before the loop (no tracked loads beside the DMAs); t1 -> LDS, 9 rows
//            of 16 slots x 64 channels (256-byte slot rows, 16-byte chunk c stored at c ^ t1_swz(slot): the 16 lanes of a
//            ds_read_b128 group hit 16 distinct chunks for all three tap columns)
//   phase 2  c2 (3x3, 64 -> 64) for 4 (then 3) output rows: A = weights from global four steps ahead, B = shifted t1 tiles;
//            t2 -> LDS (the ring's 16 KB: 4 rows).  [r4] WINO = true (the forward's default; ChainArgs.u2): the same conv in Winograd
//            F(2x2, 3x3) form over all seven rows at once -- 512 MFMAs per wave instead of 1008, t2 written over t1 -- see there
//   phase 3  c3 (1x1, 64 -> 256; 28a: 128 -> 256 over [t2 | x0], x0 read from global) + bias + residual + ReLU -> global
// Phases 2 and 3 run twice (rows 0-3, rows 4-6) so that t2 fits the ring's 16 KB and three blocks fit a CU.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlot = 256;                              // bytes of one pixel slot of t1 / t2: 64 fp32 channels
constexpr int kT1Rows = 9;                              // 7 output rows + one halo row either side
constexpr int kT1Off = kSlot;                           // one guard slot in front (tap dx = -1 of slot 0 of the first row)
constexpr int kR2Off = kT1Off + kT1Rows * 16 * kSlot + kSlot;   // guard slot behind; then the ring / t2 region
constexpr int kR2Bytes = 16384;                         // 2 x 8 KB ring stages (128 slots x 16 channels), later 4 rows of t2
constexpr int kChainLds = kR2Off + kR2Bytes;            // 53,760 B: three blocks per CU

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float relu_i(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// t1 chunk swizzle: the 16-byte chunk j (four channels) of pixel slot sl lives at chunk position j ^ t1_swz(sl).  A ds_read_b128 is
// served in four groups of 16 lanes that are NOT contiguous -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- i.e. a group holds eight pixel slots of one k quad and the OTHER eight of its neighbour.  With the
// plain j ^ sl the centre column of a 3x3 tap is conflict-free but the columns dx = -1 / +1 (slots shifted by one) collide two-way
// in two chunks of every group: 26 % of the kernel's LDS cycles were conflict cycles (profiles/r03/pmc_per_kernel_final_fp32.csv).
// No XOR by a PERMUTATION of the slot index serves all three shifts (exhaustive over the GF(2)-linear maps, randomised over the
// rest); this table (a per-slot constant, found by search, checked against the documented lane groups for every shift, step and
// group) does: the sixteen lanes of a group read sixteen distinct chunk positions for dx = -1, 0 and +1.
__device__ __forceinline__ int t1_swz(int sl) { return (int)((0x1e2d57216aed598aull >> (4 * (sl & 15))) & 15ull); }
}  // namespace

// NKT1 = Cin / 16 (4 or 16); MERGED: c3 contracts [t2 | x] (28a: the branch conv on the pre-ReLU chain input folded in)
// WINO: phase 2 in Winograd F(2x2, 3x3) form (see there)
template <int NKT1, bool MERGED, bool WINO>
__global__ __launch_bounds__(256, 3) void chain14_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, kq = lane >> 4;
  // blocks b and b + 8 share an XCD (round-robin placement): the two halves of an image, which read the same halo rows
  const int grp = blockIdx.x >> 4, within = blockIdx.x & 15;
  const int img = grp * 8 + (within & 7), hf = within >> 3;
  if (img >= a.n_img) return;
  const int R0 = 7 * hf;                  // first output row of this block
  const int Rf = hf ? 6 : 0;              // first of the 8 image rows whose t1 is computed
  const int jf = hf ? 0 : 1;              // its t1 row index; t1 row (hf ? 8 : 0) lies outside the image: zeros
  constexpr int Cin = NKT1 * 16;
#ifdef OFFK_CHAIN_TIMING
  unsigned long long tm[4] = {0, 0, 0, 0};
  unsigned long long tm_q = __builtin_readcyclecounter();
  const unsigned long long tm_begin = tm_q;
#define OFFK_LAP(i) { const unsigned long long q_ = __builtin_readcyclecounter(); tm[i] += q_ - tm_q; tm_q = q_; }
#else
#define OFFK_LAP(i)
#endif

  // ================= phase 1: t1 = relu(W1 . x + b1) for 8 rows x 16 slots =================
  {
    // the wave's weight slab: rows 16w + li, per 16-channel step the four k of group kq
    f32x4 w1[NKT1];
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b1 + 16 * wave + 4 * kq);
    {
      const float* wr = a.w1 + (size_t)(16 * wave + li) * Cin + 4 * kq;
#pragma unroll
      for (int kt = 0; kt < NKT1; ++kt) w1[kt] = *reinterpret_cast<const f32x4*>(wr + 16 * kt);
    }
    // Input tiles of 32 channels (128 slots x 128 B = 16 KB) by LDS-DMA into a ring of up to three stages that spans the
    // whole allocation: t1 / t2 do not exist yet.  Tiles run two ahead of the MFMAs (the first version -- 16-channel tiles,
    // two stages in the t2 region, one tile ahead -- spent 105 k of a chain's 245 k cycles here for 16 k of MFMA issue).
    // Wave w lands slots 32w .. 32w + 31 (four instructions of 8 slots); position (slot, chunk cp) holds the slot's
    // k chunk cp ^ ((slot >> 1) & 7): the 16 lanes of a read group then cover 16 distinct 16-byte chunks.
    constexpr int NK = NKT1 / 2, NST = NK < 3 ? NK : 3;
    static_assert(NST * 16384 <= kChainLds, "ring does not fit");
    const unsigned long long xa = reinterpret_cast<unsigned long long>(a.x + a.x_coff);
    const i32x4 xdesc = {(int)(unsigned)xa, (int)(unsigned)(xa >> 32) & 0xffff, (int)a.x_bytes, 0x00020000};
    int voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sl = 32 * wave + 8 * q + (lane >> 3);           // slot of the tile: t1 row sl >> 4, slot sl & 15
      const int irow = Rf + (sl >> 4), icol = (sl & 15) - 1;
      const int kch = (lane & 7) ^ ((sl >> 1) & 7);
      voff[q] = (unsigned)icol < 14u ? ((img * 196 + irow * 14 + icol) * a.x_cs + 4 * kch) * 4 : (int)0x80000000;   // padding slots: zeros
    }
    const unsigned ring = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (unsigned)wave * 4096u;
    auto dma_tile = [&](const int kt, const int stage) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(ring + stage * 16384 + q * 1024), "v"(voff[q]), "s"(xdesc), "s"(kt * 128) : "memory", "m0");
    };
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* const rd0 = lds + li * 128 + ((kq ^ ((li >> 1) & 7)) << 4);
    const char* const rd1 = lds + li * 128 + (((4 + kq) ^ ((li >> 1) & 7)) << 4);
    dma_tile(0, 0);
    if (NK > 1) dma_tile(1, 1);
    if (NK > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tile 0 (and the weight slab) are there
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const int st = kt % NST;
      if (kt + 2 < NK) dma_tile(kt + 2, (kt + 2) % NST);    // every wave has left that stage at the last barrier
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        f32x4 xv[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          xv[m] = *reinterpret_cast<const f32x4*>((sp ? rd1 : rd0) + st * 16384 + m * 2048);
          if (a.relu_in) xv[m] = f32x4{relu_i(xv[m].x), relu_i(xv[m].y), relu_i(xv[m].z), relu_i(xv[m].w)};
        }
        const f32x4 w = w1[2 * kt + sp];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = mfma4(w.x, xv[m].x, acc[m]);
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = mfma4(w.y, xv[m].y, acc[m]);
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = mfma4(w.z, xv[m].z, acc[m]);
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = mfma4(w.w, xv[m].w, acc[m]);
      }
      if (kt + 2 < NK) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile kt + 1 has landed (tile kt + 2 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // t1 -> LDS: lane = (slot li, channels 16w + 4kq .. + 3); padding slots and the row outside the image are zeros
    const bool pad = li == 0 || li == 15;
    char* const wr = lds + kT1Off + li * kSlot + (((4 * wave + kq) ^ t1_swz(li)) << 4);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      f32x4 v = acc[m] + b1;
      v = pad ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      *reinterpret_cast<f32x4*>(wr + (jf + m) * 16 * kSlot) = v;
    }
    *reinterpret_cast<f32x4*>(lds + kT1Off + (hf ? 8 : 0) * 16 * kSlot + threadIdx.x * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  OFFK_LAP(0)

  // ================= phases 2 + 3, rows [r0, r0 + NR) of the block's seven =================
  // per-lane LDS offsets: t1 read for tap column dx and channel step s; t2 write / read
  int t1off[3][4], t2rd[4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int sl = li + dx - 1;                          // -1 .. 16: the guard slots take the two ends
      t1off[dx][s] = kT1Off + sl * kSlot + (((4 * s + kq) ^ t1_swz(sl)) << 4);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) t2rd[s] = kR2Off + li * kSlot + (((4 * s + kq) ^ li) << 4);
  const int t2wr = kR2Off + li * kSlot + (((4 * wave + kq) ^ li) << 4);
  const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.b2 + 16 * wave + 4 * kq);
  // c2 weight of step q = (tap, s): packed [co][ci / 32][tap][32] (pack_conv_weight_launch), 4 consecutive ci
  const float* const w2r = a.w2 + (size_t)(16 * wave + li) * 576 + 4 * kq;
  auto w2_load = [&](const int q) {
    const int tap = q >> 2, s = q & 3;
    return *reinterpret_cast<const f32x4*>(w2r + ((s >> 1) * 9 + tap) * 32 + 16 * (s & 1));
  };
  constexpr int K3 = MERGED ? 128 : 64, KS3 = K3 / 16;
  // c3 weights through a buffer descriptor: the lane's row offset in one VGPR, (n', s) as a compile-time scalar offset.  (With
  // 64-bit pointers hipcc kept one address pair per (n', s) row group -- offsets beyond the 4 KB immediate -- ran out of the 168
  // registers three blocks per CU allow in the MERGED form, spilled four of them and put an s_waitcnt vmcnt(0) of the reload in
  // front of the dependent weight load, inside the MFMA stream.)
  const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w3), 0, 256 * K3 * 4, 0x00020000);
  const int w3voff = ((64 * wave + li) * K3 + 4 * kq) * 4;                       // + 16 n' rows, + 16 s
  const bool px_ok = li >= 1 && li <= 14;
  const int pix0 = img * 196 + R0 * 14 + (px_ok ? li - 1 : 0);                // + row * 14

  auto pass = [&](const int r0, auto nr_tag) {
    constexpr int NR = decltype(nr_tag)::value;
    // ---- phase 2: t2 = relu(W2 * t1 + b2) ----
    if constexpr (!WINO) {
      f32x4 acc[NR];
#pragma unroll
      for (int m = 0; m < NR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 wq[4];
#pragma unroll
      for (int q = 0; q < 3; ++q) wq[q] = w2_load(q);
      // activations one step ahead of the MFMAs (two register sets), weights three steps ahead
      f32x4 xv[2][NR];
      auto x_load = [&](f32x4 (&x)[NR], const int q) {
        const int tap = q >> 2, sq = q & 3, dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int m = 0; m < NR; ++m) x[m] = *reinterpret_cast<const f32x4*>(lds + t1off[dx][sq] + (r0 + m + dy) * 16 * kSlot);
      };
      x_load(xv[0], 0);
#pragma unroll
      for (int q = 0; q < 36; ++q) {
        if (q + 3 < 36) wq[(q + 3) & 3] = w2_load(q + 3);
        if (q + 1 < 36) x_load(xv[(q + 1) & 1], q + 1);
        __builtin_amdgcn_sched_barrier(0);       // (hipcc otherwise hoists the weight loads of all 36 steps to the top: spills)
        const f32x4 w = wq[q & 3];
        const f32x4 (&x)[NR] = xv[q & 1];
#pragma unroll
        for (int m = 0; m < NR; ++m) acc[m] = mfma4(w.x, x[m].x, acc[m]);
#pragma unroll
        for (int m = 0; m < NR; ++m) acc[m] = mfma4(w.y, x[m].y, acc[m]);
#pragma unroll
        for (int m = 0; m < NR; ++m) acc[m] = mfma4(w.z, x[m].z, acc[m]);
#pragma unroll
        for (int m = 0; m < NR; ++m) acc[m] = mfma4(w.w, x[m].w, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        const f32x4 v = acc[m] + b2;
        *reinterpret_cast<f32x4*>(lds + t2wr + m * 16 * kSlot) = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      }
    }
    __syncthreads();
    OFFK_LAP(1)
    // ---- phase 3: y = post(W3 * [t2 | x] + b3 + res), output channels 64w + 16n' + 4kq .. + 3, two n' at a time ----
#pragma unroll
    for (int np = 0; np < 2; ++np) {
      f32x4 acc[2][NR];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < NR; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      // weights three steps ahead of their MFMAs (a step is 1024 cycles of one wave's MFMAs; one step ahead left the L2
      // latency of every load exposed: phase 3 took 77-113 k of a block's 250 k cycles for 14 k cycles of MFMA issue)
      f32x4 wq[4][2];
      auto w3_load = [&](f32x4 (&w)[2], const int s) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w3rs, w3voff, ((16 * (2 * np + n)) * K3 + 16 * s) * 4, 0);
          w[n] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
      };
      f32x4 xg[MERGED ? NR : 1];       // MERGED: the chain input (pre-ReLU) of this lane's pixels, channel step s - 4, from global
      auto xg_load = [&](const int s) {
        if constexpr (MERGED) {
#pragma unroll
          for (int m = 0; m < NR; ++m)
            xg[m] = *reinterpret_cast<const f32x4*>(a.x + (size_t)(pix0 + (r0 + m) * 14) * a.x_cs + a.x_coff + 16 * (s - 4) + 4 * kq);
        }
      };
#pragma unroll
      for (int s = 0; s < 3 && s < KS3; ++s) w3_load(wq[s], s);
      // bias and the residual rows of both n' go out in front of the MFMAs (loaded in the epilogue they cost it an L2 latency each)
      f32x4 rv[MERGED ? 1 : 2][MERGED ? 1 : NR], bb[2];      // (the merged form has its branch conv in K: no residual)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int ch = 64 * wave + 16 * (2 * np + n) + 4 * kq;
        bb[n] = *reinterpret_cast<const f32x4*>(a.b3 + ch);
        if constexpr (!MERGED) {
#pragma unroll
          for (int m = 0; m < NR; ++m)
            rv[n][m] = a.res ? *reinterpret_cast<const f32x4*>(a.res + (size_t)(pix0 + (r0 + m) * 14) * a.res_cs + a.res_coff + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if constexpr (MERGED) xg_load(4);
#pragma unroll
      for (int s = 0; s < KS3; ++s) {
        if (s + 3 < KS3) w3_load(wq[(s + 3) & 3], s + 3);
        f32x4 xv[NR];
        if (s < 4) {
#pragma unroll
          for (int m = 0; m < NR; ++m) xv[m] = *reinterpret_cast<const f32x4*>(lds + t2rd[s] + (WINO ? (kT1Off - kR2Off) + (r0 + m) * 16 * kSlot : m * 16 * kSlot));
        } else {
#pragma unroll
          for (int m = 0; m < NR; ++m) xv[m] = xg[m];
        }
        if (MERGED && s >= 4 && s + 1 < KS3) xg_load(s + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const f32x4 w = wq[s & 3][n];
#pragma unroll
          for (int m = 0; m < NR; ++m) acc[n][m] = mfma4(w.x, xv[m].x, acc[n][m]);
#pragma unroll
          for (int m = 0; m < NR; ++m) acc[n][m] = mfma4(w.y, xv[m].y, acc[n][m]);
#pragma unroll
          for (int m = 0; m < NR; ++m) acc[n][m] = mfma4(w.z, xv[m].z, acc[n][m]);
#pragma unroll
          for (int m = 0; m < NR; ++m) acc[n][m] = mfma4(w.w, xv[m].w, acc[n][m]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: one 16-byte residual load and one 16-byte store per (n', row)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int ch = 64 * wave + 16 * (2 * np + n) + 4 * kq;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
          f32x4 v = acc[n][m] + bb[n];
          if constexpr (!MERGED) v += rv[n][m];
          if (a.relu_out) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
          if (px_ok) *reinterpret_cast<f32x4*>(a.y + (size_t)(pix0 + (r0 + m) * 14) * a.y_cs + a.y_coff + ch) = v;
        }
      }
    }
    __syncthreads();        // t2 is overwritten by the next pass
    OFFK_LAP(2)
  };
  if constexpr (WINO) {
    // ---- phase 2 in Winograd F(2x2, 3x3) form, the block's seven output rows at once ----
    // 4 x 7 tiles of 2 x 2 outputs (tile n = 8 ty + tx, tx < 7; row 7 of tile row 3 is computed and dropped) = two 16-wide MFMA
    // column blocks nt = ty >> 1; sixteen points (p, q), per point a GEMM [16 channels of this wave] x [64 k] x [32 tiles]: 512
    // MFMAs per wave instead of 1008.  Two points (p, 2 h), (p, 2 h + 1) at a time -- their V, 2 x 32 tiles x 256 B, is the 16 KB
    // ring region: every thread transforms two (tile, 4-channel chunk) items of t1 (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:
    // two window rows per p), the waves multiply -- a weight quad serves both column blocks -- and the output transform
    // A^T = [1 1 1 0; 0 1 -1 -1] is folded into the accumulation (T[j] = sum_q M[q] A[q][j], Y[i][j] += A^T[i][p] T[j]): no M in
    // memory.  t2 then replaces t1 (all eight tile rows), phase 3 reads it from there.
    const int tn = threadIdx.x >> 4, tc = threadIdx.x & 15;          // transform items: tiles tn and tn + 16, channel chunk tc
    const int ttx = tn & 7;
    const bool t_ok = ttx < 7;
    auto t1_at = [&](const int ty, const int a_, const int b_) {     // window row a, slot b of tile (ty, ttx); past the region: its last row
      const int row = min(2 * ty + a_, kT1Rows - 1), sl = 2 * ttx + b_;
      return *reinterpret_cast<const f32x4*>(lds + kT1Off + (row * 16 + sl) * kSlot + ((tc ^ t1_swz(sl)) << 4));
    };
    char* const vwr = lds + kR2Off + tn * kSlot + ((tc ^ tn) << 4);                 // + point-in-pair * 8192 + (tile >= 16) * 4096
    const float* const u2r = a.u2 + (size_t)(16 * wave + li) * 64 + 4 * kq;        // + point * 4096 + 16 s
    f32x4 Y[2][2][2];                                                                // [i][j][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { Y[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; Y[i][j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        f32x4 wq[3][2], xq[2][2][2];                // weights [set][point of the pair] two channel steps ahead; V [set][point][nt] one
        auto u_load = [&](f32x4 (&w)[2], const int sq) {
#pragma unroll
          for (int e = 0; e < 2; ++e) w[e] = *reinterpret_cast<const f32x4*>(u2r + (size_t)(4 * p + 2 * hq + e) * 4096 + 16 * sq);
        };
        auto v_load = [&](f32x4 (&x)[2][2], const int sq) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) x[e][nt] = *reinterpret_cast<const f32x4*>(lds + t2rd[sq] + e * 8192 + nt * 4096);
        };
        u_load(wq[0], 0);
        u_load(wq[1], 1);
        if (t_ok) {
          constexpr int ra[4] = {0, 1, 2, 1}, rb[4] = {2, 2, 1, 3};
          constexpr float sb[4] = {-1.f, 1.f, -1.f, -1.f};
#pragma unroll
          for (int half = 0; half < 2; ++half) {          // tile rows tn >> 3 and 2 + (tn >> 3)
            const int ty = (tn >> 3) + 2 * half;
            f32x4 r[4];
#pragma unroll
            for (int b_ = 0; b_ < 4; ++b_) r[b_] = t1_at(ty, ra[p], b_) + sb[p] * t1_at(ty, rb[p], b_);
            if (hq == 0) {
              *reinterpret_cast<f32x4*>(vwr + half * 4096) = r[0] - r[2];
              *reinterpret_cast<f32x4*>(vwr + 8192 + half * 4096) = r[1] + r[2];
            } else {
              *reinterpret_cast<f32x4*>(vwr + half * 4096) = r[2] - r[1];
              *reinterpret_cast<f32x4*>(vwr + 8192 + half * 4096) = r[1] - r[3];
            }
            __builtin_amdgcn_sched_barrier(0);      // one window at a time in registers
          }
        }
        __syncthreads();
        f32x4 M[2][2];                              // [point of the pair][nt]
#pragma unroll
        for (int e = 0; e < 2; ++e) { M[e][0] = f32x4{0.f, 0.f, 0.f, 0.f}; M[e][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        v_load(xq[0], 0);
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
          if (sq + 2 < 4) u_load(wq[(sq + 2) % 3], sq + 2);
          if (sq + 1 < 4) v_load(xq[(sq + 1) & 1], sq + 1);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 (&w)[2] = wq[sq % 3];
          const f32x4 (&x)[2][2] = xq[sq & 1];
#pragma unroll
          for (int e = 0; e < 2; ++e) { M[e][0] = mfma4(w[e].x, x[e][0].x, M[e][0]); M[e][1] = mfma4(w[e].x, x[e][1].x, M[e][1]); }
#pragma unroll
          for (int e = 0; e < 2; ++e) { M[e][0] = mfma4(w[e].y, x[e][0].y, M[e][0]); M[e][1] = mfma4(w[e].y, x[e][1].y, M[e][1]); }
#pragma unroll
          for (int e = 0; e < 2; ++e) { M[e][0] = mfma4(w[e].z, x[e][0].z, M[e][0]); M[e][1] = mfma4(w[e].z, x[e][1].z, M[e][1]); }
#pragma unroll
          for (int e = 0; e < 2; ++e) { M[e][0] = mfma4(w[e].w, x[e][0].w, M[e][0]); M[e][1] = mfma4(w[e].w, x[e][1].w, M[e][1]); }
          __builtin_amdgcn_sched_barrier(0);
        }
        // (M A)[0] = M0 + M1 + M2, (M A)[1] = M1 - M2 - M3 over the row's four points q, straight into Y[i][j] += A^T[i][p] (M A)[j]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 c0 = hq == 0 ? M[0][nt] + M[1][nt] : M[0][nt];
          const f32x4 c1 = hq == 0 ? M[1][nt] : f32x4{0.f, 0.f, 0.f, 0.f} - (M[0][nt] + M[1][nt]);
          if (p < 3) { Y[0][0][nt] += c0; Y[0][1][nt] += c1; }
          if (p == 1) { Y[1][0][nt] += c0; Y[1][1][nt] += c1; }
          if (p >= 2) { Y[1][0][nt] -= c0; Y[1][1][nt] -= c1; }
        }
        asm volatile("" : "+v"(Y[0][0][0]), "+v"(Y[0][0][1]), "+v"(Y[0][1][0]), "+v"(Y[0][1][1]), "+v"(Y[1][0][0]), "+v"(Y[1][0][1]), "+v"(Y[1][1][0]), "+v"(Y[1][1][1]));
        __syncthreads();          // this pair's V is overwritten by the next one (and, behind the last, t1 by t2)
      }
    }
    // t2[row 2 ty + i][slot 1 + 2 tx + j] into the t1 region (every wave is past its last window read): lane li is tile 16 nt + li
    const int otx = li & 7;
    if (otx < 7) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = 2 * (2 * nt + (li >> 3)) + i, sl = 1 + 2 * otx + j;
            const f32x4 v = Y[i][j][nt] + b2;
            *reinterpret_cast<f32x4*>(lds + kT1Off + (row * 16 + sl) * kSlot + (((4 * wave + kq) ^ sl) << 4)) =
                f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
          }
    }
    __syncthreads();
  }
  pass(0, std::integral_constant<int, 4>());
  pass(4, std::integral_constant<int, 3>());
#ifdef OFFK_CHAIN_TIMING
  if (a.dbg && threadIdx.x == 0) {
    atomicAdd(a.dbg + 0, tm[0]); atomicAdd(a.dbg + 1, tm[1]); atomicAdd(a.dbg + 2, tm[2]);
    atomicAdd(a.dbg + 3, __builtin_readcyclecounter() - tm_begin); atomicAdd(a.dbg + 4, 1ull);
  }
#endif
#undef OFFK_LAP
}

// U[4 p + q][co][ci] = sum_ab G[p][a] G[q][b] w[co][ci][a][b], G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1] (F(2, 3), points 0, 1, -1, inf)
__global__ void chain_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (co, ci)
  if (i >= 64 * 64) return;
  const int co = i >> 6, ci = i & 63;
  float g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((co * 2 + (ci >> 5)) * 9 + t) * 32 + (ci & 31)];
  float h[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    h[0][b] = g[0][b];
    h[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
    h[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
    h[3][b] = g[2][b];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float u0 = h[p][0], u1 = 0.5f * ((h[p][0] + h[p][1]) + h[p][2]), u2 = 0.5f * ((h[p][0] - h[p][1]) + h[p][2]), u3 = h[p][2];
    U[(size_t)(4 * p + 0) * 4096 + i] = u0;
    U[(size_t)(4 * p + 1) * 4096 + i] = u1;
    U[(size_t)(4 * p + 2) * 4096 + i] = u2;
    U[(size_t)(4 * p + 3) * 4096 + i] = u3;
  }
}
hipError_t chain_wino_weight_launch(const float* w2_packed, float* U, hipStream_t st) {
  hipLaunchKernelGGL(chain_wino_weight_kernel, dim3(16), dim3(256), 0, st, w2_packed, U);
  return hipGetLastError();
}

hipError_t chain14_launch(const ChainArgs& a_in, hipStream_t st, const char** why) {
  ChainArgs a = a_in;
  *why = nullptr;
#ifdef OFFK_CHAIN_TIMING
  {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc(reinterpret_cast<void**>(&dbg), 64); (void)hipMemset(dbg, 0, 64); }
    a.dbg = dbg;
    if (getenv("OFFK_CHAIN_TIMING_DUMP")) {
      unsigned long long hb[8];
      (void)hipMemcpy(hb, dbg, 64, hipMemcpyDeviceToHost);
      if (hb[4]) fprintf(stderr, "[chain timing, thread 0 cycles per block] phase 1 %llu  phase 2 (both passes) %llu  phase 3 %llu  total %llu  (blocks %llu)\n",
                         hb[0] / hb[4], hb[1] / hb[4], hb[2] / hb[4], hb[3] / hb[4], hb[4]);
      (void)hipMemset(dbg, 0, 64);
    }
  }
#endif
  if ((a.Cin != 64 && a.Cin != 256) || (a.K3 != 64 && a.K3 != 128) || a.n_img < 1 || a.x_cs % 4 || a.x_coff % 4 || a.y_cs % 4 || a.y_coff % 4 ||
      (a.res && (a.res_cs % 4 || a.res_coff % 4)) || (a.K3 == 128 && (a.Cin != 64 || a.res))) {
    *why = "chain14: need Cin in {64, 256}, K3 in {64, 128 (with Cin 64)}, 16-byte aligned channel slices";
    return hipErrorInvalidValue;
  }
  if (!a.x_bytes) { *why = "chain14: input beyond 31-bit byte offsets"; return hipErrorInvalidValue; }
  const int blocks = (a.n_img + 7) / 8 * 16;
#define OFFK_CHAIN_LAUNCH_W(NKT, MRG, WN)                                                                             \
  {                                                                                                                   \
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(chain14_kernel<NKT, MRG, WN>), kChainLds);             \
    if (e != hipSuccess) return e;                                                                                    \
    hipLaunchKernelGGL((chain14_kernel<NKT, MRG, WN>), dim3(blocks), dim3(256), kChainLds, st, a);                    \
  }
#define OFFK_CHAIN_LAUNCH(NKT, MRG) { if (a.u2) OFFK_CHAIN_LAUNCH_W(NKT, MRG, true) else OFFK_CHAIN_LAUNCH_W(NKT, MRG, false) }
  if (a.Cin == 64 && a.K3 == 128) OFFK_CHAIN_LAUNCH(4, true)
  else if (a.Cin == 64) OFFK_CHAIN_LAUNCH(4, false)
  else OFFK_CHAIN_LAUNCH(16, false)
#undef OFFK_CHAIN_LAUNCH
#undef OFFK_CHAIN_LAUNCH_W
  return hipGetLastError();
}

}  // namespace offk
