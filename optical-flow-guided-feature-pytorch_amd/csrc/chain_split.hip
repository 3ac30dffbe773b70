// K4cs: one bottleneck chain of fusion@28 in ONE kernel, in split-fp32 arithmetic on the bf16 matrix pipe (round 6).
//
// The chains of chain_fused.hip (reference RGB_OFF.py:670-676, :679-685: t1 = relu(c1(x)); t2 = relu(c2_3x3(t1)); y = relu(c3(t2) + x))
// with every contraction in the arithmetic of pw_tdiff_split.hip / wino_gemm_split.hip: fp32 operands as three bf16 planes (h + m + l = the
// value, exactly), the six plane products above 2^-24 of the leading one on v_mfma_f32_16x16x32_bf16 into two running fp32 accumulators per
// tile (w_h x_h | the five small products), added once.  The fp32 kernel is latency-bound (sixteen barrier-separated phases, the 3x3 in
// Winograd F(2x2, 3x3) form to halve its 32-cycle fp32 MFMAs: 0.47 of the fp32 pipe on executed FLOPs); here an MFMA is 16 cycles for eight
// times the products, so the 3x3 runs DIRECT (no transforms, no extra barriers) and the block has four barriers in all.
//
// Block = half an image (7 of the 14 rows), 256 threads, 48 KB of LDS (three blocks per CU), as chain_fused.hip; an image row is one 16-wide
// MFMA tile (slots 0 and 15 = the zero padding of the 3x3).  Weights = A operand: the library's plane images [K-tile][channel tile][plane][lane]
// x 16 B (wino_pack_split_launch) straight from L2 into registers; activations = B operand from plane images in LDS:
//   phase 1  c1 (1x1, Cin -> 64) for 8 rows (one halo row): thread (16-byte chunk, pixel) loads 4 x 16 B of x per K-step (eight lanes = one
//            pixel's 128-byte line), cuts them in registers and writes 8 B per plane into the K-step's plane image [row][plane][k group]
//            [slot ^ 2 g] (two stages);
//            t1 = relu(. + b1) is cut in the epilogue and written as planes [plane][k group 0..7][128 slots] x 16 B, slot index
//            15 row + slot + 1 (a row's slot 15 IS the next row's slot 0: both are zero padding), 2 KB per k group: the sixteen-lane
//            groups of a ds_read_b128 cover one 256-byte bank row for all three tap columns
//   phase 2  c2 (3x3, 64 -> 64) direct: 18 K-steps (tap, channel half), per step the seven output rows with the tap's shifted t1 tiles;
//            the one (row, dy) pair that reads outside the image is masked to zero; t2 = relu(. + b2) -> planes [row][K-step][plane][k group][slot]
//   phase 3  c3 (1x1, 64 -> 256) + bias + residual + ReLU -> global, one channel tile of 16 at a time
// Chain 28a (RGB_OFF.py:658-667: sa = relu(c3(t2) + branch(x0)), c1 reads relu(x0), the branch the PRE-ReLU x0; BR = true): phase 1 cuts x0
// itself into planes -- both K-steps of its 64 channels stay in LDS -- and c1 applies the ReLU to the operand as it reads it (a negative
// value's three planes are all <= 0: v_pk_max_i16 with 0 on the packed bf16 words IS relu on all three); the branch 1x1 (64 -> 256) runs
// off the same planes in front of c1, as a pass of the phase-3 item loop that stores branch(x0) + b_branch into y, where phase 3 finds it as
// its residual (every lane reads back exactly the 16 bytes it stored).
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// relu on a packed-bf16 operand whose values' planes share their sign (truncating cut): per 16-bit word max(x, 0) as a signed integer.
// (inline asm: hipcc 7.2 compiled the same thing written with __builtin_elementwise_max on the four bit-cast short2 words to ONE
// v_pk_max_i16 of the first word, selected into all four -- found by tools/debug_chain_br.py)
__device__ __forceinline__ u32x4 relu_planes(const u32x4& v) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v[e]));
    o[e] = r;
  }
  return o;
}

constexpr int CS_XTILE = 3 * 1024;             // phase 1: one image row (16 slots) of a K-step: planes h, m, l x [4 k groups][16 slots] x 16 B
constexpr int CS_XSTAGE = 8 * CS_XTILE;        // 8 rows: 24 KB
constexpr int CS_KG = 2048;                    // t1: one k group (8 channels) of a plane: 128 slots x 16 B (123 used)
constexpr int CS_T1PLANE = 8 * CS_KG;          // 64 channels
constexpr int CS_T2ROW = 2 * CS_XTILE;         // phase 3: one image row of t2: 2 K-steps x 3 planes x 1 KB
constexpr int CS_LDS = 2 * CS_XSTAGE;          // 49152 B = 3 * CS_T1PLANE >= 7 * CS_T2ROW: three blocks per CU
constexpr int CS_OOB = (int)0x80000000;

__device__ __forceinline__ f32x4 mfs(f32x4 c, const u32x4& a, const u32x4& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one (weight planes, activation planes) pair into the tile's two accumulators; planes 0 = h, 1 = m, 2 = l
__device__ __forceinline__ void mul6(f32x4& a1, f32x4& a2, const u32x4 (&w)[3], const u32x4 (&x)[3]) {
  a2 = mfs(a2, w[2], x[0]);
  a2 = mfs(a2, w[0], x[2]);
  a1 = mfs(a1, w[0], x[0]);
  a2 = mfs(a2, w[1], x[1]);
  a2 = mfs(a2, w[1], x[0]);
  a2 = mfs(a2, w[0], x[1]);
}
// two tiles that share the weight planes, their chains interleaved
__device__ __forceinline__ void mul6x2(f32x4& a1, f32x4& a2, f32x4& b1, f32x4& b2, const u32x4 (&w)[3], const u32x4 (&x)[3], const u32x4 (&y)[3]) {
  a2 = mfs(a2, w[2], x[0]); b2 = mfs(b2, w[2], y[0]);
  a2 = mfs(a2, w[0], x[2]); b2 = mfs(b2, w[0], y[2]);
  a1 = mfs(a1, w[0], x[0]); b1 = mfs(b1, w[0], y[0]);
  a2 = mfs(a2, w[1], x[1]); b2 = mfs(b2, w[1], y[1]);
  a2 = mfs(a2, w[1], x[0]); b2 = mfs(b2, w[1], y[0]);
  a2 = mfs(a2, w[0], x[1]); b2 = mfs(b2, w[0], y[1]);
}
// two independent (weights, activations) pairs, their chains interleaved
__device__ __forceinline__ void mul6ab(f32x4& a1, f32x4& a2, f32x4& b1, f32x4& b2, const u32x4 (&w)[3], const u32x4 (&x)[3], const u32x4 (&u)[3],
                                       const u32x4 (&y)[3]) {
  a2 = mfs(a2, w[2], x[0]); b2 = mfs(b2, u[2], y[0]);
  a2 = mfs(a2, w[0], x[2]); b2 = mfs(b2, u[0], y[2]);
  a1 = mfs(a1, w[0], x[0]); b1 = mfs(b1, u[0], y[0]);
  a2 = mfs(a2, w[1], x[1]); b2 = mfs(b2, u[1], y[1]);
  a2 = mfs(a2, w[1], x[0]); b2 = mfs(b2, u[1], y[0]);
  a2 = mfs(a2, w[0], x[1]); b2 = mfs(b2, u[0], y[1]);
}
// four fp32 values -> the three planes' 8 bytes each (truncating cut: h = upper 16 bits, exact remainder, again)
__device__ __forceinline__ void cut4(const f32x4& v, u32x2& ph, u32x2& pm, u32x2& pl) {
  unsigned h[4], m[4];
  float l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = __float_as_uint(v[e]) & 0xffff0000u;
    const float r = v[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r) & 0xffff0000u;
    l[e] = r - __uint_as_float(m[e]);
  }
  ph = u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302), __builtin_amdgcn_perm(h[3], h[2], 0x07060302)};
  pm = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302), __builtin_amdgcn_perm(m[3], m[2], 0x07060302)};
  pl = u32x2{__builtin_amdgcn_perm(__float_as_uint(l[1]), __float_as_uint(l[0]), 0x07060302),
             __builtin_amdgcn_perm(__float_as_uint(l[3]), __float_as_uint(l[2]), 0x07060302)};
}
}  // namespace

// NK1 = Cin / 32 (2 or 8); BR: chain 28a (NK1 = 2) with its branch conv inside
template <int NK1, bool BR>
__global__ __launch_bounds__(256, 3) void chain14_split_kernel(ChainArgs a) {
  static_assert(!BR || NK1 == 2, "the branch form keeps the whole chain input in the two plane stages");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  // blocks b and b + 8 share an XCD (round-robin placement): the two halves of an image, which read the same halo rows
  const int grp = blockIdx.x >> 4, within = blockIdx.x & 15;
  const int img = grp * 8 + (within & 7), hf = within >> 3;
  if (img >= a.n_img) return;
  const int R0 = 7 * hf;                  // first output row of this block
  const int Rf = hf ? 6 : 0;              // first of the 8 image rows whose t1 is computed (t1 row j = image row Rf + j)
  const int wlane = lane * 16;
  const bool pad = li == 0 || li == 15;

  f32x4 acc1[8], acc2[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // ---- the 1x1 conv 64 -> 256 as 28 items (channel tile n of the wave's four, row r), software-pipelined: the residual of item i + 3 and
  //      the activation tiles of item i + 1 are requested behind the MFMAs of item i, the epilogue of item i - 1 (its accumulators have
  //      landed by then) behind those.  rd(x, r): the two K-steps' planes of row r.  Output channels 64 wave + 16 n + 4 kq .. + 3. ----
  const bool px_ok = li >= 1 && li <= 14;
  const int pix0 = img * 196 + R0 * 14 + (px_ok ? li - 1 : 0);                // + row * 14
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + a.y_coff, 0, (int)(((size_t)a.n_img * 196 * a.y_cs - a.y_coff) * 4), 0x00020000);
  auto item_loop = [&](const __amdgpu_buffer_rsrc_t wrs, auto rd, const float* bias, const float* res, const int res_cs, const int res_coff,
                       const bool res_is_y, const bool relu) {
    u32x4 w3[2][2][3];                    // [set][K-step][plane]
    auto load_w3 = [&](const int set, const int n) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          w3[set][s][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, ((s * 16 + 4 * wave + n) * 3 + pl) * 1024, 0));
    };
    auto res_load = [&](const int i) {
      const int n = i / 7, r = i % 7;
      const int ch = 64 * wave + 16 * n + 4 * kq;
#if defined(OFFK_CS_EXP) && OFFK_CS_EXP == 3
      if (res_is_y) return px_ok ? *reinterpret_cast<const volatile f32x4*>(a.y + (size_t)(pix0 + r * 14) * a.y_cs + a.y_coff + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      if (res_is_y)       // what this lane stored there in the branch pass (glc: past the L1)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, px_ok ? ((pix0 + r * 14) * a.y_cs + ch) * 4 : CS_OOB, 0, 1));
      return res ? *reinterpret_cast<const f32x4*>(res + (size_t)(pix0 + r * 14) * res_cs + res_coff + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    u32x4 xb[2][2][3];                    // [item parity][K-step][plane]
    f32x4 rv[4], bb[2], pc[4];
    auto finish = [&](const int i) {      // item i: pc = its four accumulators
      const int n = i / 7, r = i % 7;
      const int ch = 64 * wave + 16 * n + 4 * kq;
      f32x4 v = ((pc[0] + pc[2]) + (pc[1] + pc[3])) + bb[n & 1] + rv[i & 3];
      if (relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      if (px_ok) *reinterpret_cast<f32x4*>(a.y + (size_t)(pix0 + r * 14) * a.y_cs + a.y_coff + ch) = v;
    };
    load_w3(0, 0);
    bb[0] = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 4 * kq);
    rv[0] = res_load(0); rv[1] = res_load(1); rv[2] = res_load(2);
    rd(xb[0], 0);
#pragma unroll
    for (int i = 0; i < 28; ++i) {
      const int n = i / 7, r = i % 7;
      if (r == 0 && n + 1 < 4) load_w3((n + 1) & 1, n + 1);
      // (the next tile's bias one item later: finish(i - 1) of the tile before still reads its own at r == 0)
      if (r == 1 && n + 1 < 4) bb[(n + 1) & 1] = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 16 * (n + 1) + 4 * kq);
      if (i + 1 < 28) rd(xb[(i + 1) & 1], (i + 1) % 7);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f};
      mul6ab(c1, c2, d1, d2, w3[n & 1][0], xb[i & 1][0], w3[n & 1][1], xb[i & 1][1]);      // the two K-steps as two interleaved chains
      if (i > 0) finish(i - 1);
      if (i + 3 < 28) rv[(i + 3) & 3] = res_load(i + 3);
      __builtin_amdgcn_sched_barrier(0);
      pc[0] = c1; pc[1] = c2; pc[2] = d1; pc[3] = d2;
    }
    finish(27);
  };

  // ================= phase 1: t1 = relu(W1 . x + b1) for 8 rows x 16 slots =================
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + a.x_coff), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w1p), 0, 64 * NK1 * 32 * 6, 0x00020000);
    // loader: thread = (16-byte chunk lc of the K-step's 128 B, pixel slot tid >> 3 (+ 32 q)): eight lanes fetch one pixel's whole 128-byte
    // line, an instruction eight lines (one lane per pixel and 64 B of it, the first form, touched 64 lines per instruction: the vector-memory
    // path was as loaded as the matrix pipe); padding slots read zeros.  Plane image of a row: [plane][k group g][position = slot ^ 2 g] x 16 B
    // (wino_gemm_split.hip's layout: conflict-free for the cut's ds_write_b64 and for the operand's ds_read_b128)
    const int lc = tid & 7, lkg = lc >> 1;
    int xvoff[4];
    char* xwr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sl = (tid >> 3) + 32 * q, lm = sl >> 4, ls = sl & 15, icol = ls - 1;
      xvoff[q] = (unsigned)icol < 14u ? ((img * 196 + (Rf + lm) * 14 + icol) * a.x_cs + 4 * lc) * 4 : CS_OOB;
      xwr[q] = lds + lm * CS_XTILE + lkg * 256 + ((ls ^ (2 * lkg)) << 4) + (lc & 1) * 8;            // + stage, + plane * 1024
    }
    const char* const xrd = lds + kq * 256 + ((li ^ (2 * kq)) << 4);                   // + stage, + m * CS_XTILE, + plane * 1024
    u32x4 xr2[2][4];                      // [set = K-step parity][q]: loaded two K-steps ahead of the multiply, cut one step ahead
    auto load_x = [&](const int kt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) xr2[kt & 1][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff[q], kt * 128, 0));
    };
    auto cut_x = [&](const int stage) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = __builtin_bit_cast(f32x4, xr2[stage][q]);
        if (!BR && a.relu_in) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};      // (BR: c1 applies it as it reads)
        u32x2 h2, m2, l2;
        cut4(v, h2, m2, l2);
        char* d = xwr[q] + stage * CS_XSTAGE;
        *reinterpret_cast<u32x2*>(d) = h2;
        *reinterpret_cast<u32x2*>(d + 1024) = m2;
        *reinterpret_cast<u32x2*>(d + 2048) = l2;
      }
    };
    u32x4 w1[2][3];
    auto load_w1 = [&](const int set, const int kt) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        w1[set][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w1rs, wlane, ((kt * 4 + wave) * 3 + q) * 1024, 0));
    };
    load_x(0);
    if (NK1 > 1) load_x(1);
    load_w1(0, 0);
    cut_x(0);
    if (BR) cut_x(1);
    if (NK1 > 2) load_x(2);
    __syncthreads();
    if constexpr (BR) {
      // the branch 1x1 off the planes of the PRE-ReLU chain input: y = W_b x0 + b_b, to be read back as phase 3's residual
      const __amdgpu_buffer_rsrc_t wbrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wbp), 0, 256 * 64 * 6, 0x00020000);
      const char* const brd = xrd + hf * CS_XTILE;        // output row r = t1 row r + hf
      item_loop(wbrs, [&](u32x4 (&x)[2][3], const int r) {
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) x[sx][pl] = *reinterpret_cast<const u32x4*>(brd + sx * CS_XSTAGE + r * CS_XTILE + pl * 1024);
      }, a.bbr, nullptr, 0, 0, false, false);
    }
#pragma unroll
    for (int kt = 0; kt < NK1; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < NK1) load_w1(st ^ 1, kt + 1);
      u32x4 xb[2][3];
      auto rdx = [&](u32x4 (&x)[3], const int m) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          x[q] = *reinterpret_cast<const u32x4*>(xrd + st * CS_XSTAGE + m * CS_XTILE + q * 1024);
          if (BR && a.relu_in) x[q] = relu_planes(x[q]);
        }
      };
      rdx(xb[0], 0);
      rdx(xb[1], 1);
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        __builtin_amdgcn_sched_barrier(0);
        mul6x2(acc1[2 * mp], acc2[2 * mp], acc1[2 * mp + 1], acc2[2 * mp + 1], w1[st], xb[0], xb[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (mp + 1 < 4) { rdx(xb[0], 2 * mp + 2); rdx(xb[1], 2 * mp + 3); }
        // the cut of the next K-step's values (loaded two steps ago) rides behind the first tile pair; its register set is re-loaded behind the second
        if (!BR && mp == 0 && kt + 1 < NK1) cut_x(st ^ 1);
        if (mp == 1 && kt + 3 < NK1) load_x(kt + 3);
      }
      __syncthreads();
    }
    // t1 -> planes: lane = (slot li, channels 16 wave + 4 kq .. + 3): k group 2 wave + (kq >> 1), its 8-byte half kq & 1
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b1 + 16 * wave + 4 * kq);
    char* const twr = lds + (2 * wave + (kq >> 1)) * CS_KG + (li + 1) * 16 + (kq & 1) * 8;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      f32x4 v = (acc1[m] + acc2[m]) + b1;
      v = pad ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      u32x2 h2, m2, l2;
      cut4(v, h2, m2, l2);
      char* d = twr + m * 240;
      *reinterpret_cast<u32x2*>(d) = h2;
      *reinterpret_cast<u32x2*>(d + CS_T1PLANE) = m2;
      *reinterpret_cast<u32x2*>(d + 2 * CS_T1PLANE) = l2;
    }
  }
  __syncthreads();
#if defined(OFFK_CS_EXP) && OFFK_CS_EXP == 1      /* timing experiment: phase 1 only */
  return;
#endif

  // ================= phase 2: t2 = relu(W2 * t1 + b2), seven output rows, direct 3x3 =================
#pragma unroll
  for (int m = 0; m < 7; ++m) { acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  {
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2p), 0, 64 * 576 * 6, 0x00020000);
    u32x4 w2[2][3];
    auto load_w2 = [&](const int set, const int q) {        // K-tile q = (channel half s = q / 9, tap q % 9): the library's packed K order
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        w2[set][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rs, wlane, ((q * 4 + wave) * 3 + pl) * 1024, 0));
    };
    // t1 row of output row r under tap row dy: j = r + dy - 1 + hf; j = -1 (first half) and j = 8 (second half) lie outside the image
    const char* const trd = lds + kq * CS_KG + li * 16;                               // + plane, + 4 s k groups, + (15 j + dx) slots
    const unsigned keep_top = hf == 0 ? 0u : 0xffffffffu, keep_bot = hf == 1 ? 0u : 0xffffffffu;
    load_w2(0, 0);
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int s = q / 9, tap = q % 9, dy = tap / 3, dx = tap % 3;
      if (q + 1 < 18) load_w2((q + 1) & 1, q + 1);
      u32x4 xb[4][3];
      auto rdt = [&](u32x4 (&x)[3], const int r) {
        const int j = min(max(r + dy - 1 + hf, 0), 7);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const u32x4*>(trd + pl * CS_T1PLANE + 4 * s * CS_KG + (15 * j + dx) * 16);
        if (r == 0 && dy == 0) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) x[pl] &= keep_top;
        }
        if (r == 6 && dy == 2) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) x[pl] &= keep_bot;
        }
      };
      rdt(xb[0], 0);
      rdt(xb[1], 1);
      // rows in pairs (two independent chains per product), the next pair's tiles read behind the current pair's MFMAs
#pragma unroll
      for (int rp = 0; rp < 4; ++rp) {
        const int b0 = 2 * (rp & 1), n0 = 2 * ((rp + 1) & 1);
        if (rp + 1 < 4) { rdt(xb[n0], 2 * rp + 2); if (2 * rp + 3 < 7) rdt(xb[n0 + 1], 2 * rp + 3); }
        __builtin_amdgcn_sched_barrier(0);
        if (rp < 3) mul6x2(acc1[2 * rp], acc2[2 * rp], acc1[2 * rp + 1], acc2[2 * rp + 1], w2[q & 1], xb[b0], xb[b0 + 1]);
        else mul6(acc1[6], acc2[6], w2[q & 1], xb[b0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                       // every wave is past its last t1 read: t2 replaces it
  {
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.b2 + 16 * wave + 4 * kq);
    char* const twr = lds + (wave >> 1) * CS_XTILE + (2 * (wave & 1) + (kq >> 1)) * 256 + li * 16 + (kq & 1) * 8;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      f32x4 v = (acc1[r] + acc2[r]) + b2;
      v = pad ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      u32x2 h2, m2, l2;
      cut4(v, h2, m2, l2);
      char* d = twr + r * CS_T2ROW;
      *reinterpret_cast<u32x2*>(d) = h2;
      *reinterpret_cast<u32x2*>(d + 1024) = m2;
      *reinterpret_cast<u32x2*>(d + 2048) = l2;
    }
  }
  __syncthreads();
#if defined(OFFK_CS_EXP) && OFFK_CS_EXP == 2      /* timing experiment: phases 1 and 2 only */
  return;
#endif

  // ================= phase 3: y = post(W3 . t2 + b3 + res) =================
  {
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w3p), 0, 256 * 64 * 6, 0x00020000);
    const char* const trd = lds + kq * 256 + li * 16;                                 // + row, + K-step * CS_XTILE, + plane * 1024
    item_loop(w3rs, [&](u32x4 (&x)[2][3], const int r) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[sx][pl] = *reinterpret_cast<const u32x4*>(trd + r * CS_T2ROW + sx * CS_XTILE + pl * 1024);
    }, a.b3, BR ? nullptr : a.res, a.res_cs, a.res_coff, BR, a.relu_out != 0);
  }
}

bool chain14_split_supported(const ChainArgs& a) {
  if (a.wbp && (a.Cin != 64 || a.res || !a.bbr)) return false;       // the branch form: chain 28a
  return a.w1p && a.w2p && a.w3p && (a.Cin == 64 || a.Cin == 256) && a.K3 == 64 && a.n_img >= 1 && a.x_bytes && !(a.x_cs % 4) && !(a.x_coff % 4) &&
         !(a.y_cs % 4) && !(a.y_coff % 4) && !(a.res && (a.res_cs % 4 || a.res_coff % 4)) && (size_t)a.n_img * 196 * a.y_cs * 4 < 0x7fffff00ull;
}

hipError_t chain14_split_launch(const ChainArgs& a, hipStream_t st, const char** why) {
  *why = nullptr;
  if (!chain14_split_supported(a)) {
    *why = "chain14 (split-fp32): need the three plane images, Cin in {64, 256}, K3 = 64 (the branch form: its plane image and bias, Cin 64, no residual), "
           "16-byte aligned channel slices, input and output below 2^31 bytes";
    return hipErrorInvalidValue;
  }
  const int blocks = (a.n_img + 7) / 8 * 16;
#define OFFK_CS_LAUNCH(NK, BRV)                                                                                \
  {                                                                                                            \
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(chain14_split_kernel<NK, BRV>), CS_LDS);        \
    if (e != hipSuccess) return e;                                                                             \
    hipLaunchKernelGGL((chain14_split_kernel<NK, BRV>), dim3(blocks), dim3(256), CS_LDS, st, a);               \
  }
  if (a.wbp) OFFK_CS_LAUNCH(2, true)
  else if (a.Cin == 64) OFFK_CS_LAUNCH(2, false)
  else OFFK_CS_LAUNCH(8, false)
#undef OFFK_CS_LAUNCH
  return hipGetLastError();
}

}  // namespace offk
