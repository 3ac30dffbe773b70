// K4m: what sits BETWEEN two Winograd convolutions on 7x7 maps, in one launch (exact fp32).
//
// The 7x7-map stages of the fusion network alternate 3x3 convolutions (on the Winograd path: winograd.hip) with 1x1 convolutions
// (RGB_OFF.py:762-767 x1 -> c1_14a -> c2_14a, :773-780 c1_14b -> c2_14b -> c3_14b, :833-838 x2 -> c1 -> c2).  As separate launches a
// 3x3 -> 1x1 -> 3x3 run costs an output transform, the 1x1 conv and an input transform: three short launches (10-35 us each at
// P = 384, most of it launch / drain latency) and two activation round trips through HBM for 0.6 - 2.5 GFLOP of matrix work
// (profiles/r03/bench_b64_final.json, roofline_in_path).  Here a block owns ONE image and does all of it on chip:
//   stage A  x = relu(A^T M A + bias)      the output transform of the conv in front, from its GEMM output M [121][P][CIN];
//                                          x goes to LDS ([64 pixel slots][CIN], 16-byte chunks XOR-swizzled by the slot) and,
//                                          where a later conv reads it, to its channel slice in HBM (x1 / x2 of the merged convs)
//   stage B  t = relu(W1 x + b1)           the 1x1 conv: v_mfma_f32_16x16x4_f32, weights = A operand straight from L2 (a lane ends
//                                          with four consecutive channels of one pixel), pixels = B operand from LDS; t -> LDS
//   stage C  V = B^T t B                   the input transform of the conv behind, to its GEMM input V [121][P][CMID]
// SPL (round 6; split-fp32 handles): stage B in the split arithmetic of wino_gemm_split.hip -- stage A cuts every activation into its three
// bf16 planes as it stores it (2-byte stores into the plane image [pixel tile][plane][k group g][position = slot ^ 2 g] x 16 B, the layout of
// wino_gemm_split.hip), the weights come as the library's plane image, six v_mfma_f32_16x16x32_bf16 per (pixel tile, channel tile, 32 k) into two
// running accumulators: 384 16-cycle MFMAs per wave of the 256 -> 256 form instead of 512 32-cycle ones.
// GEMM = false drops stage B: output transform + ReLU + input transform between two 3x3 convs that follow each other
// (motion_conv2_trans_14b -> motion_conv3_trans_14b).
// The transforms are the arithmetic of wino_output_kernel / wino_input_kernel element for element; only the 1x1 conv's summation
// order differs from the generic kernel's (k in steps of 4 instead of 2).
#include <type_traits>

#include "offk_common.h"
#include "offk_internal.h"
#include "winograd_common.h"

namespace offk {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// byte offset of channel c of pixel slot px in a [slot][C] tile whose 16-byte chunks are XOR-swizzled by the slot: a ds_read_b128 of
// (16 slots, one k quad) and a ds_write / ds_read_b32 of 64 consecutive channels of one slot are both conflict-free
template <int C>
__device__ __forceinline__ int tile_off(int px, int c) { return px * (C * 4) + ((((c >> 2) ^ (px & 15))) << 4) + (c & 3) * 4; }

// split-fp32 stage B: the x tile as plane images.  Byte offset of channel c (of the tile's LC) of pixel slot px in plane 0; planes are
// kPlaneStride<LC> apart.  [pixel tile px >> 4][plane][k group c >> 3][position (px & 15) ^ 2 (g & 7)] x 16 B + (c & 7) x 2 B
template <int LC> constexpr int kPlaneStride = (LC / 8) * 256;
template <int LC> constexpr int kPtStride = 3 * kPlaneStride<LC>;
template <int LC>
__device__ __forceinline__ int plane_off(int px, int c) {
  const int g = c >> 3;
  return (px >> 4) * kPtStride<LC> + g * 256 + ((((px & 15) ^ (2 * (g & 7)))) << 4) + (c & 7) * 2;
}
// one activation -> its three bf16 planes (truncating cut, as everywhere), 2-byte stores
template <int LC>
__device__ __forceinline__ void store_planes(char* xt, int px, int c, float v) {
  const unsigned h = __float_as_uint(v) & 0xffff0000u;
  const float r = v - __uint_as_float(h);
  const unsigned m = __float_as_uint(r) & 0xffff0000u;
  const float l = r - __uint_as_float(m);
  char* d = xt + plane_off<LC>(px, c);
  *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h >> 16);
  *reinterpret_cast<unsigned short*>(d + kPlaneStride<LC>) = (unsigned short)(m >> 16);
  *reinterpret_cast<unsigned short*>(d + 2 * kPlaneStride<LC>) = (unsigned short)(__float_as_uint(l) >> 16);
}

// stage A for one class: M -> relu(A^T M A + bias) -> LDS tile (+ HBM), lane = channel within a group of 64.  TWO channel groups at
// a time: their 2 x NY x NX loads -- every one a 256-byte row of a different point plane -- are in flight together (one group at a
// time, the four groups of a 256-channel image were four exposed HBM latencies per block: the first version's 60 us at P = 384).
// (G0 .. G1: the channel groups of 64 this call covers; LC: channels per row of the LDS tile, which starts at channel 64 G0)
template <int NPH, int CY, int CX, int CIN, int G0 = 0, int G1 = CIN / 64, int LC = CIN, bool SPL = false>
__device__ __forceinline__ void mid_out_class(const WinoMidArgs& a, int img, int lane, char* xt, float* xg) {
  constexpr int NY = CY ? 5 : 6, NX = CX ? 5 : 6, OY = CY ? 3 : 4, OX = CX ? 3 : 4, oy = CY ? 4 : 0, ox = CX ? 4 : 0;
  const size_t pstride = (size_t)a.n_img * CIN;
#pragma unroll
  for (int g0 = G0; g0 < G1; g0 += 2) {
    float m[2][NY][NX];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* mp = a.M + (size_t)img * CIN + (g0 + u) * 64 + lane;
#pragma unroll
      for (int j = 0; j < NX; ++j)
#pragma unroll
        for (int i = 0; i < NY; ++i) m[u][i][j] = mp[(size_t)wino_mindex<NPH, CY, CX>(i, j) * pstride];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = (g0 + u) * 64 + lane;
      float s[OY][NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        float col[NY], sc[OY];
#pragma unroll
        for (int i = 0; i < NY; ++i) col[i] = m[u][i][j];
        wat(col, sc);
#pragma unroll
        for (int i = 0; i < OY; ++i) s[i][j] = sc[i];
      }
      const float bv = a.bias_in ? a.bias_in[c] : 0.f;
#pragma unroll
      for (int i = 0; i < OY; ++i) {
        float yv[OX];
        wat(s[i], yv);
#pragma unroll
        for (int j = 0; j < OX; ++j) {
          const int px = (oy + i) * 7 + ox + j;
          const float v = fmaxf(yv[j] + bv, 0.f);
          if constexpr (SPL) store_planes<LC>(xt, px, c - 64 * G0, v);
          else *reinterpret_cast<float*>(xt + tile_off<LC>(px, c - 64 * G0)) = v;
          if (xg) xg[((size_t)img * 49 + px) * a.x_cs + a.x_coff + c] = v;
        }
      }
    }
  }
}

// stage C for one class: LDS tile -> B^T t B -> V, lane = channel within a group of 64 of the block's C channels (V has CV)
template <int CY, int CX, int C, int CV>
__device__ __forceinline__ void mid_in_class(const WinoMidArgs& a, int img, int lane, int c_first, const char* ot) {
  constexpr int NY = CY ? 5 : 6, NX = CX ? 5 : 6;
  constexpr int oy = CY ? 4 : 0, ox = CX ? 4 : 0;          // output origin; the window starts one sample up / left of it
  const size_t pstride = (size_t)a.n_img * CV;
#pragma unroll
  for (int g = 0; g < C / 64; ++g) {
    const int c = g * 64 + lane;
    float t[NY][NX];
#pragma unroll
    for (int j = 0; j < NX; ++j)
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int row = oy - 1 + i, col = ox - 1 + j;          // (compile-time: which samples are the zero padding)
        t[i][j] = row >= 0 && row < 7 && col >= 0 && col < 7 ? *reinterpret_cast<const float*>(ot + tile_off<C>(row * 7 + col, c)) : 0.f;
      }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      float d[NY], tc[NY];
#pragma unroll
      for (int i = 0; i < NY; ++i) d[i] = t[i][j];
      wbt(d, tc);
#pragma unroll
      for (int i = 0; i < NY; ++i) t[i][j] = tc[i];
    }
    float* vp = a.V + (size_t)img * CV + c_first + c;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      float v[NX];
      wbt(t[i], v);
#pragma unroll
      for (int j = 0; j < NX; ++j) vp[(size_t)wino_mindex<1, CY, CX>(i, j) * pstride] = v[j];
    }
  }
}
}  // namespace

// NPH: point order of M (1: a 3x3 / stride 1 conv in front, 4: the polyphase 5x5 / stride 2 conv).  NS: the 1x1 conv's output
// channels are split over NS blocks per image (blockIdx.y), each of which repeats stage A for all CIN channels (M comes out of L2 the
// second time): P = 384 images alone are 1.5 blocks per CU, and a 256 -> 256 block is 33 k cycles of MFMAs per wave.
// KH = 2 (CIN = 256, NS = 2): the x tile in two k halves of 128 channels -- stage A and stage B alternate per half, the
// accumulators stay in registers -- so that x, and t behind it, take 32 KB instead of 64 KB and three blocks fit a CU.  Same k order:
// bit-identical to KH = 1.
template <int CIN, int CMID, bool GEMM, int NPH, int NS, int KH = 1, bool SPL = false>
__global__ __launch_bounds__(256, (KH == 2 || SPL) ? 3 : 2) void wino_mid_kernel(WinoMidArgs a) {
  static_assert(!SPL || (GEMM && CIN / KH == 128), "split stage B: x tiles of 128 channels (48 KB of planes: three blocks per CU)");
  // (CIN % 128: mid_out_class walks the channel groups of 64 two at a time)
  static_assert(CIN % 128 == 0 && CMID % (64 * NS) == 0 && (GEMM || (CIN == CMID && NS == 1)), "whole waves of channels");
  static_assert(KH == 1 || (KH == 2 && GEMM && CIN == 256), "k halves: the 256 -> 256 form");
  constexpr int CM = CMID / NS;                                   // output channels of this block
  constexpr int XC = CIN / KH;                                    // channels per row of the x tile
  extern __shared__ __attribute__((aligned(16))) char lds[];      // [64 slots][max(XC, CM)] fp32: x (a k half of it), then t in the same place
  const int img = blockIdx.x;
  const int c_first = (int)blockIdx.y * CM;                       // first output channel of this block
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- stage A: wave w = class w (its 6x6 / 6x5 / 5x6 / 5x5 points), every channel group; the activation goes to HBM from the first
  //      of the NS blocks ----
  float* const xg = blockIdx.y == 0 ? a.x : nullptr;
  auto zero_tail = [&]() {
    // pixel slots 49 .. 63 of the x tile: stage A never writes them and stage B multiplies them (MFMA columns that are dropped behind
    // it) -- zeroed so that no uninitialised LDS word is ever read
    if constexpr (SPL) {      // the whole fourth pixel tile of the plane image (pixel 48 is written behind the barrier that follows)
      for (int i = tid; i < kPtStride<XC> / 16; i += 256) reinterpret_cast<f32x4*>(lds + 3 * kPtStride<XC>)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      __syncthreads();
    } else {
      for (int i = tid; i < 15 * XC / 4; i += 256) reinterpret_cast<f32x4*>(lds + 49 * (XC * 4))[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage_a = [&](auto half) {
    constexpr int H = decltype(half)::value;
    constexpr int G0 = KH == 2 ? 2 * H : 0, G1 = KH == 2 ? 2 * H + 2 : CIN / 64;
    switch (wave) {
      case 0: mid_out_class<NPH, 0, 0, CIN, G0, G1, XC, SPL>(a, img, lane, lds, xg); break;
      case 1: mid_out_class<NPH, 0, 1, CIN, G0, G1, XC, SPL>(a, img, lane, lds, xg); break;
      case 2: mid_out_class<NPH, 1, 0, CIN, G0, G1, XC, SPL>(a, img, lane, lds, xg); break;
      default: mid_out_class<NPH, 1, 1, CIN, G0, G1, XC, SPL>(a, img, lane, lds, xg); break;
    }
  };
  if constexpr (GEMM) zero_tail();
  stage_a(std::integral_constant<int, 0>{});
  __syncthreads();

  if constexpr (GEMM) {
    // ---- stage B: t[ch][px] = relu(sum_k W1[ch][k] x[px][k] + b1[ch]); wave w owns the block's channels [w CM / 4, (w + 1) CM / 4) ----
    constexpr int CT = CM / 64, PT = 4, KS = CIN / 16, KSH = KS / KH;  // channel tiles per wave, pixel tiles (49 of 64 slots used), k steps (per half)
    const int li = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, CMID * CIN * 4, 0x00020000);
    const int wvoff = ((c_first + wave * (CM / 4) + li) * CIN + 4 * kq) * 4;
    auto w_load = [&](f32x4 (&w)[CT], const int s) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (16 * ct * CIN + 16 * s) * 4, 0);
        w[ct] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    const char* const xrd = lds + li * (XC * 4);                    // + pt * 16 rows, + chunk (4 s + kq) ^ li   (s within the half)
    auto x_load = [&](f32x4 (&x)[PT], const int s) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        x[pt] = *reinterpret_cast<const f32x4*>(xrd + pt * 16 * (XC * 4) + (((4 * s + kq) ^ li) << 4));
    };
    f32x4 acc[PT][CT], acs[SPL ? PT : 1][SPL ? CT : 1];      // (split: acc = sum w_h x_h, acs = the five small products)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (SPL) acs[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // split-fp32: the K-steps of 32 channels [t0, t0 + XC / 32) of the conv out of the plane image in LDS (which holds the channels from 32 t0 on)
    auto stage_b_split = [&](const int t0) {
      if constexpr (SPL) {
        constexpr int NT = XC / 32, NCT = CMID / 16;
        const __amdgpu_buffer_rsrc_t wps = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w1p), 0, CMID * CIN * 6, 0x00020000);
        const int ct0 = (c_first + wave * (CM / 4)) >> 4;
        const char* const prd = lds + kq * 256;          // + pt, + plane, + 4 t k groups; position li ^ 2 (g & 7)
        auto mfs = [&](f32x4 c, const u32x4& w, const u32x4& x) {
          return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
        };
        u32x4 wq[2][CT][3];
        auto wl = [&](const int set, const int t) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              wq[set][ct][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wps, lane * 16, (((t0 + t) * NCT + ct0 + ct) * 3 + pl) * 1024, 0));
        };
        u32x4 xq[2][3];
        auto xl = [&](u32x4 (&x)[3], const int t, const int pt) {
          const int g = 4 * t + kq;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            x[pl] = *reinterpret_cast<const u32x4*>(prd + pt * kPtStride<XC> + pl * kPlaneStride<XC> + 4 * t * 256 + ((li ^ (2 * (g & 7))) << 4));
        };
        wl(0, 0);
        xl(xq[0], 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t + 1 < NT) wl((t + 1) & 1, t + 1);
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const int nx = t * PT + pt + 1;
            if (nx < NT * PT) xl(xq[nx & 1], nx / PT, nx % PT);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&x)[3] = xq[(t * PT + pt) & 1];
            // planes 0 = h, 1 = m, 2 = l; the channel tiles' chains interleaved
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acs[pt][ct] = mfs(acs[pt][ct], wq[t & 1][ct][2], x[0]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acs[pt][ct] = mfs(acs[pt][ct], wq[t & 1][ct][0], x[2]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = mfs(acc[pt][ct], wq[t & 1][ct][0], x[0]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acs[pt][ct] = mfs(acs[pt][ct], wq[t & 1][ct][1], x[1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acs[pt][ct] = mfs(acs[pt][ct], wq[t & 1][ct][1], x[0]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acs[pt][ct] = mfs(acs[pt][ct], wq[t & 1][ct][0], x[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    // the k steps [s0, s0 + KSH) out of the tile in LDS (which holds the channels from 16 s0 on)
    auto stage_b = [&](const int s0) {
      if constexpr (SPL) { stage_b_split(s0 / 2); return; }
      f32x4 wq[3][CT], xv[2][PT];
      w_load(wq[0], s0);
      if (KSH > 1) w_load(wq[1], s0 + 1);
      x_load(xv[0], 0);
#pragma unroll
      for (int s = 0; s < KSH; ++s) {
        if (s + 2 < KSH) w_load(wq[(s + 2) % 3], s0 + s + 2);         // weights two steps ahead of their MFMAs (L2 latency)
        if (s + 1 < KSH) x_load(xv[(s + 1) & 1], s + 1);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 (&w)[CT] = wq[s % 3];
        const f32x4 (&x)[PT] = xv[s & 1];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct].x, x[pt].x, acc[pt][ct], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct].y, x[pt].y, acc[pt][ct], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct].z, x[pt].z, acc[pt][ct], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct].w, x[pt].w, acc[pt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    stage_b(0);
    if constexpr (KH == 2) {
      __syncthreads();                                              // every wave has read the first half of x
      zero_tail();
      stage_a(std::integral_constant<int, 1>{});
      __syncthreads();
      stage_b(KSH);
    }
    __syncthreads();                                                // every wave has read x: t takes its place
    // lane = (pixel slot li of tile pt, channels ch0 + 16 ct + 4 kq .. + 3)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int ch = wave * (CM / 4) + 16 * ct + 4 * kq;             // channel within the block's slice
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b1 + c_first + ch);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int px = 16 * pt + li;
        f32x4 v = acc[pt][ct] + bb;
        if constexpr (SPL) v = (acc[pt][ct] + acs[pt][ct]) + bb;
        if (px < 49)
          *reinterpret_cast<f32x4*>(lds + tile_off<CM>(px, ch)) = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
      }
    }
    __syncthreads();
  }

  // ---- stage C: wave w = class w, the block's CM channels (tile channel c <-> V channel c_first + c) ----
  switch (wave) {
    case 0: mid_in_class<0, 0, CM, CMID>(a, img, lane, c_first, lds); break;
    case 1: mid_in_class<0, 1, CM, CMID>(a, img, lane, c_first, lds); break;
    case 2: mid_in_class<1, 0, CM, CMID>(a, img, lane, c_first, lds); break;
    default: mid_in_class<1, 1, CM, CMID>(a, img, lane, c_first, lds); break;
  }
}

bool wino_mid_supported(int Cin, int Cmid, bool gemm, int phases_in) {
  if (gemm) return (Cin == 128 && Cmid == 128 && (phases_in == 4 || phases_in == 1)) || (Cin == 256 && Cmid == 256 && phases_in == 1);
  return Cin == Cmid && (Cin == 128 || Cin == 256) && phases_in == 1;
}

hipError_t wino_mid_launch(const WinoMidArgs& a, hipStream_t st) {
  const bool gemm = a.w1 != nullptr;
  if (!wino_mid_supported(a.Cin, a.Cmid, gemm, a.phases_in) || a.n_img < 1 || !a.M || !a.V || (gemm && !a.b1) ||
      (a.x && (a.x_cs % 4 || a.x_coff % 4)))
    return hipErrorInvalidValue;
#define OFFK_MID_LAUNCH_KH(CI, CM, G, PH, NSP, KHV)                                                                \
  {                                                                                                                \
    constexpr int kLds = 64 * (CI / KHV > CM / NSP ? CI / KHV : CM / NSP) * 4;                                     \
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(wino_mid_kernel<CI, CM, G, PH, NSP, KHV>), kLds);   \
    if (e != hipSuccess) return e;                                                                                 \
    hipLaunchKernelGGL((wino_mid_kernel<CI, CM, G, PH, NSP, KHV>), dim3(a.n_img, NSP), dim3(256), kLds, st, a);    \
  }
  // split-fp32 stage B: two blocks per image, x tiles of 128 channels as plane images (48 KB; t behind them is smaller)
#define OFFK_MID_LAUNCH_SPL(CI, CM, PH, KHV)                                                                                  \
  {                                                                                                                           \
    constexpr int kLds = 4 * kPtStride<CI / KHV>;                                                                             \
    static_assert(kLds >= 64 * (CM / 2) * 4, "t tile behind the planes");                                                     \
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(wino_mid_kernel<CI, CM, true, PH, 2, KHV, true>), kLds);       \
    if (e != hipSuccess) return e;                                                                                            \
    hipLaunchKernelGGL((wino_mid_kernel<CI, CM, true, PH, 2, KHV, true>), dim3(a.n_img, 2), dim3(256), kLds, st, a);          \
    return hipGetLastError();                                                                                                 \
  }
  if (gemm && a.w1p && a.nsplit <= 0) {
    if (a.Cin == 128 && a.phases_in == 4) OFFK_MID_LAUNCH_SPL(128, 128, 4, 1)
    else if (a.Cin == 128) OFFK_MID_LAUNCH_SPL(128, 128, 1, 1)
    else OFFK_MID_LAUNCH_SPL(256, 256, 1, 2)
  }
#undef OFFK_MID_LAUNCH_SPL
#define OFFK_MID_LAUNCH(CI, CM, G, PH, NSP) OFFK_MID_LAUNCH_KH(CI, CM, G, PH, NSP, 1)
  // blocks per image (tools/bench_between.py, P = 384, device time): 128 -> 128: 20.8 us with one block per image, 18.7 with two;
  // 256 -> 256: 56.8 / 60.1 / 61.3 us with 1 / 2 / 4 (its x tile is 64 KB whatever the split: two blocks per CU either way); [r5] two
  // blocks per image with the x tile in two k halves (32 KB, three blocks per CU): 51.9 us against 60.7 in the forward, bit-identical
#ifndef OFFK_MID_NS256
#define OFFK_MID_NS256 22     /* the 256 -> 256 form: blocks per image (1, 2, 4), or 22 = two blocks per image with the x tile in two k halves (60.7 -> 51.9 us at P = 384; 2: 64.0) */
#endif
  const int ns = a.nsplit > 0 ? a.nsplit : (gemm && a.Cin == 128 ? 2 : OFFK_MID_NS256);
  if (gemm && a.Cin == 128 && a.phases_in == 4) { if (ns == 2) OFFK_MID_LAUNCH(128, 128, true, 4, 2) else OFFK_MID_LAUNCH(128, 128, true, 4, 1) }
  else if (gemm && a.Cin == 128) { if (ns == 2) OFFK_MID_LAUNCH(128, 128, true, 1, 2) else OFFK_MID_LAUNCH(128, 128, true, 1, 1) }
  else if (gemm) { if (ns == 4) OFFK_MID_LAUNCH(256, 256, true, 1, 4) else if (ns == 2) OFFK_MID_LAUNCH(256, 256, true, 1, 2) else if (ns == 22) OFFK_MID_LAUNCH_KH(256, 256, true, 1, 2, 2) else OFFK_MID_LAUNCH(256, 256, true, 1, 1) }
  else if (a.Cin == 128) OFFK_MID_LAUNCH(128, 128, false, 1, 1)
  else OFFK_MID_LAUNCH(256, 256, false, 1, 1)
#undef OFFK_MID_LAUNCH
#undef OFFK_MID_LAUNCH_KH
  return hipGetLastError();
}

}  // namespace offk
