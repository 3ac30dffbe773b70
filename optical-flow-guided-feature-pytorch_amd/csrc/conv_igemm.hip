// K4: channels-last implicit-GEMM convolution on the fp32 matrix cores (gfx950).
//
// Stands for the fusion convolutions of the OFF sub-network (reference
// RGB_OFF.py:657-685 fusion@28, :762-780 fusion@14, :833-841 fusion@7): nn.Conv2d +
// bias with the surrounding ReLU / residual-add folded into the epilogue.
//
// GEMM view:  Y[m][co] = sum_k A[m][k] * Wt[co][k]
//   m  = (image, ho, wo) output pixel, k = (ci / 32, kh, kw, ci % 32)  -- channels-last makes
//   every (pixel, tap) a contiguous run of Ci floats, so a 32-wide K tile is one tap and 32
//   consecutive channels (Ci % 32 == 0 for every fusion conv): one bounds check and one
//   16-B load per lane, no per-element im2col arithmetic.  The channel chunk is the OUTER
//   loop and the taps the inner one: the KH*KW tiles of one chunk re-read the same
//   32-channel slice of the input window, so the re-reads hit L2 instead of streaming
//   the whole 320..1056-channel rows again per tap (measured: the tap-outer order ran the
//   7x7 conv at 39 % L2 hit rate and 5x the input bytes fetched).
//   Work decomposition: 1-D grid over (k-split, m-tile, n-tile), remapped so that each XCD
//   (private 4 MiB L2) owns a contiguous range of m-tiles; optional split-K writes fp32
//   partial slabs that splitk_reduce_kernel sums in a fixed order (deterministic).
//   M = pixels sits on the MFMA row axis, N = out-channels on the lane axis, so each
//   accumulator register stores 32 consecutive channels of one pixel (128-B runs).
#include <cstdio>
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

struct ConvArgs {
  const float* x; int x_cs, x_coff;
  int n_img, H, W, Ci, Ho, Wo, pad;
  const float* w;      // [Co][KH*KW*Ci]
  const float* bias;   // [Co]
  int Co;
  const float* res; int res_cs, res_coff;
  float* y; int y_cs, y_coff;
  int flags, M;
  int gm, gn, splitk;  // m-tiles, n-tiles, k-splits (grid = gm*gn*splitk blocks)
  int co_limit;        // output channels >= co_limit are not stored (Co padded for the tiling)
  int batch;           // gridDim.y (1: a single problem)
  int spread;          // LDS-DMA form: issue the next tile's DMA pieces between the k-groups (grids of >= 1024 blocks, four per CU
                       // over: +0.5 % at B = 64) instead of in a bunch at the top of the step (small grids: with one block per CU
                       // the later issue is exposed latency -- B = 1: 0.81 vs 0.98 ms)
  long long x_bs, w_bs, y_bs;   // batched launch (gridDim.y problems of the same shape): element strides between problems
  // grouped batch (ngroups > 0; winograd.hip's polyphase form): gridDim.y = sum of g_batch; the problems of group g are dense
  // 1x1 GEMMs [M][g_Ci] x [Co][g_Ci] -> [M][Co] packed one behind the other from x + g_x / w + g_w / y + g_y
  int ngroups, g_batch[4], g_Ci[4];
  long long g_x[4], g_w[4], g_y[4];
  float* pool_part;    // != nullptr: per 32-row slab and output channel, the sums of the stored values over the slab's rows of its
  int pool_hw;         // first / second image (pool_hw rows per image, >= 32): [slab][2][Co] -- the average pool of a head folded in
  float* partial;      // [splitk][M][Co] when splitk > 1
  int no_dma;                  // tools: keep the register-staged loader (OFFK_CONV_DMA=0 in the tuning build)
  unsigned x_bytes, w_bytes;   // PREC 2 (fp32, buffer addressing): bytes behind p.x + p.x_coff / behind p.w, both < 2^31
#ifdef OFFK_CONV_TIMING
  unsigned long long* dbg;   // [8] timing sums of the bf16x3 producer / consumer waves (tools only)
#endif
#ifdef OFFK_TUNING_KNOBS
  int ablate;                // tools only (OFFK_CONV_ABLATE): 1 no activation loads, 2 no weight loads, 4 no LDS stores, 8 no MFMAs, 16 no epilogue
#endif
};

// PREC 0 / 2 / 4: three loaders in front of the fp32 MFMA core (PREC 1 / 3 were the two-plane bf16x3 core of rounds 1 - 4, retired with that
// mode in round 5: a 512-thread producer / consumer form of this kernel -- git history).
// PREC 2 is PREC 0 with a LEAN K loop (round 2).  On gfx950 the fp32 MFMA shares the SIMD's fp32 lanes with the vector ALU:
// tools/mfma_f32_probe.hip shows every VALU instruction in the loop coming straight out of the matrix throughput, at any
// occupancy (92 % with none, 76 % with 32 extra per 16 MFMAs, 63 % with 64).  The PREC 0 loop spends ~12 VALU instructions
// per activation row and K-tile on addressing (two 64-bit multiplies, pointer selects) plus zero / ReLU selects before the
// LDS store.  PREC 2 addresses through buffer descriptors instead: the per-row offset is loop-invariant, the (tap, chunk)
// offset is a scalar (soffset), padding taps take the out-of-range offset 0x80000000 and the hardware returns zeros --
// three VALU instructions per activation row visit (two shifts and an and-or on precomputed separable tap masks), none per
// weight row, none for LDS addresses (the K loop is unrolled by two: stages are literals).  Needs both tensors below 2^31
// bytes (checked on the host, PREC 0 otherwise).
// PREC 4 is PREC 2 with the tiles moved global -> LDS by the load unit itself (`buffer_load_dwordx4 ... lds`): no
// prefetch registers, no ds_write, no s_waitcnt vmcnt in front of an LDS store.  The probe prices four ds_write_b128 per
// K-tile at 8-9 % of the matrix throughput (92 -> 84 %); in the forward the step is worth 1.4 % (same-box 5.40 -> 5.33 ms).
// A wave instruction lands 64 x 16 B contiguously (8 rows of 128 B), so rows cannot be padded: position (row, slot) holds
// the row's 16-byte chunk slot ^ ((row >> 1) & 7) -- the swizzle goes into each lane's SOURCE offset -- and the
// ds_read_b128 of 16 consecutive rows stays conflict-free.  No ReLU-on-load (the data never passes a register): convs
// with OFFK_CONV_RELU_IN use PREC 2.
// Threads: 256 (4 waves, every wave loads and multiplies).
constexpr int kDmaStages = 2;   // 3 (tiles two ahead, 48 KB for the 64x64 tile = 3 blocks per CU) measured 2 % slower than 2
template <int KH, int KW, int S, int TM, int TN, int WM, int WN, int PREC>
__global__ __launch_bounds__(256, 1) void conv_igemm_kernel(ConvArgs p) {
  static_assert(PREC == 0 || PREC == 2 || PREC == 4, "fp32 core only");
  constexpr bool LEAN = PREC >= 2;
#ifdef OFFK_CONV_TIMING
  const unsigned long long tm_t0 = __builtin_readcyclecounter();
  unsigned long long tm_t1 = 0, tm_t2 = 0, tm_ta = 0;
#endif
  if (p.ngroups > 0) {          // grouped batch: this block's group and its problem inside the group (static indices: scalar code)
    int b = blockIdx.y, ci = p.g_Ci[0], first = 0;
    long long gx = p.g_x[0], gw = p.g_w[0], gy = p.g_y[0];
#pragma unroll
    for (int g = 1; g < 4; ++g) {
      first += p.g_batch[g - 1];
      if (g < p.ngroups && (int)blockIdx.y >= first) { b = blockIdx.y - first; ci = p.g_Ci[g]; gx = p.g_x[g]; gw = p.g_w[g]; gy = p.g_y[g]; }
    }
    p.x += gx + (long long)b * p.M * ci; p.w += gw + (long long)b * p.Co * ci; p.y += gy + (long long)b * p.M * p.Co;
    p.Ci = ci; p.x_cs = ci; p.x_bytes = (unsigned)p.M * (unsigned)ci * 4u; p.w_bytes = (unsigned)p.Co * (unsigned)ci * 4u;
  } else if (gridDim.y > 1) { p.x += blockIdx.y * p.x_bs; p.w += blockIdx.y * p.w_bs; p.y += blockIdx.y * p.y_bs; }   // batched GEMMs (winograd.hip)
  constexpr bool DMA = PREC == 4;                          // fp32 core, tiles DMA'd into swizzled 128-byte LDS rows
  constexpr int DST = kDmaStages;                          // DMA: LDS stages (tiles run DST - 1 ahead of the MFMAs)
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As0 = smem;                         // fp32: [2][BM][LDS_K]
  float* Bs0 = smem + 2 * BM * LDS_K;        // fp32: [2][BN][LDS_K]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = threadIdx.x & 255;          // index within the 4 loading waves (fp32: all; bf16x3: producers)
  const int wm = (wave & 3) / WN, wn = (wave & 3) % WN;
  // XCD-aware bijective remap of the linear block id (blocks are dealt round-robin over the 8
  // XCDs): XCD x owns the contiguous logical range, ordered n-tile fastest, then m-tile, then split
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = lid % p.gn, mt = (lid / p.gn) % p.gm, zs = lid / (p.gn * p.gm);
  const int m0 = mt * BM, n0 = nt * BN;
  constexpr int TAPS = KH * KW;
  const int K = TAPS * p.Ci;
  const int nkt_all = TAPS * (p.Ci / BK);
  // (unsplit: no 64-bit divisions in the block's set-up -- the set-up runs with every co-resident block in it at the same time)
  const int kt_begin = p.splitk == 1 ? 0 : (int)((long long)nkt_all * zs / p.splitk);
  const int kt_end = p.splitk == 1 ? nkt_all : (int)((long long)nkt_all * (zs + 1) / p.splitk);

  // Loader mapping (both precisions): thread = (row (tid>>3) + 32*r, 16-B chunk tid&7), so every load
  // instruction of a wave reads eight whole 128-B lines (the texture-address path works per line; an
  // earlier 32-B-per-lane bf16x3 mapping touched every line with two instructions and was load-bound).
  //   activations: 4 fp32 k-values per visit; bf16x3 splits them into 8 B of hi + 8 B of lo.
  //   weights: fp32 [Co][K] (fp32 path) or, pre-split, [Co][K/32][hi 32 | lo 32] bf16 (bf16x3 path):
  //            chunks 0-3 of a row's 128 B are the hi plane of the K-tile, chunks 4-7 the lo plane.
  constexpr int RSH = 3, RSTEP = 32;
  constexpr int NRA = BM / RSTEP, NRB = BN / RSTEP;  // row visits per thread
  constexpr int VA = 1, VB = 1;
  const int kpos = DMA ? 4 * ((tid & 7) ^ ((tid >> 4) & 7)) : 4 * (tid & 7);    // DMA: the chunk this LDS slot holds (row = (tid >> 3) + 32 r)

  int hi0[NRA], wi0[NRA];
  int pix0[NRA];      // image * H * W (fits: M and the feature maps are < 2^31 elements)
  // 1x1 / stride 1 / no padding (every 1x1 conv of the network and the batched Winograd GEMMs): the input pixel of output row m IS m --
  // no (image, row, column) split, i.e. none of the four integer divisions per thread below (round 4: the set-up of a block was
  // 6-9 k cycles of its ~140 k, and co-resident blocks run in step, so nobody multiplies meanwhile; tools/conv_dma_timing.py)
  const bool flat = TAPS == 1 && S == 1 && p.pad == 0 && LEAN;
#pragma unroll
  for (int r = 0; r < NRA; ++r) {
    int m = m0 + (tid >> RSH) + RSTEP * r;
    bool ok = m < p.M;
    if (flat) { hi0[r] = ok ? 0 : -100000; wi0[r] = 0; pix0[r] = ok ? m : 0; continue; }
    int mm = ok ? m : 0;
    int img = mm / (p.Ho * p.Wo);
    int rem = mm - img * (p.Ho * p.Wo);
    int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    hi0[r] = ok ? ho * S - p.pad : -100000;  // pushes every tap out of bounds
    wi0[r] = wo * S - p.pad;
    pix0[r] = img * p.H * p.W;
  }
  const float* xbase = p.x + p.x_coff + kpos;
  const bool relu_in = p.flags & OFFK_CONV_RELU_IN_;
  // fp32: weight rows of this block's N slab, fp32 [Co][K].  bf16x3: bf16 hi plane [Co][K] then lo plane.
  const float* wbase = p.w + (size_t)(n0 + (tid >> 3)) * K + 4 * (tid & 7);   // same byte offsets in both formats

  // ---- PREC 2: buffer addressing (see the header comment) ----
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  int rowoff[LEAN ? NRA : 1], woff[LEAN ? NRB : 1];
  unsigned inv[LEAN ? NRA : 1];
  __amdgpu_buffer_rsrc_t xrs, wrs;
  if constexpr (LEAN) {
    // the descriptor base sits (pad rows + pad pixels) in front of the tensor, so that the per-row offset of the window's
    // top-left corner is never negative; the scalar tap offset brings every VALID tap back inside the tensor
    const int bias_px = p.pad * p.W + p.pad;
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + p.x_coff - (long)bias_px * p.x_cs, 0,
                                            (int)(p.x_bytes + (unsigned)bias_px * p.x_cs * 4u), 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < NRA; ++r) {    // hi0 + pad = ho * S, wi0 + pad = wo * S
      rowoff[r] = hi0[r] < -50000 ? 0 : flat ? (pix0[r] * p.x_cs + kpos) * 4
                                             : ((pix0[r] + (hi0[r] + p.pad) * p.W + (wi0[r] + p.pad)) * p.x_cs + kpos) * 4;   // rows past M: never valid
      // tap validity, separable: bit kh = input row hi0 + kh outside the image, bit 16 + kw = input column outside.
      // Per K-tile the test is then two shifts and an and-or (the fp32 MFMA shares its lanes with the VALU: every
      // vector instruction in this loop is matrix time), against two adds, two compares and a select.
      unsigned m = 0;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) m |= (unsigned)(hi0[r] + kh) >= (unsigned)p.H ? 1u << kh : 0u;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) m |= (unsigned)(wi0[r] + kw) >= (unsigned)p.W ? 0x10000u << kw : 0u;
      if (flat) m = hi0[r] < -50000 ? 1u : 0u;
      inv[r] = m;
      if constexpr (TAPS == 1) rowoff[r] = m ? (int)0x80000000 : rowoff[r];
    }
#pragma unroll
    for (int r = 0; r < NRB; ++r) woff[r] = ((n0 + (tid >> 3) + 32 * r) * K + kpos) * 4;
  }
  // DMA: tile kt straight into LDS stage `stage`: wave w lands rows 8w + 32r .. + 7 of the A / B image, 1 KB per instruction.
  // The instruction is issued through inline asm: with the builtin hipcc treats every later ds_read as a possible reader of
  // the landing zone and puts s_waitcnt vmcnt(0) between the DMA and the MFMAs of the OTHER stage -- the overlap this
  // path exists for.  Ordering is explicit instead: s_waitcnt vmcnt(0) + barrier before a stage is read.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 xdesc = {0, 0, 0, 0}, wdesc = {0, 0, 0, 0};
  unsigned lds_base = 0;
  if constexpr (DMA) {
    const int bias_px = p.pad * p.W + p.pad;
    const unsigned long long xa = reinterpret_cast<unsigned long long>(p.x + p.x_coff - (long)bias_px * p.x_cs);
    const unsigned long long wa = reinterpret_cast<unsigned long long>(p.w);
    xdesc = i32x4{(int)(unsigned)xa, (int)(unsigned)(xa >> 32) & 0xffff, (int)(p.x_bytes + (unsigned)bias_px * p.x_cs * 4u), 0x00020000};
    wdesc = i32x4{(int)(unsigned)wa, (int)(unsigned)(wa >> 32) & 0xffff, (int)p.w_bytes, 0x00020000};
    lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) +
               (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (8 * 128);
  }
  auto dma16 = [&](const i32x4& desc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(desc), "s"(soff) : "memory", "m0");
  };
  // dma_prep: scalar offsets of tile kt; dma_piece(i): row visit i of the tile (A rows first) into LDS stage `stage`.  In the K
  // loop the pieces go out one behind each k-group's MFMAs (round 3: issued in a bunch at the top of the step they cost the
  // wave 60-180 cycles each with nothing of its own multiplying behind them; pw_tdiff.hip measured the same)
  int dm_soff_a = 0, dm_soff_b = 0, dm_kh = 0, dm_kw = 0;
  auto dma_prep = [&](int kt) {
    if constexpr (DMA) {
      int chunk = kt / TAPS, tap = kt - chunk * TAPS, c0 = chunk * BK;
      dm_kh = tap / KW; dm_kw = tap - dm_kh * KW;
      dm_soff_a = ((dm_kh * p.W + dm_kw) * p.x_cs + c0) * 4; dm_soff_b = kt * BK * 4;     // scalar
    }
  };
  auto dma_piece = [&](const int i, const int stage) {
    if constexpr (DMA) {
      const unsigned la = lds_base + stage * (BM * 128), lb = lds_base + DST * (BM * 128) + stage * (BN * 128);
      if (i < NRA) {
        int voff = rowoff[i];
        if constexpr (TAPS > 1) voff |= (int)(((inv[i] << (31 - dm_kh)) | (inv[i] << (15 - dm_kw))) & 0x80000000u);
        dma16(xdesc, la + i * 32 * 128, voff, dm_soff_a);
      } else {
        dma16(wdesc, lb + (i - NRA) * 32 * 128, woff[i - NRA], dm_soff_b);
      }
    }
  };
  auto dma_tile = [&](int kt, const int stage) {
    dma_prep(kt);
#pragma unroll
    for (int i = 0; i < NRA + NRB; ++i) dma_piece(i, stage);
  };
  constexpr int NRG = NRA * VA + NRB * VB;
  // prefetch registers: A rows then B rows (one array per set: separate A / B arrays end up in scratch).
  float4 rg0[NRG];
  unsigned okm0 = 0;   // bit r: activation row r of the set is inside the image (else it reads as zero)
  // Loads are branch-free and nothing touches the loaded values here: a padding tap reads pixel 0 through a
  // selected pointer and is zeroed by store_tile, ReLU-on-load is applied there as well.  With
  // `t = 0; if (ok) t = load; rg = relu_in ? relu(t) : t` every load was followed by s_waitcnt vmcnt(0): the
  // row visits of a tile went out one memory latency after the other.
  auto load_tile = [&](float4 (&rg)[NRG], unsigned& okm, int kt) {
    int chunk = kt / TAPS, tap = kt - chunk * TAPS, c0 = chunk * BK;
    int kh = tap / KW, kw = tap - kh * KW;
    if constexpr (LEAN) {
      const int soff_a = ((kh * p.W + kw) * p.x_cs + c0) * 4, soff_b = kt * BK * 4;     // scalar
#pragma unroll
      for (int r = 0; r < NRA; ++r) {
#ifdef OFFK_TUNING_KNOBS
        if (p.ablate & 1) continue;
#endif
        int voff = rowoff[r];        // a padding tap: an offset past the descriptor, the load returns zeros
        if constexpr (TAPS > 1) voff |= (int)(((inv[r] << (31 - kh)) | (inv[r] << (15 - kw))) & 0x80000000u);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, soff_a, 0);
        rg[r] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
#ifdef OFFK_TUNING_KNOBS
        if (p.ablate & 2) continue;
#endif
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[r], soff_b, 0);
        rg[NRA * VA + r] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      okm = 0xffffffffu;      // padding taps already read as zeros
      return;
    }
    unsigned mask = 0;
#pragma unroll
    for (int r = 0; r < NRA; ++r) {
      int hi = hi0[r] + kh, wi = wi0[r] + kw;
      bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      mask |= ok ? (1u << r) : 0u;
#ifdef OFFK_TUNING_KNOBS
      if (p.ablate & 1) continue;
#endif
      rg[r] = *reinterpret_cast<const float4*>(ok ? xbase + (size_t)(pix0[r] + hi * p.W + wi) * p.x_cs + c0 : xbase);
    }
    okm = mask;
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
#ifdef OFFK_TUNING_KNOBS
      if (p.ablate & 2) continue;
#endif
      rg[NRA * VA + r] = *reinterpret_cast<const float4*>(wbase + (size_t)32 * r * K + kt * BK);
    }
  };
  auto a_value = [&](const float4 (&rg)[NRG], unsigned okm, int r) {
    float4 t = rg[r];
    if (!LEAN && !((okm >> r) & 1u)) t = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LEAN) {
      // one integer max per value (a negative float is a negative integer); fmaxf costs a canonicalise + a max
      auto rl = [](float x) { return __int_as_float(max(__float_as_int(x), 0)); };
      return relu_in ? make_float4(rl(t.x), rl(t.y), rl(t.z), rl(t.w)) : t;
    }
    return relu_in ? relu4(t) : t;
  };
  auto store_tile = [&](const float4 (&rg)[NRG], unsigned okm, int stage) {
#ifdef OFFK_TUNING_KNOBS
    if (p.ablate & 4) return;
#endif
    {
      float* As = As0 + stage * BM * LDS_K;
#pragma unroll
      for (int r = 0; r < NRA; ++r)
        *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = a_value(rg, okm, r);
      float* Bs = Bs0 + stage * BN * LDS_K;
#pragma unroll
      for (int r = 0; r < NRB; ++r)
        *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[NRA + r];
    }
  };

  WaveAcc<TM, TN> acc;
  acc.zero();
  if constexpr (DMA) {
    // per-lane LDS byte offsets of the four k-groups: row r32 of the wave's slab, chunk (2g + h) ^ swizzle(row)
    const int r32 = lane & 31, hh = lane >> 5, sw = hh ^ ((r32 >> 1) & 7);
    int aoff[BK / 8], boff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      aoff[g] = (wm * (32 * TM) + r32) * 128 + ((sw ^ (2 * g)) << 4);
      boff[g] = DST * (BM * 128) + (wn * (32 * TN) + r32) * 128 + ((sw ^ (2 * g)) << 4);
    }
    const char* const lds_c8 = reinterpret_cast<const char*>(smem);
    auto mma = [&](const int st, const int st_next) {
      float4 a[2][TM], b[2][TN];
      auto rd = [&](const int g, float4 (&aa)[TM], float4 (&bb)[TN]) {
#pragma unroll
        for (int t = 0; t < TM; ++t) aa[t] = *reinterpret_cast<const float4*>(lds_c8 + st * (BM * 128) + t * 32 * 128 + aoff[g]);
#pragma unroll
        for (int t = 0; t < TN; ++t) bb[t] = *reinterpret_cast<const float4*>(lds_c8 + st * (BN * 128) + t * 32 * 128 + boff[g]);
      };
      rd(0, a[0], b[0]);
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        if (g + 1 < BK / 8) rd(g + 1, a[(g + 1) & 1], b[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc.acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].x, b[g & 1][tn].x, acc.acc[tm][tn], 0, 0, 0);
            acc.acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].y, b[g & 1][tn].y, acc.acc[tm][tn], 0, 0, 0);
            acc.acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].z, b[g & 1][tn].z, acc.acc[tm][tn], 0, 0, 0);
            acc.acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][tm].w, b[g & 1][tn].w, acc.acc[tm][tn], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (st_next >= 0) {
#pragma unroll
          for (int i = g; i < NRA + NRB; i += BK / 8) dma_piece(i, st_next);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // s_waitcnt vmcnt(n): all but this thread's newest n DMA loads have landed (gfx9 encoding: vmcnt = bits 3:0 and 15:14)
    constexpr int NL = NRA + NRB;
    constexpr int kWaitTile = DST == 3 ? (0x0F70 | (NL & 15) | ((NL >> 4) << 14)) : 0x0F70;
#ifdef OFFK_CONV_TIMING
    tm_ta = __builtin_readcyclecounter();
#endif
    dma_tile(kt_begin, 0);
    // Two stages: the SECOND tile goes out here as well (both stages are free), not from inside the first step -- the blocks of a CU start
    // together, so the first step's MFMAs (4 x 1024 cycles per SIMD) do not cover a cold tile's ~10 k cycles; in flight beside the first
    // tile it costs no latency of its own.  The first step then issues nothing (peeled below).
    constexpr bool PEEL = DST == 2;
    if constexpr (DST == 3 || PEEL) dma_tile(min(kt_begin + 1, kt_end - 1), 1);
    constexpr int kWaitFirst = (DST == 3 || PEEL) ? (0x0F70 | (NL & 15) | ((NL >> 4) << 14)) : 0x0F70;
    __builtin_amdgcn_s_waitcnt(kWaitFirst);   // the first tile is in LDS (all but this thread's newest NL loads have landed)
    __syncthreads();
#ifdef OFFK_CONV_TIMING
    tm_t1 = __builtin_readcyclecounter();
#endif
    // one K-tile: the load unit fills the stage every wave left at the last barrier with tile kt + DST - 1 while the MFMAs
    // read stage st (a literal: the loop is unrolled by DST)
    auto step = [&](int kt, const int st) {
      if (DST == 2 && p.spread) {
        dma_prep(min(kt + 1, kt_end - 1));
        mma(st, st ^ 1);                       // the pieces of tile kt + 1 go out between the k-groups
      } else {
        dma_tile(min(kt + DST - 1, kt_end - 1), (st + DST - 1) % DST);
        __builtin_amdgcn_sched_barrier(0);
        mma(st, -1);
      }
      __builtin_amdgcn_s_waitcnt(kWaitTile);   // tile kt + 1 has landed
      __syncthreads();
    };
    int kt = kt_begin;
    if constexpr (DST == 3) {
      for (; kt + 2 < kt_end; kt += 3) {
        step(kt, 0);
        step(kt + 1, 1);
        step(kt + 2, 2);
      }
      if (kt < kt_end) step(kt, 0);
      if (kt + 1 < kt_end) step(kt + 1, 1);
    } else {
      // peeled first step: tile kt_begin + 1 is already on its way into stage 1
      mma(0, -1);
      __builtin_amdgcn_s_waitcnt(kWaitTile);
      __syncthreads();
      ++kt;
      for (; kt + 1 < kt_end; kt += 2) {
        step(kt, 1);
        step(kt + 1, 0);
      }
      if (kt < kt_end) step(kt, 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // nothing of this block may still be landing when its LDS is handed on
    // (Round 4 tried to hide the next block's first-tile latency -- co-resident blocks run in step, so nobody multiplies while they
    //  wait ~10 k cycles for their first tiles: a block touched the first K-tiles of the block 1024 positions later in launch order,
    //  the next occupant of its slot on the same XCD, before its epilogue.  Every launch got SLOWER, 1-20 %: the loads keep the wave
    //  alive until they return.  profiles/r04/conv_block_cycle_split.txt)
#ifdef OFFK_CONV_TIMING
    tm_t2 = __builtin_readcyclecounter();
#endif
  } else {
    load_tile(rg0, okm0, kt_begin);
    store_tile(rg0, okm0, 0);
    __syncthreads();
    // one K-tile: prefetch tile kt + 1 into registers, multiply stage `st`, store the prefetch into stage st ^ 1.
    // `st` is a literal at both call sites (the loop is unrolled by two), so every LDS address is a register + an
    // immediate; with st = (kt - kt_begin) & 1 each K-tile paid three vector adds for the stage offset.
    auto step = [&](int kt, const int st) {
      // LEAN: unconditional prefetch + store (the last step re-loads its own tile into the idle stage): with
      // `if (kt + 1 < kt_end) load_tile(...)` the two paths merge in a phi and hipcc reconciles the prefetch registers
      // with v_mov copies of the just-issued loads -- an s_waitcnt vmcnt in FRONT of the MFMAs, i.e. a memory latency
      // exposed per K-tile instead of hidden behind them
      if constexpr (LEAN) {
        load_tile(rg0, okm0, min(kt + 1, kt_end - 1));
        __builtin_amdgcn_sched_barrier(0);    // keep the loads in FRONT of the MFMAs (hipcc sinks them behind most of them otherwise)
      } else {
        if (kt + 1 < kt_end) load_tile(rg0, okm0, kt + 1);     // 64-bit pointer loader: the conditional form measured faster
      }
#ifdef OFFK_TUNING_KNOBS
      if (!(p.ablate & 8))
#endif
      acc.mma_ktile(As0 + st * BM * LDS_K + wm * (32 * TM) * LDS_K,
                    Bs0 + st * BN * LDS_K + wn * (32 * TN) * LDS_K, lane);
      if (LEAN || kt + 1 < kt_end) store_tile(rg0, okm0, st ^ 1);
      __syncthreads();
    };
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {      // (an exit between the two steps made hipcc copy the accumulators every round)
      step(kt, 0);
      step(kt + 1, 1);
    }
    if (kt < kt_end) step(kt, 0);
  }

  const int r32 = lane & 31, h = lane >> 5;
#ifdef OFFK_TUNING_KNOBS
  if (p.ablate & 16) {   // no epilogue: keep the accumulators alive without storing them
    float sacc = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) sacc += acc.acc[tm][tn][reg];
    if (sacc == 12345.678f) p.y[0] = sacc;
    return;
  }
#endif
  if (p.splitk > 1) {   // raw partial sums; bias / ReLU / residual happen in splitk_reduce_kernel
    float* part = p.partial + (size_t)zs * p.M * p.Co;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int m = m0 + wm * 32 * TM + tm * 32 + acc_row(reg, h);
          if (m < p.M) part[(size_t)m * p.Co + n0 + wn * 32 * TN + tn * 32 + r32] = acc.acc[tm][tn][reg];
        }
    return;
  }
  // epilogue: y = post( pre(acc + bias) + res )
  const bool relu_pre = p.flags & OFFK_CONV_RELU_PRE_, relu_post = p.flags & OFFK_CONV_RELU_POST_;
  // No residual, no pooled sums and y well below 2^31 bytes (the batched Winograd GEMMs, conv1 of a bottleneck): stores through a buffer
  // descriptor that ends behind row M - 1 -- a row past M falls outside and is dropped by the hardware: ONE per-lane offset per
  // accumulator tile plus a scalar per register, no compare / exec mask / 64-bit address per store (the pointer form spent ~11 k cycles
  // issuing a block's 16 stores per wave, with every co-resident block in its epilogue at the same time: tools/conv_dma_timing.py)
  const unsigned long long ybytes = ((unsigned long long)(p.M > 0 ? p.M - 1 : 0) * p.y_cs + p.y_coff + p.co_limit) * 4ull;
  const bool fast_store = LEAN && !p.res && !p.pool_part && p.M > 0 && ybytes < 0x7f000000ull;
  if (fast_store) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)ybytes, 0x00020000);
    const int ycs4 = p.y_cs * 4;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int co = n0 + wn * 32 * TN + tn * 32 + r32;
      const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int mb = m0 + wm * 32 * TM + tm * 32;                      // (scalar) first row of the tile
        // (the row goes into the per-lane offset, not the scalar one: only the former is compared with the descriptor's size.  Staging the
        //  tile through LDS to leave in four 16-byte stores per lane instead of sixteen dwords measured no better: 6.8 k vs 7.9 k cycles to
        //  the last store's issue, most of which is the SIMD's queue of MFMAs draining -- profiles/r04/conv_block_cycle_split.txt)
        const int voff = co < p.co_limit ? (4 * h * p.y_cs + p.y_coff + co) * 4 : (int)0x80000000;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float v = acc.acc[tm][tn][reg] + bv;
          if (relu_pre || relu_post) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, voff + (mb + (reg & 3) + 8 * (reg >> 2)) * ycs4, 0, 0);
        }
      }
    }
  } else {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int co = n0 + wn * 32 * TN + tn * 32 + r32;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      // the 16 residual values of this accumulator tile first, branch-free (a load per element in front of its
      // use costs one L2 latency per element: the compiler waits after each)
      float rv[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) rv[reg] = 0.f;
      if (p.res) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = m0 + wm * 32 * TM + tm * 32 + acc_row(reg, h);
          const bool ok = m < p.M && co < p.co_limit;
          rv[reg] = *(ok ? p.res + (size_t)m * p.res_cs + p.res_coff + co : p.res);
        }
      }
      // values first, in straight-line code (one wait for the residuals), conditional stores afterwards: a loaded
      // value used inside the `if (row in range)` blocks makes every block start with s_waitcnt vmcnt(0), which
      // also waits for the previous block's store
      float ov[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc.acc[tm][tn][reg] + bv;
        if (relu_pre) v = fmaxf(v, 0.f);
        v += rv[reg];
        if (relu_post) v = fmaxf(v, 0.f);
        asm volatile("" : "+v"(v));   // keep the arithmetic out of the conditional blocks below
        ov[reg] = v;
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 32 * TM + tm * 32 + acc_row(reg, h);
        if (m < p.M && co < p.co_limit) p.y[(size_t)m * p.y_cs + p.y_coff + co] = ov[reg];
      }
      if (p.pool_part) {
        // column sums of this accumulator tile, split at the image boundary inside the slab (an image has >= 32 rows, so
        // a slab touches at most two); fixed order: registers ascending, then the lower half-wave + the upper one
        const int slab = (m0 + wm * 32 * TM + tm * 32) >> 5;
        const int b1 = ((slab * 32) / p.pool_hw + 1) * p.pool_hw;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = slab * 32 + acc_row(reg, h);
          const float v = m < p.M ? ov[reg] : 0.f;
          s0 += m < b1 ? v : 0.f;
          s1 += m < b1 ? 0.f : v;
        }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (h == 0 && co < p.co_limit && slab * 32 < p.M) {     // (a sub-tile wholly past M owns no slab: the region ends at ceil(M / 32))
          p.pool_part[((size_t)slab * 2) * p.Co + co] = s0;
          p.pool_part[((size_t)slab * 2 + 1) * p.Co + co] = s1;
        }
      }
    }
  }
  }
#ifdef OFFK_CONV_TIMING
  if constexpr (DMA) {      // LDS-DMA form, wave 0 of every block: prologue (entry -> first tile in LDS), K loop, epilogue (incl. its stores' drain)
    const unsigned long long tm_te = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg && threadIdx.x == 0) {
      const unsigned long long tm_t3 = __builtin_readcyclecounter();
      atomicAdd(p.dbg + 8, tm_t1 - tm_t0); atomicAdd(p.dbg + 9, tm_t2 - tm_t1); atomicAdd(p.dbg + 10, tm_t3 - tm_t2);
      atomicAdd(p.dbg + 13, tm_ta - tm_t0); atomicAdd(p.dbg + 14, tm_te - tm_t2);
      atomicAdd(p.dbg + 11, 1ull); atomicAdd(p.dbg + 12, (unsigned long long)(kt_end - kt_begin));
    }
  }
#endif
}

// sum of the split-K slabs in split order, then the conv epilogue; 4 channels per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs p) {
  const size_t n4 = (size_t)p.M * (p.Co >> 2);
  const bool relu_pre = p.flags & OFFK_CONV_RELU_PRE_, relu_post = p.flags & OFFK_CONV_RELU_POST_;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / (p.Co >> 2);
    const int co = (int)(i - m * (p.Co >> 2)) * 4;
    float4 v = *reinterpret_cast<const float4*>(p.partial + m * p.Co + co);
    for (int z = 1; z < p.splitk; ++z) {
      const float4 u = *reinterpret_cast<const float4*>(p.partial + ((size_t)z * p.M + m) * p.Co + co);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + co); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (relu_pre) v = relu4(v);
    if (p.res) {
      const float4 r = *reinterpret_cast<const float4*>(p.res + m * p.res_cs + p.res_coff + co);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu_post) v = relu4(v);
    *reinterpret_cast<float4*>(p.y + m * p.y_cs + p.y_coff + co) = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// "Patch" variant of the kernel (tile_cfg 6: 128 output channels per block, 7: 64) for the k x k convs (first built for the bf16x3 core).
// Every k x k conv of the fusion stages produces 196 output pixels per image (14x14) or per four images (7x7), from
// 784 or 196 input pixels.  A block owns one such 196-pixel output group (seven 32-row MFMA tiles) and, per 32-channel
// chunk, keeps the WHOLE input patch of the group in LDS: the KH*KW taps then read
// their A operand straight from that patch at shifted pixel slots, so an input value crosses L2 -> LDS once per chunk
// instead of once per tap, and only the weight tile of the (chunk, tap) streams through the two-stage LDS ring.
// Operand traffic per FLOP drops ~4x against the 128x128 im2col tile; the generic kernel was bound by exactly that
// traffic (PMC: 45-55 % of the wave cycles waiting, the 7x7 conv fetching 3.4x its input from HBM because the
// per-block windows of the resident blocks do not fit the XCD's L2).
//   patch slot of input pixel (img, y, x) = img*W*W + y*W + xperm(x); for stride 2 xperm puts the even columns of a
//   row first, so the 32 consecutive output pixels of an MFMA row tile read consecutive slots for every tap (the
//   16-B chunk swizzle of the generic kernel then keeps the ds_read_b128 conflict-free); slot NP is a zero pixel
//   that out-of-image taps read.
// Waves 0-3 multiply (wave = one 32-column tile x all seven row tiles for 128 channels; column tile x alternate row
// tiles for 64), waves 4-7 stream the weight tiles two steps ahead and hold the next chunk's patch in registers.
// ---------------------------------------------------------------------------------------------------
// fp32 (v_mfma_f32_32x32x2_f32): ONE plane of fp32 rows of 128 B whose 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7): a
// ds_read_b128 lane group (16 consecutive rows, same chunk) then covers all 16 slots of the 256-byte bank row.  The fp32
// form exists because the generic fp32 kernel re-fetches every input value once per tap from L2: its load / LDS-store
// pipeline alone took as long as the MFMAs (tools/conv_ablate.py: 0.93 ms of staging beside 0.96 ms of MFMAs on the 7x7),
// so the matrix pipe sat at 68-70 %; with the patch in LDS the fp32 consumers are purely MFMA-bound.
template <int KH, int S, int W, int NT, int PREC>
__global__ __launch_bounds__(512, 2) void conv_patch_kernel(ConvArgs p) {
  static_assert(PREC == 0, "fp32 only (PREC 1 was the two-plane bf16x3 form, retired in round 5)");
  constexpr int KW = KH, TAPS = KH * KW, PAD = KH / 2, H = W, HW = H * W;
  constexpr int WO = (W + 2 * PAD - KW) / S + 1, HOWO = WO * WO, IMG = 196 / HOWO, NP = IMG * HW;
  static_assert(196 % HOWO == 0 && (S == 1 || (W % 2) == 0), "196-pixel output groups only");
  constexpr int BN = 32 * NT, RT = NT == 4 ? 7 : 4, B3R = 128;
  constexpr int PATCH_PLANE = ((NP + 1) * B3R + 127) / 128 * 128, B_PLANE = BN * B3R, B_STAGE = B_PLANE;
  constexpr int NJ = (NP + 31) / 32;   // patch pixels per producer thread
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  char* const patch = lds_c;
  char* const bst = lds_c + PATCH_PLANE;            // two stages of B

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = threadIdx.x & 255;
  int lid;
  {
    const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = lid % p.gn, mt = (lid / p.gn) % p.gm, zs = lid / (p.gn * p.gm);
  const int n0 = nt * BN;
  const int K = TAPS * p.Ci, nch = p.Ci / BK;
  const int c_begin = (int)((long long)nch * zs / p.splitk), c_end = (int)((long long)nch * (zs + 1) / p.splitk);
  const int G = (c_end - c_begin) * TAPS;           // (chunk, tap) steps of this block
  const bool relu_in = p.flags & OFFK_CONV_RELU_IN_;

  if (wave >= 4) {
    // ================================ producers ================================
    const int c8 = tid & 7;
    const float* wbase = p.w + (size_t)(n0 + (tid >> 3)) * K + 4 * c8;
    const float* xbase = p.x + p.x_coff + 4 * c8;
    const long long npix_all = (long long)p.n_img * HW;
    float4 rgP[NJ + 2 * NT];   // patch pieces, then the two weight-tile sets (one array: separate ones go to scratch)
    constexpr int B0 = NJ, B1 = NJ + NT;
    unsigned okp = 0;
    auto load_patch = [&](int chunk) {
      unsigned m = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pp = (tid >> 3) + 32 * j;
        const long long gp = (long long)mt * NP + pp;
        const bool ok = pp < NP && gp < npix_all;
        m |= ok ? (1u << j) : 0u;
        rgP[j] = *reinterpret_cast<const float4*>(ok ? xbase + (size_t)gp * p.x_cs + chunk * BK : xbase);
      }
      okp = m;
    };
    auto store_patch = [&]() {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pp = (tid >> 3) + 32 * j;
        if (pp < NP) {
          float4 t = rgP[j];
          if (!((okp >> j) & 1u)) t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (relu_in) t = relu4(t);
          const int img = pp / HW, rem = pp - img * HW, y = rem / W, x = rem - y * W;
          const int xp = S == 2 ? (x & 1) * (W / 2) + (x >> 1) : x;
          const int slot = img * HW + y * W + xp;
          *reinterpret_cast<float4*>(patch + slot * B3R + ((c8 ^ ((slot >> 1) & 7)) << 4)) = t;
        }
      }
    };
    auto load_b = [&](const int set, int g) {
      const int cl = g / TAPS, tap = g - cl * TAPS;
      const size_t koff = (size_t)((c_begin + cl) * TAPS + tap) * BK;
#pragma unroll
      for (int r = 0; r < NT; ++r) rgP[set + r] = *reinterpret_cast<const float4*>(wbase + (size_t)32 * r * K + koff);
    };
    auto store_b = [&](const int set, int stage) {
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        const int row = (tid >> 3) + 32 * r;
        *reinterpret_cast<float4*>(bst + stage * B_STAGE + row * B3R + ((c8 ^ ((row >> 1) & 7)) << 4)) = rgP[set + r];
      }
    };
    if (tid < 8) {   // the zero pixel
      *reinterpret_cast<float4*>(patch + NP * B3R + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_patch(c_begin);
    load_b(B0, 0);
    store_patch();
    store_b(B0, 0);
    if (c_begin + 1 < c_end) load_patch(c_begin + 1);
    if (1 < G) load_b(B0, 1);
    if (2 < G) load_b(B1, 2);
    __syncthreads();
    int g = 0, chunk = c_begin;
#define OFFK_PATCH_PRODUCER_STEP(RG)                                                                    \
    if (g >= G) break;                                                                                  \
    if (g + 1 < G) store_b(RG, (g + 1) & 1);          /* weight tile g+1 while step g is multiplied */  \
    if (g + 3 < G) load_b(RG, g + 3);                                                                   \
    __syncthreads();                                  /* end of step g */                               \
    ++g;                                                                                                \
    if (g < G && g % TAPS == 0) {                     /* next step starts a chunk: swap the patch */     \
      ++chunk;                                                                                          \
      store_patch();                                                                                    \
      if (chunk + 1 < c_end) load_patch(chunk + 1);                                                     \
      __syncthreads();                                                                                  \
    }
    for (;;) {
      OFFK_PATCH_PRODUCER_STEP(B0)
      OFFK_PATCH_PRODUCER_STEP(B1)
    }
#undef OFFK_PATCH_PRODUCER_STEP
    return;
  }

  // ================================ consumers ================================
  const int r32 = lane & 31, h = lane >> 5;
  const int ct = NT == 4 ? wave : (wave & 1);
  int geo[4 * RT];   // per row tile: y0, x0, image base, A offset of the current tap (one array: separate ones go to scratch)
#define y0(i) geo[4 * (i)]
#define x0(i) geo[4 * (i) + 1]
#define ib(i) geo[4 * (i) + 2]
#define aoff(i) geo[4 * (i) + 3]
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int rt = NT == 4 ? i : (wave >> 1) + 2 * i;
    const int ml = rt * 32 + r32;
    const bool mv = rt < 7 && ml < 196 && (long long)mt * 196 + ml < p.M;
    const int img = ml / HOWO, rem = ml - img * HOWO, ho = rem / WO, wo = rem - ho * WO;
    y0(i) = mv ? ho * S - PAD : -100000;
    x0(i) = wo * S - PAD;
    ib(i) = img * HW;
    aoff(i) = 0;
  }
  f32x16 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int brow = ct * 32 + r32;
  const int bbase = brow * B3R + ((h ^ ((brow >> 1) & 7)) << 4);
  __syncthreads();
  int tap = 0;
  for (int g = 0; g < G; ++g) {
    const int kh = tap / KW, kw = tap - kh * KW;
    const char* bh_p = bst + (g & 1) * B_STAGE + bbase;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int y = y0(i) + kh, x = x0(i) + kw;
      const bool inb = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      const int xp = S == 2 ? (x & 1) * (W / 2) + (x >> 1) : x;
      const int slot = inb ? ib(i) + y * W + xp : NP;
      aoff(i) = slot * B3R + ((h ^ ((slot >> 1) & 7)) << 4);
    }
    {
      // chunk (2 q + h) ^ swz == ((h ^ swz) ^ 2 q): a lane reads four consecutive k of its row per visit
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(bh_p + ((bbase ^ (q << 5)) - bbase));
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          if (NT == 4 || (wave >> 1) + 2 * i < 7) {
            const float4 a = *reinterpret_cast<const float4*>(patch + (aoff(i) ^ (q << 5)));
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                   // end of step g
    if (++tap == TAPS) {
      tap = 0;
      if (g + 1 < G) __syncthreads();                  // the next chunk's patch is in place
    }
  }

#undef y0
#undef x0
#undef ib
#undef aoff
  // ---- epilogue (as conv_igemm_kernel) -------------------------------------------------------------
  const int co = n0 + ct * 32 + r32;
  if (p.splitk > 1) {
    float* part = p.partial + (size_t)zs * p.M * p.Co;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int rt = NT == 4 ? i : (wave >> 1) + 2 * i;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ml = rt * 32 + acc_row(reg, h);
        const long long m = (long long)mt * 196 + ml;
        if (rt < 7 && ml < 196 && m < p.M) part[(size_t)m * p.Co + co] = acc[i][reg];
      }
    }
    return;
  }
  const bool relu_pre = p.flags & OFFK_CONV_RELU_PRE_, relu_post = p.flags & OFFK_CONV_RELU_POST_;
  const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int rt = NT == 4 ? i : (wave >> 1) + 2 * i;
    float rv[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) rv[reg] = 0.f;
    if (p.res) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ml = rt * 32 + acc_row(reg, h);
        const long long m = (long long)mt * 196 + ml;
        const bool ok = rt < 7 && ml < 196 && m < p.M;
        rv[reg] = *(ok ? p.res + (size_t)m * p.res_cs + p.res_coff + co : p.res);
      }
    }
    float ov[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      float v = acc[i][reg] + bv;
      if (relu_pre) v = fmaxf(v, 0.f);
      v += rv[reg];
      if (relu_post) v = fmaxf(v, 0.f);
      asm volatile("" : "+v"(v));
      ov[reg] = v;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int ml = rt * 32 + acc_row(reg, h);
      const long long m = (long long)mt * 196 + ml;
      if (rt < 7 && ml < 196 && m < p.M) p.y[(size_t)m * p.y_cs + p.y_coff + co] = ov[reg];
    }
  }
}

// (Rounds 1-4 also had a half-chunk bf16x3 form of this kernel, conv_patch16_kernel / tile_cfg 10: retired with the two-plane
//  bf16x3 mode in round 5 -- git history.)

template <int KH, int S, int W, int NT, int PREC>
static hipError_t launch_patch(ConvArgs a, hipStream_t st) {
  constexpr int PAD = KH / 2, WO = (W + 2 * PAD - KH) / S + 1, IMG = 196 / (WO * WO), NP = IMG * W * W, BN = 32 * NT;
  constexpr size_t lds = (size_t)(NP + 1) * 128 + 2 * (size_t)(BN * 128);
  if (a.Co % BN) return hipErrorInvalidConfiguration;
  auto kern = conv_patch_kernel<KH, S, W, NT, PREC>;
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds);
    if (e != hipSuccess) return e;
  }
  a.gm = (a.M + 195) / 196;
  a.gn = a.Co / BN;
  hipLaunchKernelGGL(kern, dim3(a.gm * a.gn * a.splitk), dim3(512), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || a.splitk == 1) return e;
  size_t n4 = (size_t)a.M * (a.Co / 4);
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}
// shapes the patch kernel covers: (k, stride, W) = (7, 2, 28), (5, 2, 14), (3, 1, 14), (3, 1, 7), pad = k / 2, square maps
static hipError_t launch_patch_shape(const ConvArgs& a, int KH, int S, int W, int nt, int prec, hipStream_t st) {
#define OFFK_PATCH_CASE(k, s, w)                                                                                     \
  if (KH == k && S == s && W == w) {                                                                                 \
    if (prec != 0) return hipErrorInvalidValue;                                                                      \
    return nt == 4 ? launch_patch<k, s, w, 4, 0>(a, st) : launch_patch<k, s, w, 2, 0>(a, st);                       \
  }
  OFFK_PATCH_CASE(7, 2, 28)
  OFFK_PATCH_CASE(5, 2, 14)
  OFFK_PATCH_CASE(3, 1, 14)
  OFFK_PATCH_CASE(3, 1, 7)
#undef OFFK_PATCH_CASE
  return hipErrorInvalidConfiguration;
}

template <int KH, int KW, int S, int TM, int TN, int WM, int WN, int PREC>
static hipError_t launch_cfg(ConvArgs a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = PREC == 4 ? kDmaStages * (size_t)(BM + BN) * 128
                                   : 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  if (a.Co % BN) return hipErrorInvalidConfiguration;
  auto kern = conv_igemm_kernel<KH, KW, S, TM, TN, WM, WN, PREC>;
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds);
    if (e != hipSuccess) return e;
  }
  a.gm = (a.M + BM - 1) / BM;
  a.gn = a.Co / BN;
  a.spread = (long long)a.gm * a.gn * a.splitk * a.batch >= 1024 ? 1 : 0;      // >= four blocks per CU
  hipLaunchKernelGGL(kern, dim3(a.gm * a.gn * a.splitk, a.batch), dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || a.splitk == 1) return e;
  size_t n4 = (size_t)a.M * (a.Co / 4);
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

// tile configurations: block = 4 waves arranged WM x WN, wave tile = (32*TM) x (32*TN)
//   0: 128x128 (2x2 of 64x64)   1: 128x64 (4x1 of 32x64)   2: 256x64 (4x1 of 64x64)
//   3: 64x64   (2x2 of 32x32)   4: 64x128 (2x2 of 32x64)   5: 128x256 (2x2 of 64x128)
constexpr int kNumTileCfg = 6;
const int kTileBM[kNumTileCfg] = {128, 128, 256, 64, 64, 128};
const int kTileBN[kNumTileCfg] = {128, 64, 64, 64, 128, 256};

template <int KH, int KW, int S, int PREC>
static hipError_t launch_prec(const ConvArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_cfg<KH, KW, S, 2, 2, 2, 2, PREC>(a, st);
    case 1: return launch_cfg<KH, KW, S, 1, 2, 4, 1, PREC>(a, st);
    case 2: return launch_cfg<KH, KW, S, 2, 2, 4, 1, PREC>(a, st);
    case 3: return launch_cfg<KH, KW, S, 1, 1, 2, 2, PREC>(a, st);
    case 4: return launch_cfg<KH, KW, S, 1, 2, 2, 2, PREC>(a, st);
    case 5: return launch_cfg<KH, KW, S, 2, 4, 2, 2, PREC>(a, st);
    default: return hipErrorInvalidConfiguration;
  }
}
template <int KH, int KW, int S>
static hipError_t launch_shape(const ConvArgs& a, int cfg, int prec, hipStream_t st) {
  // the lean (buffer-addressed) loader when both tensors fit 31-bit byte offsets
  if (prec != 0) return hipErrorInvalidValue;      // (the two-plane bf16x3 core, PREC 1 / 3, is no longer instantiated: retired in round 5)
  // (LDS-DMA destinations above 64 KB work: tools/lds_dma_probe_hi.hip lands rows up to 139 KB into a 144 KB allocation)
  if (a.x_bytes && !(a.flags & OFFK_CONV_RELU_IN_) && !a.no_dma) return launch_prec<KH, KW, S, 4>(a, cfg, st);
  return a.x_bytes ? launch_prec<KH, KW, S, 2>(a, cfg, st) : launch_prec<KH, KW, S, 0>(a, cfg, st);
}

// Heuristic when the caller gives no plan, distilled from tools/tune_conv.py sweeps on MI355X:
// small tiles win (more resident waves hide the gather latency, finer work quantisation):
// 64x64 for short K, 64x128 for long K with wide outputs; then split K until the grid has
// >= 1536 blocks (6 per CU) while every slice keeps >= 16 K-tiles.
// Exact fp32 (round 3, tools/tune_forward.py --wide at B = 42 / P = 252: the rule above lost 5.8 % to the in-situ optimum,
// almost all of it in the split of the three big convs): the 64x64 tile everywhere and a split that brings the grid to
// >= 3500 blocks (2.7 rounds of the 1280 resident blocks: the tuned plans at P = 240 / 252 / 384 all sit at 3500-7000)
// while a slice keeps >= 24 K-tiles.
void conv2d_auto_plan(long long M, int Co, int nkt, int* cfg_out, int* splitk_out, int precision) {
  static const int cand[] = {1, 2, 3, 4, 6, 8, 12};
  if (precision == 0) {
    // 64x64 tile, split to >= 3500 blocks with >= 24 K-tiles per slice (in-situ sweeps at B = 42 / 64); while the grid does not
    // even cover the chip twice (small batches: the 7x7 conv of ONE clip is 19 tiles x 490 K-tiles) slices go down to 16 K-tiles
    // (tools/sweep_splitk_small.py: B = 1, 7x7 48 us at 32 slices, 42 at 12 - 24; 5x5 52 us at 64, 40 - 48 at 24 - 48)
    static const int cand32[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    const long long blocks = ((M + 63) / 64) * (Co / 64);
    int sk = 1;
    for (int c : cand32) {
      if (c > 1 && nkt / c < (blocks * sk < 512 ? 16 : 24)) break;
      sk = c;
      if (blocks * c >= 3500) break;
    }
    *cfg_out = 3;
    *splitk_out = sk;
    return;
  }
  int cfg = (nkt >= 36 && Co % 128 == 0) ? 4 : 3;
  long long blocks = ((M + kTileBM[cfg] - 1) / kTileBM[cfg]) * (Co / kTileBN[cfg]);
  int sk = 1;
  for (int c : cand) {
    if (nkt / c < 16) break;
    sk = c;
    if (blocks * c >= 1536) break;
  }
  *cfg_out = cfg;
  *splitk_out = sk;
}

hipError_t conv2d_launch(const ConvDesc& d, hipStream_t st, const char** why) {
  *why = nullptr;
  if (d.precision != 0) { *why = "conv2d: precision must be 0 (fp32; the bf16x3 mode was retired in ABI v9)"; return hipErrorInvalidValue; }
  const bool narrow = d.co_limit > 0 && d.co_limit < d.Co;   // padded-N GEMM with scalar stores (the FC heads)
  if (d.Ci % 32 || d.Co % 64 || d.x_cs % 4 || d.x_coff % 4 || d.y_cs <= 0 || (!narrow && (d.y_cs % 4 || d.y_coff % 4)) ||
      (d.res && (d.res_cs % 4 || d.res_coff % 4))) {
    *why = "conv2d: need Ci % 32 == 0, Co % 64 == 0, 16-byte aligned channel slices";
    return hipErrorInvalidValue;
  }
  ConvArgs a;
  a.x = d.x; a.x_cs = d.x_cs; a.x_coff = d.x_coff;
  a.n_img = d.n_img; a.H = d.H; a.W = d.W; a.Ci = d.Ci; a.pad = d.pad;
  a.Ho = (d.H + 2 * d.pad - d.KH) / d.stride + 1;
  a.Wo = (d.W + 2 * d.pad - d.KW) / d.stride + 1;
  a.w = d.w; a.bias = d.bias; a.Co = d.Co;
  a.res = d.res; a.res_cs = d.res_cs; a.res_coff = d.res_coff;
  a.y = d.y; a.y_cs = d.y_cs; a.y_coff = d.y_coff;
  a.flags = d.flags;
  {
    // PREC 2 needs 31-bit byte offsets: input bytes behind x + x_coff (+ the descriptor bias), weight bytes
    const unsigned long long xb = ((unsigned long long)d.n_img * d.H * d.W * d.x_cs - d.x_coff) * 4ull;
    const unsigned long long bias = (unsigned long long)(d.pad * d.W + d.pad) * d.x_cs * 4ull;
    const unsigned long long wb = (unsigned long long)d.Co * d.KH * d.KW * d.Ci * 4ull;
    const bool fits = xb + bias < 0x7fffffffull && wb < 0x7fffffffull;
    a.x_bytes = fits ? (unsigned)xb : 0u;
    a.w_bytes = fits ? (unsigned)wb : 0u;
    a.no_dma = 0;
#ifdef OFFK_NO_CONV_DMA      // A/B builds (tools)
    a.no_dma = 1;
#endif
#ifdef OFFK_TUNING_KNOBS
    { const char* e = getenv("OFFK_CONV_DMA"); if (e && *e == '0') a.no_dma = 1; }
    { const char* e = getenv("OFFK_CONV_LEAN"); if (e && !((atoi(e) >> (d.precision & 1)) & 1)) a.x_bytes = 0; }   // bit 0 fp32, bit 1 bf16x3
#endif
  }
  long long M = (long long)d.n_img * a.Ho * a.Wo;
  if (M <= 0 || M > 0x7fffffffLL - 256) { *why = "conv2d: bad problem size"; return hipErrorInvalidValue; }
  a.M = (int)M;
  const int nkt = d.KH * d.KW * (d.Ci / 32);
  int cfg = d.tile_cfg, sk = d.splitk;
  if (cfg < 0 || sk < 1) conv2d_auto_plan(M, d.Co, nkt, &cfg, &sk, d.precision);
  if (sk > nkt) sk = nkt;
  if (narrow) sk = 1;
  a.co_limit = narrow ? d.co_limit : d.Co;
  a.pool_part = d.pool_part; a.pool_hw = d.pool_hw;
  a.batch = d.batch > 1 ? d.batch : 1; a.x_bs = d.x_bstride; a.w_bs = d.w_bstride; a.y_bs = d.y_bstride;
  a.ngroups = d.ngroups;
  for (int g = 0; g < 4; ++g) { a.g_batch[g] = d.g_batch[g]; a.g_Ci[g] = d.g_Ci[g]; a.g_x[g] = d.g_x[g]; a.g_w[g] = d.g_w[g]; a.g_y[g] = d.g_y[g]; }
  if (d.ngroups > 0) {
    if (d.ngroups > 4 || d.x_coff || d.y_coff || d.y_cs != d.Co) { *why = "conv2d: grouped batches are dense [M][Ci] x [Co][Ci] -> [M][Co] problems"; return hipErrorInvalidValue; }
    int nb = 0;
    for (int g = 0; g < d.ngroups; ++g) {
      if (d.g_Ci[g] < 32 || d.g_Ci[g] % 32 || d.g_batch[g] < 1) { *why = "conv2d: bad batch group"; return hipErrorInvalidValue; }
      nb += d.g_batch[g];
    }
    a.batch = nb;
  }
  if (a.batch > 1) {
    if (d.KH != 1 || d.KW != 1 || d.res || d.pool_part || cfg == 6 || cfg == 7 || cfg == 10 || a.batch > 65535) { *why = "conv2d: batched launches are plain 1x1 GEMMs on a generic tile"; return hipErrorInvalidValue; }
    sk = 1;
  }
  if (d.pool_part) {
    if (d.pool_hw < 32 || cfg == 6 || cfg == 7 || cfg == 10) { *why = "conv2d: pooled partial sums need >= 32 rows per image and a generic tile"; return hipErrorInvalidValue; }
    sk = 1;                               // the sums are taken in the epilogue of an unsplit conv
  }
  if (sk > 1 && !d.partial) sk = 1;                                                          // no slab space: unsplit
  while (sk > 1 && d.partial_floats < (size_t)sk * (size_t)M * d.Co) --sk;                   // ... or as many slices as fit
  a.splitk = sk; a.partial = d.partial; a.gm = a.gn = 0;
#ifdef OFFK_TUNING_KNOBS
  { const char* e = getenv("OFFK_CONV_ABLATE"); a.ablate = e ? atoi(e) : 0; }
#endif
#ifdef OFFK_CONV_TIMING
  {
    static unsigned long long* dbg = nullptr;
    if (!dbg) { (void)hipMalloc(reinterpret_cast<void**>(&dbg), 128); (void)hipMemset(dbg, 0, 128); }
    a.dbg = dbg;
    if (getenv("OFFK_CONV_TIMING_DUMP")) {
      unsigned long long h[16];
      (void)hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
      fprintf(stderr, "[conv timing] consumer mma %llu barrier %llu ktiles %llu | producer store %llu load-issue %llu barrier %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
      if (h[11])
        fprintf(stderr, "[conv timing, LDS-DMA form, wave 0, cycles per block] prologue %.0f (setup %.0f, first tile %.0f)  K loop %.0f (%.1f K-tiles: %.0f per tile)  epilogue %.0f (to the last store's issue %.0f, drain %.0f) | blocks %llu\n",
                (double)h[8] / h[11], (double)h[13] / h[11], (double)(h[8] - h[13]) / h[11], (double)h[9] / h[11], (double)h[12] / h[11], (double)h[9] / (double)h[12],
                (double)h[10] / h[11], (double)h[14] / h[11], (double)(h[10] - h[14]) / h[11], h[11]);
      (void)hipMemset(dbg, 0, 128);
    }
  }
#endif
  const int key = d.KH * 100 + d.KW * 10 + d.stride;
  hipError_t e;
  if (cfg == 10) { *why = "conv2d: tile_cfg 10 (the half-chunk patch kernel) existed for the bf16x3 mode only, retired in ABI v9"; return hipErrorInvalidValue; }
  if (cfg == 6 || cfg == 7) {   // patch kernel: square map, pad = k / 2, one of its four shapes (both precisions)
    if (narrow || d.KH != d.KW || d.H != d.W || d.pad != d.KH / 2) {
      *why = "conv2d: the patch kernel (tile_cfg 6 / 7) needs a square map and pad = k / 2";
      return hipErrorInvalidValue;
    }
    if (sk > d.Ci / 32) sk = d.Ci / 32;   // it splits over channel chunks
    a.splitk = sk;
    e = launch_patch_shape(a, d.KH, d.stride, d.H, cfg == 6 ? 4 : 2, d.precision, st);
    if (e == hipErrorInvalidConfiguration) *why = "conv2d: the patch kernel covers 7x7s2@28, 5x5s2@14, 3x3s1@14, 3x3s1@7 with Co % 128 (cfg 6) / % 64 (cfg 7)";
    return e;
  }
  switch (key) {
    case 111: e = launch_shape<1, 1, 1>(a, cfg, d.precision, st); break;
    case 331: e = launch_shape<3, 3, 1>(a, cfg, d.precision, st); break;
    case 552: e = launch_shape<5, 5, 2>(a, cfg, d.precision, st); break;
    case 772: e = launch_shape<7, 7, 2>(a, cfg, d.precision, st); break;
    default: *why = "conv2d: unsupported kernel/stride (have 1x1s1, 3x3s1, 5x5s2, 7x7s2)"; return hipErrorInvalidValue;
  }
  if (e == hipErrorInvalidConfiguration) *why = "conv2d: tile configuration does not divide Co";
  return e;
}

// [Co][Ci][KH][KW] (PyTorch) -> [Co][Ci/32][KH*KW][32]: the K order of conv_igemm_kernel
__global__ void pack_oihw_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int KHW) {
  size_t n = (size_t)Co * Ci * KHW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int cl = (int)(i & 31);
    size_t t = i >> 5;
    int tap = (int)(t % KHW);
    t /= KHW;
    int chunk = (int)(t % (Ci >> 5));
    int co = (int)(t / (Ci >> 5));
    dst[i] = src[((size_t)co * Ci + chunk * 32 + cl) * KHW + tap];
  }
}

hipError_t pack_conv_weight_launch(const float* src, int Co, int Ci, int KH, int KW, float* dst, hipStream_t st) {
  size_t n = (size_t)Co * Ci * KH * KW;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_oihw_kernel, dim3(blocks), dim3(256), 0, st, src, dst, Co, Ci, KH * KW);
  return hipGetLastError();
}

}  // namespace offk
