// Training side of the OFF units: backward of
//   G = relu(motion_conv_gen_s(X)); T = G[t+1] - G[t]            RGB_OFF.py:597-604
//   D = motion_spatial_down_s(X[:P]); S = dropout(dw3x3(D))       :609-612
//   motion_s = cat(S, T)                                          :616
// for the parameters train_off.py:39-45 leaves trainable (every `*motion*` tensor).  The feature
// maps come from the frozen backbone and get no gradient, so the backward is
//   K2b  units_bwd_kernel   dM -> dGpre [N*HW][128] (temporal-difference transpose, ReLU mask from the
//                           saved G) and dD [P*HW][32] (transposed depthwise 3x3 of dS = dM_S * dropout
//                           multiplier), plus per-block partial sums of the depthwise weight / bias
//                           gradients.  Same two block roles as the forward K2 (sobel_tdiff.hip); HBM-bound.
//   K1b  pw_wgrad_kernel    dW[160][C] = [dGpre | dD]^T . X as a split-K MFMA GEMM over the (frame, pixel)
//                           axis, all nine sites in one grouped launch; per-chunk slabs + bias partials.
//   reduce kernels          slabs / partials summed in a fixed order into the caller's gradient buffer
//                           in the reference's parameter layouts ([128,C,1,1], [32,C,1,1], [32,1,3,3]).
// Every sum has a fixed order: the gradients are bit-reproducible run to run.
#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

unsigned long long drop_stream_base(unsigned long long seed, int site) {
  const unsigned long long stream = ((seed & 0xFFFFFFFFFFFFull) << 8) | (unsigned long long)site;
  return mix64(stream * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull);
}

namespace {

constexpr int UB_THREADS = 256;
constexpr int UB_STAGE_MAX = 9;   // as ST_STAGE_MAX / ST_OUT_MAX in sobel_tdiff.hip
constexpr int UB_OUT_MAX = 7;
constexpr int UB_TGROUP = 7;      // frames per temporal step (L = 7 -> one step)
// T-blocks: one (pixel, channel quad) task per thread (8 pixels per block, offk_api.hip kUbTpix) and non-temporal
// accesses -- G, dT are read once and dG is next read by another kernel after a GB of other traffic.  The same recipe
// took K2 (sobel_tdiff.hip) from 270 to 227 us.
typedef float ubf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ub_ldnt(const float* q) {
  const ubf4 v = __builtin_nontemporal_load(reinterpret_cast<const ubf4*>(q));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ub_stnt(float* q, float4 v) {
  const ubf4 u = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(u, reinterpret_cast<ubf4*>(q));
}

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 fma4(float4 w, float4 x, float4 acc) {
  return make_float4(fmaf(w.x, x.x, acc.x), fmaf(w.y, x.y, acc.y), fmaf(w.z, x.z, acc.z), fmaf(w.w, x.w, acc.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}

}  // namespace

// =====================================================================================================
// K2b
// =====================================================================================================
__global__ __launch_bounds__(UB_THREADS, 2) void units_bwd_kernel(UbParams p) {   // two LDS tiles: 2 blocks per CU anyway
  extern __shared__ __attribute__((aligned(16))) float tile[];   // dS tile | D tile | taps [9][32] | wave partials [4][10][32]

  const unsigned bid = blockIdx.x, total = (unsigned)(p.total_s + p.total_t);
  const unsigned t_before = (unsigned)((unsigned long long)bid * p.total_t / total);
  const unsigned t_after = (unsigned)((unsigned long long)(bid + 1) * p.total_t / total);
  const bool is_t = t_after > t_before;
  const int ridx = is_t ? (int)t_before : (int)(bid - t_before);

  UbSite S;
#define OFFK_UB_PICK(i)                                                                                        \
  S.G = p.s[i].G; S.D = p.s[i].D; S.dw = p.s[i].dw; S.gm = p.s[i].gm; S.gm_cs = p.s[i].gm_cs;                    \
  S.dw_ref = p.s[i].dw_ref;                                                                                      \
  S.gm_coff = p.s[i].gm_coff; S.dG = p.s[i].dG; S.dD = p.s[i].dD; S.dw_part = p.s[i].dw_part;                    \
  S.drop_base = p.s[i].drop_base; S.H = p.s[i].H; S.strips = p.s[i].strips; S.rows = p.s[i].rows;                \
  S.tchunks = p.s[i].tchunks; S.s_begin = p.s[i].s_begin; S.t_begin = p.s[i].t_begin;
  OFFK_UB_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && ridx >= (is_t ? p.s[i].t_begin : p.s[i].s_begin)) { OFFK_UB_PICK(i) }
#undef OFFK_UB_PICK
  const int H = S.H, W = S.H, HW = H * H;
  const int L = p.L, T = L - 1;
  const int tid = threadIdx.x;

  if (is_t) {
    // ---- temporal branch: dGpre[b,t] = (dT[b,t-1] - dT[b,t]) * (G[b,t] > 0) -------------------------
    const int local = ridx - S.t_begin;
    const int b = local / S.tchunks, chunk = local - b * S.tchunks;
    const int q0 = chunk * p.tpix, npix = min(p.tpix, HW - q0);
    const size_t f0 = (size_t)b * L, p0 = (size_t)b * T;
    const size_t gstride = (size_t)HW * kGenCh, mstride = (size_t)HW * S.gm_cs;
    for (int task = tid; task < npix * 32; task += UB_THREADS) {
      const int q = q0 + (task >> 5), c4 = (task & 31) * 4;
      const float* g = S.G + (f0 * HW + q) * kGenCh + c4;
      float* dg = S.dG + (f0 * HW + q) * kGenCh + c4;
      const float* m = S.gm + (p0 * HW + q) * S.gm_cs + S.gm_coff + kDownCh + c4;
      float4 prev = zero4();
      for (int t0 = 0; t0 < L; t0 += UB_TGROUP) {
        float4 gv[UB_TGROUP], dt[UB_TGROUP];
#pragma unroll
        for (int j = 0; j < UB_TGROUP; ++j) {   // branch-free (see the S-blocks): frames past the clip read the zero page
          gv[j] = ub_ldnt(t0 + j < L ? g + (size_t)(t0 + j) * gstride : p.zeros);
          dt[j] = ub_ldnt(t0 + j < T ? m + (size_t)(t0 + j) * mstride : p.zeros);
        }
#pragma unroll
        for (int j = 0; j < UB_TGROUP; ++j)
          if (t0 + j < L) {
            float4 v = sub4(j ? dt[j - 1] : prev, dt[j]);
            v.x = gv[j].x > 0.f ? v.x : 0.f; v.y = gv[j].y > 0.f ? v.y : 0.f;
            v.z = gv[j].z > 0.f ? v.z : 0.f; v.w = gv[j].w > 0.f ? v.w : 0.f;
            ub_stnt(dg + (size_t)(t0 + j) * gstride, v);
          }
        prev = dt[UB_TGROUP - 1];
      }
    }
    return;
  }

  // ---- spatial branch --------------------------------------------------------------------------------
  const int local = ridx - S.s_begin;
  const int pr = local / S.strips, strip = local - pr * S.strips;
  const int y0 = strip * S.rows;
  const int R = min(S.rows, H - y0);
  const int npix = R * W, q0 = y0 * W;
  const int cq = tid & 7, cq4 = cq * 4;
  const int TW = W + 2, nstage = (R + 2) * TW;
  const int tile_floats = (S.rows + 2) * TW * kDownCh;
  const bool wgrad = S.dw_part != nullptr;
  float* tg = tile;                       // dS = dM_S * dropout multiplier, zero halo
  float* td = tile + tile_floats;         // D, zero halo (only read for the weight gradient)
  float* wl = tile + 2 * tile_floats;     // taps [9][32]
  float* wp = wl + 9 * kDownCh;           // wave partials [4][10][32]
  const float* gm = S.gm + (size_t)pr * HW * S.gm_cs + S.gm_coff + cq4;
  const float* d = S.D + (size_t)pr * HW * kDownCh + cq4;
  // every load of the block in flight before the first wait (branch-free: halo pieces read the zero page)
  float wreg[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = tid + UB_THREADS * r;
    wreg[r] = *(i < 9 * kDownCh ? S.dw + (S.dw_ref ? (i & 31) * 9 + (i >> 5) : i) : p.zeros);   // dw_ref: bound [32][1][3][3] parameter
  }
  float4 sgd[2 * UB_STAGE_MAX];   // dS pieces, then D pieces (one array: two arrays end up in scratch)
  float4* const sg = sgd;
  float4* const sd = sgd + UB_STAGE_MAX;
#pragma unroll
  for (int j = 0; j < UB_STAGE_MAX; ++j) {
    const int tp = (tid >> 3) + 32 * j;
    const int ty = tp / TW, tx = tp - ty * TW;
    const int y = y0 - 1 + ty, x = tx - 1;
    const bool inside = tp < nstage && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int px = y * W + x;
    sg[j] = *reinterpret_cast<const float4*>(inside ? gm + (size_t)px * S.gm_cs : p.zeros);
    sd[j] = *reinterpret_cast<const float4*>((inside && wgrad) ? d + (size_t)px * kDownCh : p.zeros);
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from sinking the D loads behind the first LDS stores
#pragma unroll
  for (int r = 0; r < 2; ++r)
    if (tid + UB_THREADS * r < 9 * kDownCh) wl[tid + UB_THREADS * r] = wreg[r];
#pragma unroll
  for (int j = 0; j < UB_STAGE_MAX; ++j) {
    const int tp = (tid >> 3) + 32 * j;
    if (tp < nstage) {
      if (p.drop_thresh) {
        const int ty = tp / TW, tx = tp - ty * TW;
        const int px = (y0 - 1 + ty) * W + (tx - 1);     // halo pieces are zero whatever the multiplier
        sg[j] = mul4(sg[j], drop_mul(S.drop_base, ((unsigned long long)pr * HW + (unsigned)px) * 8 + cq, p.drop_thresh, p.drop_scale));
      }
      *reinterpret_cast<float4*>(tg + tp * kDownCh + cq4) = sg[j];
      if (wgrad) *reinterpret_cast<float4*>(td + tp * kDownCh + cq4) = sd[j];
    }
  }
  __syncthreads();
  int coff[UB_OUT_MAX];
#pragma unroll
  for (int j = 0; j < UB_OUT_MAX; ++j) {
    const int px = (tid >> 3) + 32 * j;
    const int r = px / W, x = px - r * W;
    coff[j] = px < npix ? ((r + 1) * TW + (x + 1)) * kDownCh + cq4 : -1;
  }
  // dD[y][x] = sum_taps w[dy][dx] * dS[y - (dy-1)][x - (dx-1)]   (S[y'][x'] = sum w[dy][dx] * D[y'+dy-1][x'+dx-1])
  {
    float4 acc[UB_OUT_MAX];
#pragma unroll
    for (int j = 0; j < UB_OUT_MAX; ++j) acc[j] = zero4();
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 w4 = *reinterpret_cast<const float4*>(wl + (dy * 3 + dx) * kDownCh + cq4);
        const int toff = ((1 - dy) * TW + (1 - dx)) * kDownCh;
#pragma unroll
        for (int j = 0; j < UB_OUT_MAX; ++j)
          if (coff[j] >= 0) acc[j] = fma4(w4, *reinterpret_cast<const float4*>(tg + coff[j] + toff), acc[j]);
      }
    float* drow = S.dD + (size_t)pr * HW * kDownCh + cq4;
#pragma unroll
    for (int j = 0; j < UB_OUT_MAX; ++j)
      if (coff[j] >= 0) *reinterpret_cast<float4*>(drow + (size_t)(q0 + (tid >> 3) + 32 * j) * kDownCh) = acc[j];
  }
  if (!wgrad) return;
  // depthwise weight / bias gradient of this strip: dW[c][dy][dx] += dS[y][x] * D[y+dy-1][x+dx-1], db[c] += dS[y][x]
  float4 ws[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) ws[k] = zero4();
#pragma unroll
  for (int j = 0; j < UB_OUT_MAX; ++j)
    if (coff[j] >= 0) {
      const float4 gs = *reinterpret_cast<const float4*>(tg + coff[j]);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          ws[dy * 3 + dx] = fma4(gs, *reinterpret_cast<const float4*>(td + coff[j] + ((dy - 1) * TW + (dx - 1)) * kDownCh), ws[dy * 3 + dx]);
      ws[9] = add4(ws[9], gs);
    }
  // the 8 pixel-threads of a wave that share a channel quad (lane bits 3..5), then the 4 waves in order
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    ws[k] = add4(ws[k], shfl_xor4(ws[k], 8));
    ws[k] = add4(ws[k], shfl_xor4(ws[k], 16));
    ws[k] = add4(ws[k], shfl_xor4(ws[k], 32));
    if (lane < 8) *reinterpret_cast<float4*>(wp + (wave * 10 + k) * kDownCh + 4 * lane) = ws[k];
  }
  __syncthreads();
  float* out = S.dw_part + (size_t)local * 10 * kDownCh;
  for (int i = tid; i < 10 * kDownCh; i += UB_THREADS)
    out[i] = ((wp[i] + wp[10 * kDownCh + i]) + wp[20 * kDownCh + i]) + wp[30 * kDownCh + i];
}

hipError_t units_bwd_launch(const UbParams& p, hipStream_t st) {
  if (p.total_s + p.total_t <= 0) return hipSuccess;
  size_t tile_px = 0;
  for (int i = 0; i < p.nsites; ++i) {
    const size_t px = (size_t)(p.s[i].rows + 2) * (p.s[i].H + 2);
    if (px > tile_px) tile_px = px;
    if (px > 32 * UB_STAGE_MAX || p.s[i].rows * p.s[i].H > 32 * UB_OUT_MAX) return hipErrorInvalidValue;
  }
  const size_t lds = (2 * tile_px + 9 + 40) * kDownCh * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(units_bwd_kernel), (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(units_bwd_kernel, dim3(p.total_s + p.total_t), dim3(UB_THREADS), lds, st, p);
  return hipGetLastError();
}

// =====================================================================================================
// K1b: weight gradient of the stacked 1x1 reduce convs
//   dW[m][c] = sum over (frame f, pixel q) A[(f,q)][m] * X[f][c][q],  A = [dGpre (128) | dD (32, row = down_row(f))]
// Tile 160 x 128 (all stacked outputs x one 128-channel slab of X), BK = 32 pixels of one frame; a block walks
// `kt_per_blk` consecutive (frame, 32-pixel) K-tiles and writes its partial tile to a slab.  X is NCHW, so the
// B operand ([c][pixel], pixels contiguous) is copied to LDS as it is; the channels-last A operand is
// transposed in registers (4 pixels x 4 channels per thread) into the [m][k] image the MFMA core wants.
// =====================================================================================================
namespace {
constexpr int WG_BM = 160, WG_BN = 128;

__device__ __forceinline__ int wg_down_row(int f, int L, int P, int slice_mode) {
  if (slice_mode == 0) return f < P ? f : -1;
  const int b = f / L, t = f - b * L;
  return t < L - 1 ? b * (L - 1) + t : -1;
}
}  // namespace

__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(WgParams p) {
  // fp32 MFMA, LDS [row][k] fp32 (stride LDS_K)
  constexpr int LDS_BYTES = (WG_BM + WG_BN) * LDS_K * 4;
  static_assert(LDS_BYTES >= (8 * kGenCh + 32 * kDownCh) * 4, "bias reduction reuses the tile memory");
  __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
  float* lds = reinterpret_cast<float*>(lds_raw);
  float* As = lds;                   // [160][LDS_K]: rows = stacked output channel, k = pixel
  float* Bs = lds + WG_BM * LDS_K;   // [128][LDS_K]: rows = input channel of this slab

  // XCD-aware order: consecutive logical blocks (the channel slabs of one K-chunk, which share the A operand)
  // land on the same XCD / L2
  unsigned bid = blockIdx.x;
  {
    const unsigned t8 = (unsigned)p.total_blocks & ~7u;
    if (bid < t8) bid = (bid & 7u) * (t8 >> 3) + (bid >> 3);
  }
  WgSite S;
#define OFFK_WG_PICK(i)                                                                                   \
  S.dG = p.s[i].dG; S.dD = p.s[i].dD; S.slab = p.s[i].slab; S.bpart = p.s[i].bpart; S.C = p.s[i].C;         \
  S.HW = p.s[i].HW; S.tpf = p.s[i].tpf; S.kt_total = p.s[i].kt_total; S.ntiles = p.s[i].ntiles;              \
  S.blk_begin = p.s[i].blk_begin; S.nparts = p.s[i].nparts;                                                 \
  S.xp[0] = p.s[i].xp[0]; S.xp[1] = p.s[i].xp[1]; S.xp[2] = p.s[i].xp[2]; S.xp[3] = p.s[i].xp[3];           \
  S.cp[0] = p.s[i].cp[0]; S.cp[1] = p.s[i].cp[1]; S.cp[2] = p.s[i].cp[2]; S.cp[3] = p.s[i].cp[3];
  OFFK_WG_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)bid >= p.s[i].blk_begin) { OFFK_WG_PICK(i) }
#undef OFFK_WG_PICK
  const int C = S.C, HW = S.HW;
  const int local = (int)bid - S.blk_begin;
  const int chunk = local / S.ntiles, nt = local - chunk * S.ntiles;
  const int kt0 = chunk * p.kt_per_blk, kt1 = min(kt0 + p.kt_per_blk, S.kt_total);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool vec = (HW & 3) == 0;

  // B loader: rows (tid>>3) + 32r of the slab, k quad tid&7; the four 32-channel groups may sit in different parts
  const float* xrow[4];   // block-uniform (SGPRs): first row of the 32-channel group inside its part
  int xfs[4];             // frame stride of the part, in floats
  bool xok[4];
  const int rowoff = (tid >> 3) * HW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cb = nt * WG_BN + 32 * r;
    const float* xb = S.xp[0]; int cpart = S.cp[0], kl = cb;
    if (S.nparts > 1 && kl >= S.cp[0]) {
      kl -= S.cp[0]; xb = S.xp[1]; cpart = S.cp[1];
      if (S.nparts > 2 && kl >= S.cp[1]) {
        kl -= S.cp[1]; xb = S.xp[2]; cpart = S.cp[2];
        if (S.nparts > 3 && kl >= S.cp[2]) { kl -= S.cp[2]; xb = S.xp[3]; cpart = S.cp[3]; }
      }
    }
    xok[r] = cb < C;
    xrow[r] = xb + (size_t)kl * HW;
    xfs[r] = cpart * HW;
  }
  const bool wave_on = nt * WG_BN + 32 * wave < C;

  const int pq = tid & 7, cq = tid >> 3;     // gen A loader: pixel quad (fastest: conflict-free LDS stores), channel quad
  const int dpx = tid >> 3, dcq = tid & 7;   // down A loader: pixel, channel quad
  float4 rg[9];                              // 0-3 gen pixels, 4 down, 5-8 X rows
  float4 bs_g = make_float4(0.f, 0.f, 0.f, 0.f), bs_d = make_float4(0.f, 0.f, 0.f, 0.f);

  int frame = kt0 / S.tpf, kin = (kt0 - frame * S.tpf) * BK;
  int kin_ld = kin;   // k offset of the tile currently held in rg[]
  // Branch-free loads: an out-of-range piece reads the handle's zero page instead of being skipped.  With
  // `if (ok) v = load` the compiler serialises the loads of one tile (s_waitcnt vmcnt(0) in front of every
  // load whose destination registers are also written on another control-flow path).
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 7x7 maps: rows are only 4-byte aligned
  auto load_tile = [&]() {
    const float* ga = S.dG + ((size_t)frame * HW + kin + 4 * pq) * kGenCh + 4 * cq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* q = (!(p.dbg & 2) && kin + 4 * pq + j < HW) ? ga + (size_t)j * kGenCh : p.zeros;
      rg[j] = *reinterpret_cast<const float4*>(q);
    }
    const int dr = wg_down_row(frame, p.L, p.P, p.slice_mode);
    {
      const float* q = (dr >= 0 && kin + dpx < HW) ? S.dD + ((size_t)dr * HW + kin + dpx) * kDownCh + 4 * dcq : p.zeros;
      rg[4] = *reinterpret_cast<const float4*>(q);
    }
    // X quad k..k+3 of the row; a quad that straddles the row end (HW % 4 != 0) is read from HW-4 and shifted
    // into place by store_tile
    const int k = kin + 4 * (tid & 7), kk = min(k, HW - 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* q = (xok[r] && k < HW && !(p.dbg & 1)) ? xrow[r] + (size_t)frame * xfs[r] + (rowoff + kk) : p.zeros;
      const f4u v = *reinterpret_cast<const f4u*>(q);
      rg[5 + r] = make_float4(v.x, v.y, v.z, v.w);
    }
    kin_ld = kin;
  };
  auto fix_x_tail = [&]() {   // HW % 4 != 0 only: the quad was loaded `sh` floats early
    const int k = kin_ld + 4 * (tid & 7);
    const int sh = k < HW ? k - min(k, HW - 4) : 0;
    if (sh) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 v = rg[5 + r];
        rg[5 + r] = sh == 1 ? make_float4(v.y, v.z, v.w, 0.f) : (sh == 2 ? make_float4(v.z, v.w, 0.f, 0.f) : make_float4(v.w, 0.f, 0.f, 0.f));
      }
    }
  };
  auto store_tile = [&]() {
    if (!vec) fix_x_tail();
    bs_g.x += (rg[0].x + rg[1].x) + (rg[2].x + rg[3].x); bs_g.y += (rg[0].y + rg[1].y) + (rg[2].y + rg[3].y);
    bs_g.z += (rg[0].z + rg[1].z) + (rg[2].z + rg[3].z); bs_g.w += (rg[0].w + rg[1].w) + (rg[2].w + rg[3].w);
    bs_d.x += rg[4].x; bs_d.y += rg[4].y; bs_d.z += rg[4].z; bs_d.w += rg[4].w;
    float* a = As + 4 * cq * LDS_K + 4 * pq;
    *reinterpret_cast<float4*>(a) = make_float4(rg[0].x, rg[1].x, rg[2].x, rg[3].x);
    *reinterpret_cast<float4*>(a + LDS_K) = make_float4(rg[0].y, rg[1].y, rg[2].y, rg[3].y);
    *reinterpret_cast<float4*>(a + 2 * LDS_K) = make_float4(rg[0].z, rg[1].z, rg[2].z, rg[3].z);
    *reinterpret_cast<float4*>(a + 3 * LDS_K) = make_float4(rg[0].w, rg[1].w, rg[2].w, rg[3].w);
    float* dd = As + (kGenCh + 4 * dcq) * LDS_K + dpx;
    dd[0] = rg[4].x; dd[LDS_K] = rg[4].y; dd[2 * LDS_K] = rg[4].z; dd[3 * LDS_K] = rg[4].w;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * r) * LDS_K + 4 * (tid & 7)) = rg[5 + r];
  };

  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (kt0 < kt1) load_tile();
  for (int kt = kt0; kt < kt1; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < kt1) {
      kin += BK;
      if (kin >= HW) { kin = 0; ++frame; }
      load_tile();
    }
    if (wave_on) {
      const int r32 = lane & 31, h = lane >> 5;
      const float* bsrc = Bs + (wave * 32 + r32) * LDS_K + 4 * h;
      const float* asrc = As + r32 * LDS_K + 4 * h;
#pragma unroll 1
      for (int g = 0; g < BK / 8; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bsrc + 8 * g);
        float4 a[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) a[t] = *reinterpret_cast<const float4*>(asrc + t * 32 * LDS_K + 8 * g);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b.w, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: partial tile -> slab [chunk][160][ntiles*128] -----------------------------------------
  const int cpad = S.ntiles * WG_BN;
  if (wave_on) {
    const int r32 = lane & 31, h = lane >> 5;
    float* out = S.slab + (size_t)chunk * WG_BM * cpad + nt * WG_BN + wave * 32 + r32;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) out[(size_t)(t * 32 + acc_row(reg, h)) * cpad] = acc[t][reg];
  }
  // ---- bias partials (first channel slab only): sum of the A operand over this chunk's rows ---------------
  if (nt == 0) {
    float* red = lds;   // [8][128] gen, then [32][32] down
    *reinterpret_cast<float4*>(red + pq * kGenCh + 4 * cq) = bs_g;
    *reinterpret_cast<float4*>(red + 8 * kGenCh + dpx * kDownCh + 4 * dcq) = bs_d;
    __syncthreads();
    if (tid < kGenCh) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += red[i * kGenCh + tid];
      S.bpart[(size_t)chunk * WG_BM + tid] = s;
    } else if (tid < WG_BM) {
      float s = 0.f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) s += red[8 * kGenCh + i * kDownCh + (tid - kGenCh)];
      S.bpart[(size_t)chunk * WG_BM + tid] = s;
    }
  }
}

hipError_t pw_wgrad_launch(const WgParams& p, hipStream_t st) {
  if (p.total_blocks <= 0) return hipSuccess;
  if (p.precision != 0) return hipErrorInvalidValue;      // (the bf16x3 core is no longer instantiated: retired in round 5)
  hipLaunchKernelGGL(pw_wgrad_kernel, dim3(p.total_blocks), dim3(256), 0, st, p);
  return hipGetLastError();
}

// =====================================================================================================
// reductions into the caller's gradient buffer (reference parameter layouts), fixed summation order
// =====================================================================================================
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WrParams p) {
  WrSite S;
#define OFFK_WR_PICK(i)                                                                                        \
  S.slab = p.s[i].slab; S.bpart = p.s[i].bpart; S.dw_part = p.s[i].dw_part; S.gen_w = p.s[i].gen_w;              \
  S.gen_b = p.s[i].gen_b; S.down_w = p.s[i].down_w; S.down_b = p.s[i].down_b; S.dw_w = p.s[i].dw_w;              \
  S.dw_b = p.s[i].dw_b; S.C = p.s[i].C; S.cpad = p.s[i].cpad; S.nchunks = p.s[i].nchunks; S.nsblocks = p.s[i].nsblocks;
  OFFK_WR_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && (int)blockIdx.y == i) { OFFK_WR_PICK(i) }
#undef OFFK_WR_PICK
  const int tid = threadIdx.x;
  const int C = S.C;
  const int nq = kUnitCh * (C >> 2);              // float4 outputs of the two weight matrices
  const int wblocks = (nq + 255) / 256;
  if ((int)blockIdx.x < wblocks) {
    const int i = blockIdx.x * 256 + tid;
    if (i >= nq) return;
    const int m = i / (C >> 2), c4 = (i - m * (C >> 2)) * 4;
    const float* src = S.slab + (size_t)m * S.cpad + c4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S.nchunks; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * kUnitCh * S.cpad);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* dst = m < kGenCh ? S.gen_w + (size_t)m * C + c4 : S.down_w + (size_t)(m - kGenCh) * C + c4;
    if (p.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dst);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(dst) = s;
    return;
  }
  const int role = blockIdx.x - wblocks;          // 0: the two biases, 1..10: depthwise tap rows / bias
  if (role == 0) {
    if (tid < kUnitCh) {
      // four interleaved accumulators (independent load chains), combined in a fixed order
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = 0;
      for (; k + 3 < S.nchunks; k += 4) {
        a0 += S.bpart[(size_t)k * kUnitCh + tid];
        a1 += S.bpart[(size_t)(k + 1) * kUnitCh + tid];
        a2 += S.bpart[(size_t)(k + 2) * kUnitCh + tid];
        a3 += S.bpart[(size_t)(k + 3) * kUnitCh + tid];
      }
      for (; k < S.nchunks; ++k) a0 += S.bpart[(size_t)k * kUnitCh + tid];
      const float s = ((a0 + a1) + a2) + a3;
      float* dst = tid < kGenCh ? S.gen_b + tid : S.down_b + (tid - kGenCh);
      *dst = (p.accumulate ? *dst : 0.f) + s;
    }
    return;
  }
  if (role > 10 || !S.dw_part) return;
  // 32 channels x 8 interleaved slices of the S-block partials, slices then summed in order
  __shared__ float red[8][kDownCh];
  const int j = role - 1, c = tid & 31, sl = tid >> 5;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // independent load chains, fixed combine order
  int k = sl;
  for (; k + 24 < S.nsblocks; k += 32) {
    a0 += S.dw_part[((size_t)k * 10 + j) * kDownCh + c];
    a1 += S.dw_part[((size_t)(k + 8) * 10 + j) * kDownCh + c];
    a2 += S.dw_part[((size_t)(k + 16) * 10 + j) * kDownCh + c];
    a3 += S.dw_part[((size_t)(k + 24) * 10 + j) * kDownCh + c];
  }
  for (; k < S.nsblocks; k += 8) a0 += S.dw_part[((size_t)k * 10 + j) * kDownCh + c];
  red[sl][c] = ((a0 + a1) + a2) + a3;
  __syncthreads();
  if (tid < kDownCh) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][tid];
    float* dst = j < 9 ? S.dw_w + tid * 9 + j : S.dw_b + tid;      // [32][1][3][3] / [32]
    *dst = (p.accumulate ? *dst : 0.f) + t;
  }
}

hipError_t wgrad_reduce_launch(const WrParams& p, hipStream_t st) {
  int maxc = 0;
  for (int i = 0; i < p.nsites; ++i) maxc = p.s[i].C > maxc ? p.s[i].C : maxc;
  const int wblocks = (kUnitCh * (maxc >> 2) + 255) / 256;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + 11, p.nsites), dim3(256), 0, st, p);
  return hipGetLastError();
}

// consensus backward (basic_ops.py:29-33): grad_in[b*T + t][c] = grad_out[b][c] / T
__global__ void consensus_bwd_kernel(const float* __restrict__ go, int B, int T, int C, float* __restrict__ gi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T * C) return;
  const int c = i % C, b = i / (C * T);
  gi[i] = go[(size_t)b * C + c] / (float)T;
}
hipError_t consensus_bwd_launch(const float* go, int B, int T, int C, float* gi, hipStream_t st) {
  const int n = B * T * C;
  if (n <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(consensus_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, go, B, T, C, gi);
  return hipGetLastError();
}

}  // namespace offk
