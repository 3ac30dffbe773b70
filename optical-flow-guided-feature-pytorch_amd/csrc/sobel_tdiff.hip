// K2: spatial gradient + temporal difference + concat, the HBM-bound half of an OFF unit.
//
// Stands for, per tap site (reference RGB_OFF.py, site 3a lines):
//   T[b*(L-1)+t] = G[b*L+t+1] - G[b*L+t]                     :599-604 (view / slice / sub)
//   S = depthwise3x3(D) (+bias)                               :611   (RGB_OFF: learned)
//     = SobelFilter_Diagonal(D)                               Flow_OFF.py:622, util.py:52-77
//   motion_s = cat(S, T); fusion = cat(motion_*, carried)     :616, :656, :760, :832
// Neither concat exists as a copy: S and T are written straight into the channel slice [coff, coff + 32) / [coff + 32,
// coff + 160) of the site's fusion buffer (fusion_28 / fusion_14 / fusion_7, channels-last rows of 320 / 1056 / 832
// floats).  The kernel takes the two halves as separate destination views (StSite::Ms / Mt with their own stride and
// offset) -- in the library both alias the same fusion buffer (fill_st_site, offk_api.hip); a layout with dense per-site
// S / T buffers was built in round 2, measured and reverted.  dropout(p=0.8) (:612) is the
// identity in eval mode; in training mode (StParams.drop_thresh != 0, offk_off_units_train) the S half is
// multiplied by the reproducible keep-mask / (1 - p) of units_bwd.hip.
//
// Layout: everything channels-last.  G [N*HW][128], D [P*HW][32], S out [P*HW][s_cs], T out [P*HW][t_cs].
// All nine sites share ONE grouped launch made of two block roles, interleaved in block-id
// order (Bresenham) so every CU always holds a mix of both:
//   T-block = (site, clip, chunk of <= 64 pixels): temporal difference, pure streaming, no LDS.
//             A thread owns (pixel, 4 channels) and loads all L frames (6 per step) before it
//             stores, the previous frame stays in registers: every G element is read exactly
//             once with 16-B loads, 1 KiB contiguous per wave instruction -- algo 0.  algo 1
//             puts t on the lanes (8 t-slots x 8 channel quads) and takes the difference with
//             a wavefront shuffle (kept for the A/B measurement in DESIGN.md: it is slower).
//   S-block = (site, pair, strip of <= 7 rows): the D strip plus a one-pixel zero halo is
//             staged in LDS (all loads of a block in flight at once), then each thread
//             filters its <= 4 pixels for a fixed channel quad, taps outer, ds_read_b128.
//             One stage->barrier->filter round per block; the latency is hidden by the other
//             resident blocks (4 per CU) rather than by a per-block pipeline.
// Measured (profiles/r01): run back to back inside one block the two halves add up (temporal
// 5.9 TB/s + spatial 2.5 TB/s => 4.7 TB/s); as separate interleaved roles they overlap.
// Algorithmic HBM bytes per (clip, site): H*H*4*(128*L + 192*(L-1))  (SURVEY.md 8d).
#include <cstdlib>

#include "offk_common.h"
#include "offk_internal.h"

namespace offk {

constexpr int ST_THREADS = 256;
constexpr int ST_STAGE_MAX = 9;   // >= ceil((rows+2)*(W+2) / 32): staged tile pixels per thread
constexpr int ST_OUT_MAX = 7;     // >= ceil(rows*W / 32): output pixels per thread
constexpr int ST_TGROUP = 6;      // frames loaded per temporal step (L = 7 -> one step)

// Tuning knobs.  The product build uses the constants; a build with -DOFFK_TUNING_KNOBS (tools only) lets the
// environment override them for sweeps -- the shipped library never calls getenv on a launch path.
#ifdef OFFK_TUNING_KNOBS
static int knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
#else
static constexpr int knob(const char*, int dflt) { return dflt; }
#endif
int st_flat_rows() { return knob("OFFK_K2_FLATROWS", 64); }   // (pair, pixel) rows per T-block of the flat form
int st_tpix() { return knob("OFFK_K2_TPIX", 8); }    // pixels per T-block: 8 = one (pixel, channel quad) task per thread (sweep: tools/sweep_k2.py; 64 was 7 % slower)

void st_plan(int H, int* strips, int* rows) {
  // 28x28 planes: four 7-row strips; 14x14 and 7x7 planes: one whole-plane S-block per pair (OFFK_K2_ROWS14=7
  // restores two 7-row strips for 14x14)
  *rows = H >= 28 ? knob("OFFK_K2_ROWS28", 7) : (H > 7 ? knob("OFFK_K2_ROWS14", 14) : 7);
  if (*rows < 1 || (H >= 28 && *rows > 7) || *rows > 14) *rows = 7;
  *strips = (H + *rows - 1) / *rows;
}
// multiply-shift reciprocals for the S-blocks' n / W and n / (W + 2) (n < 512): checked exhaustively here
void st_recips(int H, int* wrecip, int* twrecip) {
  auto make = [](int d) {
    const int m = (65536 + d - 1) / d;
    for (int n = 0; n < 512; ++n)
      if (((n * m) >> 16) != n / d) return -1;
    return m;
  };
  *wrecip = make(H);
  *twrecip = make(H + 2);
}
int st_tpix_for(int H) { int t = st_tpix(); return t > 0 ? t : (H >= 28 ? 112 : H >= 14 ? 98 : 49); }
int st_tchunks(int H) { return (H * H + st_tpix_for(H) - 1) / st_tpix_for(H); }
int st_tchunks_flat(int H, int L) { return ((L - 1) * H * H + st_flat_rows() - 1) / st_flat_rows(); }

typedef float f4v __attribute__((ext_vector_type(4)));
// Non-temporal variants for the T-blocks (the S-blocks always use the default policy: their D halo rows are re-read by the
// neighbouring strip and nt made them slower).  Round 1, 64 pixels per T-block: no gain.  Round 2, one task per thread
// (tpix 8): nt loads + nt stores take the whole kernel from 270 to 256 us on the same box (profiles/r02/k2_sweep.txt);
// nt loads alone are slower (279), nt stores alone 262.  G is read exactly once and T is not re-read before K2 ends,
// so neither stream has any use for a cache line.
template <int NT>
__device__ __forceinline__ float4 ldg4(const float* p) {
  if (NT) {
    f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <int NT>
__device__ __forceinline__ void stg4(float* p, float4 v) {
  if (NT) {
    f4v u = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(u, reinterpret_cast<f4v*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 fma4(float4 w, float4 x, float4 acc) {
  return make_float4(fmaf(w.x, x.x, acc.x), fmaf(w.y, x.y, acc.y), fmaf(w.z, x.z, acc.z), fmaf(w.w, x.w, acc.w));
}

// ROLES only names the launch (3 = the whole K2, 1 = S-blocks alone -- what follows the fused units kernel --, 2 = T-blocks
// alone): a kernel trace (rocprofv3 --stats) then lists the full kernel apart from the partial launches.
template <int ALGO, int NTL, int NTS, int TAPS4, int ROLES>
__global__ __launch_bounds__(ST_THREADS, 7) void sobel_tdiff_kernel(StParams p) {
  static_assert(ROLES >= 1 && ROLES <= 3, "roles");
  extern __shared__ __attribute__((aligned(16))) f4v tile4[];   // [(rows+2)*(W+2) tile pixels][8 quads] + taps [9][8] + bias [8]

  // ---- block id -> (role, index within role): proportional interleave of the two roles ----
  const unsigned bid = blockIdx.x, total = (unsigned)(p.total_s + p.total_t);
  const unsigned t_before = (unsigned)((unsigned long long)bid * p.total_t / total);
  const unsigned t_after = (unsigned)((unsigned long long)(bid + 1) * p.total_t / total);
  const bool is_t = t_after > t_before;
  const int ridx = is_t ? (int)t_before : (int)(bid - t_before);

  // ---- role index -> site: field-wise scalar select chain (see pw_reduce.hip) ----
  StSite S;
#define OFFK_ST_PICK(i)                                                                                         \
  S.G = p.s[i].G; S.D = p.s[i].D; S.dw = p.s[i].dw; S.db = p.s[i].db; S.Ms = p.s[i].Ms; S.Mt = p.s[i].Mt;       \
  S.dw_ref = p.s[i].dw_ref;                                                                                     \
  S.H = p.s[i].H; S.s_cs = p.s[i].s_cs; S.s_coff = p.s[i].s_coff; S.t_cs = p.s[i].t_cs; S.t_coff = p.s[i].t_coff; \
  S.strips = p.s[i].strips; S.rows = p.s[i].rows; S.s_begin = p.s[i].s_begin; S.t_begin = p.s[i].t_begin;       \
  S.tchunks = p.s[i].tchunks; S.drop_base = p.s[i].drop_base; S.wrecip = p.s[i].wrecip; S.twrecip = p.s[i].twrecip;
  OFFK_ST_PICK(0)
#pragma unroll
  for (int i = 1; i < kNumSites; ++i)
    if (i < p.nsites && ridx >= (is_t ? p.s[i].t_begin : p.s[i].s_begin)) { OFFK_ST_PICK(i) }
#undef OFFK_ST_PICK
  const int H = S.H, W = S.H, HW = H * H;
  const int L = p.L, T = L - 1;
  const int tid = threadIdx.x;

  if (is_t) {
    // ---------------- temporal difference: Mt[.., t_coff .. t_coff+128) ----------------
    const int local = ridx - S.t_begin;
    const int b = local / S.tchunks, chunk = local - b * S.tchunks;
    const int tpix = p.tpix > 0 ? p.tpix : (H >= 28 ? 112 : H >= 14 ? 98 : 49);
    const int q0 = chunk * tpix, npix = min(tpix, HW - q0);
    const size_t f0 = (size_t)b * L, p0 = (size_t)b * T;
    const size_t gstride = (size_t)HW * kGenCh, mstride = (size_t)HW * S.t_cs;
    if (ALGO == 2) {
      // flat form: within a clip the pair rows (t, pixel) and the frame rows (t, pixel) are both pixel-major, so
      // T_row[p0*HW + r] = G_row[f0*HW + r + HW] - G_row[f0*HW + r] for r in [0, T*HW): one dense read stream, the same
      // stream HW rows later (the second read of a frame comes out of L2 / the Infinity Cache) and one write stream.
      // The blocks of a launch then walk the buffers in address order -- a compact in-flight window instead of the
      // L + T scattered streams per block of the rotation form.
      const int r0 = chunk * p.flat_rows, nrows = min(p.flat_rows, T * HW - r0);
      const float* g0 = S.G + (f0 * HW + r0) * kGenCh;
      float* m0 = S.Mt + (p0 * HW + r0) * S.t_cs + S.t_coff;
#pragma unroll 2
      for (int task = tid; task < nrows * 32; task += ST_THREADS) {
        const int r = task >> 5, c4 = (task & 31) * 4;
        const float4 a = ldg4<NTL>(g0 + (size_t)r * kGenCh + c4);
        const float4 bq = ldg4<NTL>(g0 + (size_t)r * kGenCh + gstride + c4);
        stg4<NTS>(m0 + (size_t)r * S.t_cs + c4, sub4(bq, a));
      }
      return;
    }
    if (ALGO != 1) {
      for (int task = tid; task < npix * 32; task += ST_THREADS) {
        const int q = q0 + (task >> 5), c4 = (task & 31) * 4;
        const float* g = S.G + (f0 * HW + q) * kGenCh + c4;
        float* m = S.Mt + (p0 * HW + q) * S.t_cs + S.t_coff + c4;
        float4 prev = ldg4<NTL>(g);
        for (int t0 = 1; t0 < L; t0 += ST_TGROUP) {
          float4 v[ST_TGROUP];
#pragma unroll
          for (int j = 0; j < ST_TGROUP; ++j)
            if (t0 + j < L) v[j] = ldg4<NTL>(g + (size_t)(t0 + j) * gstride);
#pragma unroll
          for (int j = 0; j < ST_TGROUP; ++j)
            if (t0 + j < L) stg4<NTS>(m + (size_t)(t0 + j - 1) * mstride, sub4(v[j], j ? v[j - 1] : prev));
          prev = v[ST_TGROUP - 1];
        }
      }
    } else {
      // lanes: slot = lane>>3 (t within a group of 8), cq = lane&7; a wave covers 8 t x 32 channels per
      // step, the next group of t overlaps by one so every pair has both ends in one wave.
      const int lane = tid & 63, wave = tid >> 6;
      const int slot = lane >> 3, cq = lane & 7;
      for (int unit = wave; unit < npix * 4; unit += ST_THREADS / 64) {
        const int q = q0 + (unit >> 2), c4 = ((unit & 3) * 8 + cq) * 4;
        for (int tb = 0; tb < T; tb += 7) {
          const int t = tb + slot;                       // frame index within the clip
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t < L) v = ldg4<NTL>(S.G + ((f0 + t) * HW + q) * kGenCh + c4);
          float4 nx;
          nx.x = __shfl_down(v.x, 8); nx.y = __shfl_down(v.y, 8);
          nx.z = __shfl_down(v.z, 8); nx.w = __shfl_down(v.w, 8);
          if (slot < 7 && t < T)
            stg4<NTS>(S.Mt + ((p0 + t) * HW + q) * S.t_cs + S.t_coff + c4, sub4(nx, v));
        }
      }
    }
    return;
  }

  // ---------------- spatial gradient: M[.., coff .. coff+32) ------------------------
  // LDS image in 16-byte units: tile4[(tile pixel)*8 + channel quad].  Indexing in f4v units (not floats) is what
  // lets the compiler prove the 16-byte alignment: the float-indexed form compiled to ds_read2_b32 pairs, whose
  // bank is (addr/4) mod 32 -- the four pixels of a 32-lane group (128 B apart) then collide 4-way
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 70 %).  As ds_read_b128 a 16-lane group reads two whole pixels =
  // one 256-byte bank row: conflict-free.
  const int local = ridx - S.s_begin;
  const int pr = local / S.strips, strip = local - pr * S.strips;   // pair, strip
  const int y0 = strip * S.rows;
  const int R = min(S.rows, H - y0);
  const int npix = R * W, q0 = y0 * W;
  const int cq = tid & 7, prow = tid >> 3;  // this thread's channel quad (fixed for the whole block) and pixel slot
  const int TW = W + 2, nstage = (R + 2) * TW;
  f4v* wl4 = tile4 + (S.rows + 2) * TW * 8;   // tap weights [9][8 quads] + bias [8 quads] behind the tile
  const float* d = S.D + (size_t)pr * HW * kDownCh;
  // stage rows y0-1 .. y0+R with the zero halo: this thread's tile pixels are prow + 32*j.  Every load of the block
  // (tap weights first) is issued before the first wait: halo / out-of-tile pieces read the handle's zero page
  // through a selected pointer -- with `if (inside) v = load` the compiler waited for the first loads before
  // issuing the rest and fetched the weights only after the tile had arrived (three latencies per block)
  // threads 0..71: one (tap, channel quad) of the depthwise weights; 72..79: the bias quads.  Library copy: [9][32]
  // tap-major.  A parameter bound in place (offk_bind_weight) keeps the reference layout [32][1][3][3]: channel stride 9.
  const bool has_w = tid < 72, has_b = S.db && tid >= 72 && tid < 80;
  const float* wq = has_w ? (S.dw_ref ? S.dw + 36 * (tid & 7) + (tid >> 3) : S.dw + 4 * tid) : (has_b ? S.db + 4 * (tid - 72) : p.zeros);
  const int ws_ = has_w ? (S.dw_ref ? 9 : 1) : (has_b ? 1 : 0);
  const float4 wreg = make_float4(wq[0], wq[ws_], wq[2 * ws_], wq[3 * ws_]);
  float4 st[ST_STAGE_MAX];
#pragma unroll
  for (int j = 0; j < ST_STAGE_MAX; ++j) {
    const int tp = prow + 32 * j;
    const int ty = (tp * S.twrecip) >> 16, tx = tp - ty * TW;        // tp / TW, tp % TW (exact for tp < 512, host-checked)
    const int y = y0 - 1 + ty, x = tx - 1;
    const bool inside = tp < nstage && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    st[j] = ldg4<0>(inside ? d + (size_t)(y * W + x) * kDownCh + 4 * cq : p.zeros);   // D halo rows are re-read by the neighbour strip: keep them cached
  }
  if (tid < 80) wl4[tid] = f4v{wreg.x, wreg.y, wreg.z, wreg.w};
#pragma unroll
  for (int j = 0; j < ST_STAGE_MAX; ++j) {
    const int tp = prow + 32 * j;
    if (tp < nstage) tile4[tp * 8 + cq] = f4v{st[j].x, st[j].y, st[j].z, st[j].w};
  }
  __syncthreads();
  // taps outer, this thread's (<= ST_OUT_MAX) output pixels inner: one weight quad live at a time.  Slots past the
  // strip read tile pixel 0 (valid LDS) and are masked at the store: the LDS reads stay unconditional.
  int base[ST_OUT_MAX];
  f4v acc[ST_OUT_MAX];
  const f4v b4 = wl4[72 + cq];
#pragma unroll
  for (int j = 0; j < ST_OUT_MAX; ++j) {
    const int px = prow + 32 * j;
    const int pp = px < npix ? px : 0;
    const int r = (pp * S.wrecip) >> 16, x = pp - r * W;           // pp / W, pp % W
    base[j] = ((r + 1) * TW + (x + 1)) * 8 + cq;
    acc[j] = b4;
  }
  if (TAPS4) {
    // diagonal Sobel (util.py:61): only (0,1) (1,0) (1,2) (2,1) are non-zero -- four taps instead of nine
    const int toffs[4] = {-TW * 8, -8, 8, TW * 8};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f4v w4 = wl4[(2 * k + 1) * 8 + cq];
#pragma unroll
      for (int j = 0; j < ST_OUT_MAX; ++j) acc[j] = w4 * tile4[base[j] + toffs[k]] + acc[j];
    }
  } else {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const f4v w4 = wl4[(dy * 3 + dx) * 8 + cq];
        const int toff = ((dy - 1) * TW + (dx - 1)) * 8;
#pragma unroll
        for (int j = 0; j < ST_OUT_MAX; ++j) acc[j] = w4 * tile4[base[j] + toff] + acc[j];
      }
  }
  float* mrow = S.Ms + (size_t)pr * HW * S.s_cs + S.s_coff + 4 * cq;
  if (p.drop_thresh) {   // training: nn.Dropout on the spatial gradient (:612)
#pragma unroll
    for (int j = 0; j < ST_OUT_MAX; ++j)
      if (prow + 32 * j < npix) {
        const float4 k4 = drop_mul(S.drop_base, ((unsigned long long)pr * HW + q0 + prow + 32 * j) * 8 + cq,
                                   p.drop_thresh, p.drop_scale);
        acc[j] = acc[j] * f4v{k4.x, k4.y, k4.z, k4.w};
      }
  }
#pragma unroll
  for (int j = 0; j < ST_OUT_MAX; ++j)
    if (prow + 32 * j < npix) stg4<0>(mrow + (size_t)(q0 + prow + 32 * j) * S.s_cs, make_float4(acc[j].x, acc[j].y, acc[j].z, acc[j].w));
}

hipError_t sobel_tdiff_launch(const StParams& p, int algo, hipStream_t st) {
  StParams q = p;
  q.tpix = st_tpix();
  q.flat_rows = st_flat_rows();
  if (algo == 2 || algo == 5) q.total_s = 0;   // diagnostic: temporal half only
  if (algo == 3) q.total_t = 0;                // diagnostic: spatial half only
  if (q.total_s + q.total_t <= 0) return hipSuccess;
  size_t tile_px = 0;
  for (int i = 0; i < p.nsites; ++i) {
    const size_t px = (size_t)(p.s[i].rows + 2) * (p.s[i].H + 2);
    if (px > tile_px) tile_px = px;
    if (px > 32 * ST_STAGE_MAX || p.s[i].rows * p.s[i].H > 32 * ST_OUT_MAX) return hipErrorInvalidValue;
    if (p.s[i].wrecip <= 0 || p.s[i].twrecip <= 0) return hipErrorInvalidValue;
  }
  size_t lds = (tile_px + 10) * kDownCh * sizeof(float);   // tile + taps + bias
  dim3 grid(q.total_s + q.total_t);
  if (algo < 0 || algo > 5) return hipErrorInvalidValue;
  const int roles = q.total_s == 0 ? 2 : (q.total_t == 0 ? 1 : 3);
#define OFFK_K2_LAUNCH(A, NL, NS, T4)                                                                               \
  do {                                                                                                              \
    if (roles == 3) hipLaunchKernelGGL((sobel_tdiff_kernel<A, NL, NS, T4, 3>), grid, dim3(ST_THREADS), lds, st, q);      \
    else if (roles == 1) hipLaunchKernelGGL((sobel_tdiff_kernel<A, NL, NS, T4, 1>), grid, dim3(ST_THREADS), lds, st, q); \
    else hipLaunchKernelGGL((sobel_tdiff_kernel<A, NL, NS, T4, 2>), grid, dim3(ST_THREADS), lds, st, q);                 \
  } while (0)
  const int nt = knob("OFFK_K2_NT", 3);     // bit 0: non-temporal loads, bit 1: non-temporal stores (T-blocks)
  const int talgo = algo == 1 ? 1 : (algo >= 4 ? 2 : 0);
#ifdef OFFK_TUNING_KNOBS
#define OFFK_K2_NTSEL(A, T4)                       \
  switch (nt & 3) {                                \
    case 0: OFFK_K2_LAUNCH(A, 0, 0, T4); break;    \
    case 1: OFFK_K2_LAUNCH(A, 1, 0, T4); break;    \
    case 2: OFFK_K2_LAUNCH(A, 0, 1, T4); break;    \
    default: OFFK_K2_LAUNCH(A, 1, 1, T4); break;   \
  }
#else
  (void)nt;
  // rotation: nt loads + nt stores; flat form: its second read of every frame must hit a cache -> nt stores only
#define OFFK_K2_NTSEL(A, T4) if (A == 2) OFFK_K2_LAUNCH(A, 0, 1, T4); else OFFK_K2_LAUNCH(A, 1, 1, T4);
#endif
  if (talgo == 1) {
    if (q.taps4) OFFK_K2_LAUNCH(1, 0, 0, 1); else OFFK_K2_LAUNCH(1, 0, 0, 0);
  } else if (talgo == 2) {
    if (q.taps4) { OFFK_K2_NTSEL(2, 1) } else { OFFK_K2_NTSEL(2, 0) }
  } else {
    if (q.taps4) { OFFK_K2_NTSEL(0, 1) } else { OFFK_K2_NTSEL(0, 0) }
  }
#undef OFFK_K2_NTSEL
#undef OFFK_K2_LAUNCH
  return hipGetLastError();
}

// [32][1][3][3] -> [9][32]
__global__ void repack_dw_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  int i = threadIdx.x;  // 288 threads
  if (i < 9 * kDownCh) {
    int tap = i / kDownCh, c = i - tap * kDownCh;
    dst[i] = src[c * 9 + tap];
  }
}
hipError_t repack_dw_launch(const float* w, float* out, hipStream_t st) {
  hipLaunchKernelGGL(repack_dw_kernel, dim3(1), dim3(320), 0, st, w, out);
  return hipGetLastError();
}

}  // namespace offk
