// Host-side declarations shared by the liboffk translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace offk {

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): the attribute is per device, a process may drive
// several (two_stream.py, tests); heads.hip implements it
hipError_t lds_attr_once(const void* kernel, int bytes);

constexpr int OFFK_CONV_RELU_IN_ = 1, OFFK_CONV_RELU_PRE_ = 2, OFFK_CONV_RELU_POST_ = 4;
constexpr int kNumSites = 9;
constexpr int kGenCh = 128, kDownCh = 32, kUnitCh = 160;

// ---- K4 ------------------------------------------------------------------------
struct ConvDesc {
  const float* x; int x_cs, x_coff;
  int n_img, H, W, Ci;
  const float* w;     // fp32 [Co][K] in the library K order
  const float* bias;
  int Co, KH, KW, stride, pad;
  const float* res; int res_cs, res_coff;
  int flags;
  float* y; int y_cs, y_coff;
  int tile_cfg = -1, splitk = 0;          // < 0 / < 1: pick automatically
  float* partial = nullptr;               // split-K slab scratch
  size_t partial_floats = 0;
  int precision = 0;                      // 0 = the fp32 MFMA pipe (the only value since ABI v9; split-fp32 is a handle property)
  int co_limit = 0;                       // > 0: store only output channels < co_limit (weights padded to Co); needs splitk == 1
  int batch = 1;                          // > 1: that many independent problems of this shape in one launch (1x1 only): problem b
  long long x_bstride = 0, w_bstride = 0, y_bstride = 0;   //   reads x + b * x_bstride, w + b * w_bstride, writes y + b * y_bstride (floats)
  int ngroups = 0;                        // > 0: grouped batch -- g_batch[g] dense 1x1 problems [M][g_Ci[g]] x [Co][g_Ci[g]] -> [M][Co] per group,
  int g_batch[4] = {0, 0, 0, 0}, g_Ci[4] = {0, 0, 0, 0};   //   packed from x + g_x[g], w + g_w[g], y + g_y[g] (floats); Ci / x_cs = the largest
  long long g_x[4] = {0, 0, 0, 0}, g_w[4] = {0, 0, 0, 0}, g_y[4] = {0, 0, 0, 0};
  float* pool_part = nullptr;             // != nullptr: [ceil(M / 32)][2][Co] partial column sums of the stored output (heads.hip: fc_pooled)
  int pool_hw = 0;                        //   rows per image (>= 32); forces splitk = 1
};
void conv2d_auto_plan(long long M, int Co, int nkt, int* cfg_out, int* splitk_out, int precision);
hipError_t conv2d_launch(const ConvDesc& d, hipStream_t st, const char** why);
hipError_t pack_conv_weight_launch(const float* src, int Co, int Ci, int KH, int KW, float* dst, hipStream_t st);

// ---- K4c: one 1x1 -> 3x3 -> 1x1 (+ residual) bottleneck chain at 14x14 per launch (chain_fused.hip, exact fp32) ----
struct ChainArgs {
  const float* x; int x_cs, x_coff;       // chain input, channels-last [n_img * 196][x_cs], Cin channels at x_coff
  int Cin;                                // 64 or 256
  int relu_in;                            // c1 reads relu(x) (28a: x is the pre-ReLU 7x7 output)
  const float* w1; const float* b1;       // c1 [64][Cin], [64]
  const float* w2; const float* b2;       // c2 3x3 64 -> 64 in the library's packed K order [64][2][9][32], [64]
  const float* u2;                        // != nullptr: c2 in Winograd F(2x2, 3x3) form inside the kernel -- its transformed weights
                                          // [16 points][64 co][64 ci] (chain_wino_weight_launch); nullptr: the direct form
  const float* w3; const float* b3;       // c3 [256][K3], [256]
  int K3;                                 // 64, or 128: c3 contracts [t2 | x] (the branch 1x1 on the chain input merged in)
  const float* res; int res_cs, res_coff; // residual (256 channels) or nullptr
  float* y; int y_cs, y_coff;             // output, 256 channels at y_coff
  int n_img, relu_out;
  unsigned x_bytes;                       // bytes behind x + x_coff (< 2^31, else 0: not launchable)
  // chain_split.hip (split-fp32 arithmetic): the plane images of c1 [64][Cin], c2 [64][576] (the packed K order) and c3 [256][64] --
  // wino_pack_split_launch(w, img, Co, K, 1): 6 bytes per element; nullptr: the chain runs on chain_fused.hip
  const void* w1p = nullptr; const void* w2p = nullptr; const void* w3p = nullptr;
  const void* wbp = nullptr; const float* bbr = nullptr;      // chain 28a on chain_split.hip: the branch 1x1 [256][64] on the PRE-ReLU chain input (its
                                                              // plane image and bias) inside the kernel; K3 = 64, no residual
#ifdef OFFK_CHAIN_TIMING
  unsigned long long* dbg;                // cycle sums per phase (tools only)
#endif
};
hipError_t chain14_launch(const ChainArgs& a, hipStream_t st, const char** why);
bool chain14_split_supported(const ChainArgs& a);      // chain_split.hip: the residual chains (K3 = 64) with their plane images
hipError_t chain14_split_launch(const ChainArgs& a, hipStream_t st, const char** why);
// U[4 p + q][co][ci] = sum_ab G[p][a] G[q][b] w[co][ci][a][b] from the packed 3x3 weights [64][2][9][32] (16 * 64 * 64 floats)
hipError_t chain_wino_weight_launch(const float* w2_packed, float* U, hipStream_t st);

// ---- K4w: Winograd for the 3x3 / stride 1 convs on 7x7 maps (winograd.hip): a map = four tiles, F(4, 3) x F(3, 3) per axis ----
// phases = 1: 3x3 / stride 1 on 7x7 maps: 121 points (batches) per image; phases = 4: the polyphase form of a 5x5 / stride 2 / pad 2
// conv on 14x14 maps (K = 4 Ci, four K groups, 400 row-Ci units per image).  A batch entry has one row per image.
constexpr int kWinoPoints = 121, kWinoUnits4 = 400;
hipError_t wino_weight_launch(const float* w_packed, int Co, int Ci, int phases, float* U, hipStream_t st);           // U [121][Co][Ci] / 400 Co Ci floats
hipError_t wino_input_launch(const float* x, int x_cs, int x_coff, int n_img, int Ci, int phases, float* V, hipStream_t st);   // V [121][n_img][Ci] / 400 n_img Ci floats
hipError_t wino_output_launch(const float* M, int n_img, int Co, int phases, const float* bias, const float* res, int res_cs,
                              int res_coff, int flags, float* y, int y_cs, int y_coff, float* pool_part, hipStream_t st);
// the GEMM launches of a conv on the Winograd path: phases == 1: one group of 121 points with K = Ci; phases == 4: four groups
// (81 points K = 4 Ci, 18 + 18 points K = 2 Ci, 4 points K = Ci; winograd.hip).  Offsets in floats for `rows` images / Co output channels.
struct WinoGroup { int batch, kmul; long long v_off, u_off, m_off; };
int wino_groups(int phases, long long rows, int Ci, int Co, WinoGroup out[4]);
// ---- K4g: the batched GEMMs of a conv on a Winograd path as one persistent launch (wino_gemm.hip).  Group g: g_batch[g] problems
//      [M x g_K[g]] . [Co x g_K[g]]^T -> [M x Co], packed one behind the other from x + g_x[g] / w + g_w[g] / y + g_y[g] (floats)
struct WinoGemmArgs {
  const float* x; const float* w; float* y;
  int M, Co;
  int ngroups, g_batch[4], g_K[4];
  long long g_x[4], g_w[4], g_y[4];
  int gm, gn, total_items;     // filled by wino_gemm_launch
  // split-fp32 form (wino_gemm_split.hip): the plane image of w (6 bytes per element at the same element offsets; wino_pack_split_launch)
  const void* w_planes = nullptr;
  long long x_bytes = 0, w_bytes = 0, y_bytes = 0;      // filled by wino_gemm_split_launch: extents of x, the plane image, y
  // split-fp32 form as a 1x1 convolution (epilogue != 0; one problem): channels-last rows with a channel stride and offset, + bias, ReLU,
  // the folded average pool of conv_igemm.hip (pool_part [ceil(M / 32)][2][Co], pool_hw rows per image)
  int epilogue = 0, x_rs = 0, x_coff = 0, y_rs = 0, y_coff = 0, relu = 0, pool_hw = 0;
  const float* bias = nullptr;
  float* pool_part = nullptr;
};
bool wino_gemm_supported(const WinoGemmArgs& a);
hipError_t wino_gemm_launch(const WinoGemmArgs& a, hipStream_t st);
// ---- K4gs: the same GEMMs in split-fp32 arithmetic on the bf16 matrix pipe (wino_gemm_split.hip; Co % 128 == 0, K % 32 == 0, K >= 64)
bool wino_gemm_split_supported(const WinoGemmArgs& a);
hipError_t wino_gemm_split_launch(const WinoGemmArgs& a, hipStream_t st);
hipError_t wino_pack_split_launch(const float* U, void* img, int Co, int K, int nproblems, hipStream_t st);
// ---- K4m: output transform -> (1x1 conv + ReLU) -> input transform between two Winograd convs on 7x7 maps, one launch (wino_mid.hip)
struct WinoMidArgs {
  const float* M;          // GEMM output of the conv in front, Winograd domain: [121][n_img][Cin]
  const float* bias_in;    // [Cin] bias of that conv (its ReLU is applied here); nullptr: none
  int phases_in;           // point order of M: 1 = a 3x3 / stride 1 conv, 4 = the polyphase 5x5 / stride 2 conv (winograd.hip)
  float* x; int x_cs, x_coff;   // != nullptr: also store x = relu(A^T M A + bias) as channels-last rows [n_img * 49][x_cs] at x_coff
  const float* w1;         // 1x1 conv [Cmid][Cin] and its bias [Cmid] (ReLU behind it); nullptr: no 1x1 conv (Cmid == Cin)
  const float* b1;
  int Cin, Cmid, n_img;
  float* V;                // GEMM input of the conv behind: [121][n_img][Cmid] (3x3 / stride 1 point order)
  int nsplit = 0;          // blocks per image the 1x1 conv's output channels are split over (0: the default of the shape; tools)
  const void* w1p = nullptr;   // != nullptr: the plane image of w1 (wino_pack_split_launch(w1, img, Cmid, Cin, 1)): stage B in split-fp32 arithmetic
};
bool wino_mid_supported(int Cin, int Cmid, bool gemm, int phases_in);
hipError_t wino_mid_launch(const WinoMidArgs& a, hipStream_t st);
// ---- K4w7: the 7x7 / stride 2 / pad 3 conv on 28x28 maps in polyphase Winograd form, F(5x5, 4x4) (winograd7.hip): 9 tiles per image,
// 64 points in four groups (49 points K = 4 Ci, 7 + 7 points K = 2 Ci, 1 point K = Ci): 225 row-Ci units of U / V per output channel / tile
constexpr int kWino7Units = 225, kWino7Points = 64, kWino7Tiles = 9;
hipError_t wino7_weight_launch(const float* w_packed, int Co, int Ci, float* U, hipStream_t st);
hipError_t wino7_input_launch(const float* x, int x_cs, int x_coff, int n_img, int Ci, float* V, hipStream_t st);
hipError_t wino7_output_launch(const float* M, int n_img, int Co, const float* bias, int flags, float* y, int y_cs, int y_coff, hipStream_t st);
int wino7_groups(long long T, int Ci, int Co, WinoGroup out[4]);

// ---- K1 ------------------------------------------------------------------------
struct PwSite {
  // feature map as 1..4 channel groups in concat order (the inception block's branches before
  // torch.cat): part p is NCHW [N][cp[p]][HW] (or NHWC [N*HW][cp[p]]), sum cp = C, each cp % 32 == 0
  const float* xp[4];
  int cp[4];
  int nparts;
  const float* w;     // motion_conv_gen rows [128][C] (fp32)
  const float* w_down;   // motion_spatial_down rows [32][C], same format as w (the library's packed copy keeps them adjacent;
                         // weights bound with offk_bind_weight live wherever the caller's parameters do)
  const float* bias;  // [128]
  const float* bias_down;   // [32]
  float* G;           // [N*HW][128]
  float* D;           // [P*HW][32]
  int C, HW, M;       // M = N*HW rows
  int blk_begin;      // first block of this site in the grouped grid
};
struct PwParams {
  PwSite s[kNumSites];
  int nsites, total_blocks;
  int L, P, slice_mode, nhwc;
  const float* zeros;  // >= 16 bytes of zeros in device memory: what a masked-out load reads
#ifdef OFFK_TUNING_KNOBS
  int ablate;          // tools only (OFFK_PW_ABLATE): 1 no feature-map loads, 2 no weight loads, 4 no LDS stores, 8 no MFMAs
#endif
};
int pw_blocks_for(int M);
hipError_t pw_reduce_launch(const PwParams& p, hipStream_t st);

// ---- K1T: K1 fused with the temporal difference (pw_tdiff.hip) -------------------------
struct PtSite {
  const float* xp[4];   // feature map parts, NCHW (as PwSite)
  int cp[4];
  int nparts;
  const float* w;       // gen rows [128][C] fp32
  const float* w_down;  // down rows [32][C]
  const float* wt;      // (kernel-local: the image the kernel reads -- wt16 or wt16s)
  const float* wt16;    // PtParams.bdirect: all 160 rows in the operand order of the 16-pixel form (pw_pack_direct16_launch), read straight into registers
  const void* wt16s;    // PtParams.f32split: the rows as three bf16 planes in the operand order of pw_tdiff_split_kernel (pw_pack_split16_launch)
  const float* bias;    // [128]
  const float* bias_down;   // [32]
  float* D;             // [P*HW][32]
  float* M;             // fusion buffer: T goes to channels [m_coff + 32, m_coff + 160)
  int m_cs, m_coff;
  int C, HW;
  // block layout of the site, filled by pw_tdiff_launch: per temporal group, batch * chunks blocks of one (clip, 32-pixel
  // chunk) each, then nrem blocks for the HW % 32 leftover pixels.  Buffer-addressed kernels pack the leftovers of
  // 32 >> rsh ... clips into one block when HW % 32 is 4, 8 or 16 (28x28: two clips, 14x14: eight clips per block) --
  // the seven 14x14 chunks of a clip were 6.125 chunks of work.
  int chunks;           // pixel chunks of 32 per clip (packed: floor, else ceil)
  int nrem;             // leftover blocks per temporal group (0: none)
  int rsh;              // log2(leftover pixels per clip) in a packed leftover block; 5: one clip per block
  int qpc;              // 16-pixel form, > 0: "quad stream" -- the site's blocks take four consecutive pixel quads each from the
                        // sequence (clip, quad) with qpc = ceil(HW / 4) quads per clip (7x7: 13, the last one holding one pixel):
                        // 49 of 52 lanes of work instead of 49 of 64 with four 16-pixel chunks per clip
  int blk_begin;
};
struct PtParams {
  PtSite s[kNumSites];
  int nsites, total_blocks;
  int B, L, P, slice_mode, tgroups;
  int f32split;              // with bdirect: the split-fp32 form (pw_tdiff_split.hip) instead of pw_tdiff16_kernel
  int bdirect;               // every site carries wt16 (/ wt16s): the weight operand bypasses LDS -- the 16-pixel forms (pw_tdiff16_kernel,
                             // pw_tdiff_split_kernel); 0 (weights bound in the caller's tensors): the fallback pw_tdiff_kernel
  const float* zeros;
#ifdef OFFK_PT_TIMING
  unsigned long long* dbg;   // cycle-counter sums (tools only)
#endif
};
int pt_tgroups(int L);
hipError_t pw_tdiff_launch(const PtParams& p, hipStream_t st);
// w160: [160][C] fp32 (gen rows, then down rows) -> out (160 * C floats), the operand-order image pw_tdiff16_kernel reads
hipError_t pw_pack_direct16_launch(const float* w160, int C, float* out, hipStream_t st);
// pw_tdiff_split.hip: out = 160 * C * 6 bytes; p with the 16-pixel form's block layout (pw_tdiff_launch fills it and calls this)
hipError_t pw_pack_split16_launch(const float* w160, int C, void* out, hipStream_t st);
hipError_t pw_tdiff_split_launch(const PtParams& p, hipStream_t st);

// ---- K2 ------------------------------------------------------------------------
struct StSite {
  const float* G;     // [N*HW][128]
  const float* D;     // [P*HW][32]
  const float* dw;    // depthwise weights: [9][32] tap-major (the library's packed copy) or, dw_ref != 0, the reference's own
                      // [32][1][3][3] layout (a parameter bound with offk_bind_weight)
  const float* db;    // [32] bias or nullptr
  int dw_ref;
  float* Ms;          // spatial gradient out: channels [s_coff, s_coff + 32) of rows of s_cs floats
  float* Mt;          // temporal difference out: channels [t_coff, t_coff + 128) of rows of t_cs floats
  int H, s_cs, s_coff, t_cs, t_coff;
  int wrecip, twrecip;  // ceil(65536 / W), ceil(65536 / (W + 2)): n / W == (n * wrecip) >> 16 for the n the S-blocks divide (st_plan checks)
  int strips, rows;   // S-blocks: strips per plane, rows per strip
  int tchunks;        // T-blocks per clip
  int s_begin;        // first S-block (spatial role) of this site within the S index space
  int t_begin;        // first T-block (temporal role) of this site within the T index space
  unsigned long long drop_base;   // training: mixed base of this site's dropout stream (drop_stream_base)
};
struct StParams {
  StSite s[kNumSites];
  int nsites;
  int total_s, total_t;   // S-blocks = sum P*strips, T-blocks = sum B*tchunks
  int B, L, tpix;         // tpix = pixels per T-block
  int flat_rows;          // (pair, pixel) rows per T-block of the flat form
  unsigned drop_thresh;   // training: 16-bit keep threshold of nn.Dropout(p) on the spatial branch, 0 = eval (identity)
  float drop_scale;       // 1 / (1 - p)
  int taps4;              // every site's depthwise weights are zero outside (0,1) (1,0) (1,2) (2,1) and there is no bias:
                          // the diagonal Sobel of util.py:61 -- four taps instead of nine
  const float* zeros;     // >= 16 bytes of zeros in device memory: what a halo / masked-out load reads
};
void st_plan(int H, int* strips, int* rows);
void st_recips(int H, int* wrecip, int* twrecip);
int st_tchunks(int H);
int st_tchunks_flat(int H, int L);   // T-blocks per clip of the flat form (algo 4 / 5)
hipError_t sobel_tdiff_launch(const StParams& p, int algo, hipStream_t st);

// ---- training side of the OFF units (units_bwd.hip) ----------------------------------
unsigned long long drop_stream_base(unsigned long long seed, int site);
struct UbSite {
  const float* G;     // saved post-ReLU gen output [N*HW][128]
  const float* D;     // saved down output [P*HW][32]
  const float* dw;    // [9][32] tap-major depthwise weights (or the Sobel taps); dw_ref != 0: the reference's [32][1][3][3]
  int dw_ref;
  const float* gm;    // gradient w.r.t. motion_<site>: channels-last rows [P*HW][gm_cs], the unit's 160 channels at gm_coff
  int gm_cs, gm_coff;
  float* dG;          // out: gradient w.r.t. the gen conv output (pre-ReLU) [N*HW][128]
  float* dD;          // out: gradient w.r.t. the down conv output [P*HW][32]
  float* dw_part;     // out: [P*strips][10][32] per-S-block partial sums (9 taps + bias); nullptr = frozen Sobel
  unsigned long long drop_base;
  int H, strips, rows, tchunks, s_begin, t_begin;
};
struct UbParams {
  UbSite s[kNumSites];
  int nsites, total_s, total_t, B, L, tpix;
  unsigned drop_thresh;
  float drop_scale;
  const float* zeros;   // device zero page for halo / masked-out loads
};
hipError_t units_bwd_launch(const UbParams& p, hipStream_t st);

struct WgSite {
  const float* xp[4];   // feature map parts, NCHW (as PwSite)
  int cp[4];
  int nparts;
  const float* dG;      // [N*HW][128]
  const float* dD;      // [P*HW][32]
  float* slab;          // [nchunks][160][ntiles*128] partial weight-gradient tiles
  float* bpart;         // [nchunks][160] partial bias gradients
  int C, HW;
  int tpf;              // 32-pixel K-tiles per frame = ceil(HW / 32)
  int kt_total;         // N * tpf
  int ntiles;           // 128-channel slabs = ceil(C / 128)
  int nchunks;          // ceil(kt_total / kt_per_blk)
  int blk_begin;
};
struct WgParams {
  WgSite s[kNumSites];
  int nsites, total_blocks, L, P, slice_mode, kt_per_blk;
  int precision;        // 0 = the fp32 MFMA pipe
  int dbg;              // ablation bits for tools (1: no X loads, 2: no dG loads)
  const float* zeros;   // >= 16 bytes of zeros in device memory: what a masked-out load reads
};
hipError_t pw_wgrad_launch(const WgParams& p, hipStream_t st);

struct WrSite {
  const float* slab; const float* bpart; const float* dw_part;
  float *gen_w, *gen_b, *down_w, *down_b, *dw_w, *dw_b;   // destinations in the caller's gradient buffer
  int C, cpad, nchunks, nsblocks;
};
struct WrParams {
  WrSite s[kNumSites];
  int nsites, accumulate;
};
hipError_t wgrad_reduce_launch(const WrParams& p, hipStream_t st);
hipError_t consensus_bwd_launch(const float* go, int B, int T, int C, float* gi, hipStream_t st);

// ---- K5 / K6 / layout helpers ----------------------------------------------------
hipError_t head_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, int maxpool, const float* fw,
                       const float* fb, int ncls, float* out, hipStream_t st, const char** why);
hipError_t pool_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, int maxpool, float* pooled,
                       hipStream_t st);
hipError_t fc_launch(const float* pooled, int n_img, int C, const float* fw, const float* fb, int ncls, float* out, hipStream_t st);
// FC of a head whose average pool was folded into the producing conv: part = [ceil(n_img * hw / 32)][2][C] (ConvDesc.pool_part)
// or, tiles != 0, [n_img * 4][C] (the Winograd output transform's per-tile sums)
hipError_t maxpool_rows_launch(const float* x, int x_cs, int x_coff, int n_img, int H, int W, int C, float* part, hipStream_t st);
struct FcPooledJob { const float* part; int hw, tiles, C; const float* fw; const float* fb; float* out; };
struct FcPooledJobs { FcPooledJob job[3]; int njobs, n_img, ncls; };
hipError_t fc_pooled_multi_launch(const FcPooledJobs& j, hipStream_t st);      // the folded-pool FCs of up to three heads in one launch
hipError_t fc_pooled_launch(const float* part, int hw, int tiles, int n_img, int C, const float* fw, const float* fb, int ncls, float* out,
                            hipStream_t st);
hipError_t vec_add_launch(const float* a, const float* b, float* o, int n, hipStream_t st);
hipError_t score_fusion_launch(const float* const* scores, const float* weights, int n, int videos, int crops, int classes,
                               float* fused, int* pred, hipStream_t st);
hipError_t consensus_launch(const float* x, int B, int T, int C, float* out, hipStream_t st);
hipError_t consensus_multi_launch(const float* const x[3], float* const out[3], int nheads, int B, int T, int C, hipStream_t st);
hipError_t nchw_to_nhwc_launch(const float* src, int n_img, int C, int HW, float* dst, hipStream_t st);
hipError_t nhwc_to_nchw_launch(const float* src, int cs, int coff, int n_img, int C, int HW, float* dst, hipStream_t st);
hipError_t repack_dw_launch(const float* w_c33, float* w_tap_c, hipStream_t st);  // [32][1][3][3] -> [9][32]

}  // namespace offk
