// liboffk C ABI: handle, weight registry/packing, workspace plan, forward orchestration.
// See include/offk.h for the contract and the reference lines each entry point stands for.
#include "../../include/offk.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "offk_internal.h"

using namespace offk;

// The batched GEMMs of a conv on a Winograd path (M[point] = V[point] . U[point]^T): one persistent launch of wino_gemm_kernel
// (wino_gemm.hip), or -- persistent == false, OFFK_WINO_GEMM=0 -- gridDim.y problems of the generic 1x1 kernel (conv_igemm.hip, 64 x 64
// LDS-DMA tile).  Bit-identical.  grp / ngrp: winograd.hip's wino_groups / winograd7.hip's wino7_groups; rows = rows of every V[point].
static hipError_t wino_gemms_launch(const WinoGroup* grp, int ngrp, int npoints, int rows, int Ci, int Co, const float* V, const float* U,
                                    float* M, bool persistent, hipStream_t s, const char** why, const void* U_planes = nullptr) {
  if (persistent || U_planes) {
    WinoGemmArgs a{};
    a.x = V; a.w = U; a.y = M; a.M = rows; a.Co = Co; a.ngroups = ngrp;
    for (int gi = 0; gi < ngrp; ++gi) {
      a.g_batch[gi] = grp[gi].batch; a.g_K[gi] = grp[gi].kmul * Ci;
      a.g_x[gi] = grp[gi].v_off; a.g_w[gi] = grp[gi].u_off; a.g_y[gi] = grp[gi].m_off;
    }
    a.w_planes = U_planes;
    if (U_planes && wino_gemm_split_supported(a)) return wino_gemm_split_launch(a, s);     // split-fp32 handles (wino_gemm_split.hip)
    if (persistent && wino_gemm_supported(a)) return wino_gemm_launch(a, s);     // every error of the launch itself goes to the caller
    // a shape the persistent kernel does not take: the generic one
  }
  const int K0 = grp[0].kmul * Ci;
  ConvDesc d;
  d.x = V; d.x_cs = K0; d.x_coff = 0; d.n_img = rows; d.H = 1; d.W = 1; d.Ci = K0;
  d.w = U; d.bias = nullptr; d.Co = Co; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.res = nullptr; d.res_cs = 0; d.res_coff = 0; d.flags = 0;
  d.y = M; d.y_cs = Co; d.y_coff = 0;
  d.tile_cfg = 3; d.splitk = 1; d.precision = 0;
#ifdef OFFK_TUNING_KNOBS
  { const char* e = getenv("OFFK_WINO_CFG"); if (e && (Co % 128 == 0 || atoi(e) == 1 || atoi(e) == 2)) d.tile_cfg = atoi(e); }     // tools: one tile for every Winograd GEMM launch
#endif
  d.batch = npoints; d.x_bstride = (long long)rows * K0; d.w_bstride = (long long)Co * K0; d.y_bstride = (long long)rows * Co;
  if (ngrp > 1) {
    d.ngroups = ngrp;
    for (int gi = 0; gi < ngrp; ++gi) {
      d.g_batch[gi] = grp[gi].batch; d.g_Ci[gi] = grp[gi].kmul * Ci;
      d.g_x[gi] = grp[gi].v_off; d.g_w[gi] = grp[gi].u_off; d.g_y[gi] = grp[gi].m_off;
    }
  }
  return conv2d_launch(d, s, why);
}
// stage entry points have no handle: OFFK_WINO_GEMM is read once per process for them (offk_create reads it per handle)
static bool wino_gemm_stage_default() {
  static const bool on = [] { const char* e = getenv("OFFK_WINO_GEMM"); return !(e && *e == '0'); }();
  return on;
}

namespace {

struct SiteSpec { const char* name; int C, H; };
// reference RGB_OFF.py:395..590 (tap sites) and the view() literals at :600..817
const SiteSpec kSites[kNumSites] = {{"3a", 256, 28}, {"3b", 320, 28}, {"3c", 576, 14}, {"4a", 576, 14}, {"4b", 576, 14},
                                    {"4c", 608, 14}, {"4d", 608, 14}, {"5a", 1024, 7}, {"5b", 1024, 7}};
// launch order of the grouped K1 grid: longest K first so the tail is made of short blocks
const int kPwOrder[kNumSites] = {7, 8, 5, 6, 2, 3, 4, 1, 0};
// fusion buffer of each site and its channel offset there (RGB_OFF.py:656, :760, :832)
const int kSiteFusion[kNumSites] = {0, 0, 1, 1, 1, 1, 1, 2, 2};
const int kSiteCoff[kNumSites] = {0, 160, 0, 160, 320, 480, 640, 0, 160};
const int kFusionC[3] = {320, 1056, 832};

struct ConvSpec { const char* key; int Co, Ci, K, stride, pad; };
enum ConvId {
  C_T28, C1_28A, C2_28A, C3_28A, CB_28A, C1_28B, C2_28B, C3_28B, C1_28C, C2_28C, C3_28C,
  C_T14, C1_14A, C2_14A, C3_14A, CE_14A, C1_14B, C2_14B, C3_14B,
  C_T7, C1_7, C2_7, C3_7, CB_7, kNumConvs
};
// reference RGB_OFF.py:278-290, 308-316, 326-330
const ConvSpec kConvs[kNumConvs] = {
    {"motion_conv_trans_28", 64, 320, 7, 2, 3},      {"motion_conv1_trans_28a", 64, 64, 1, 1, 0},
    {"motion_conv2_trans_28a", 64, 64, 3, 1, 1},     {"motion_conv3_trans_28a", 256, 64, 1, 1, 0},
    {"motion_conv_branch_28a", 256, 64, 1, 1, 0},    {"motion_conv1_trans_28b", 64, 256, 1, 1, 0},
    {"motion_conv2_trans_28b", 64, 64, 3, 1, 1},     {"motion_conv3_trans_28b", 256, 64, 1, 1, 0},
    {"motion_conv1_trans_28c", 64, 256, 1, 1, 0},    {"motion_conv2_trans_28c", 64, 64, 3, 1, 1},
    {"motion_conv3_trans_28c", 256, 64, 1, 1, 0},    {"motion_conv_trans_14", 128, 1056, 5, 2, 2},
    {"motion_conv1_trans_14a", 128, 128, 1, 1, 0},   {"motion_conv2_trans_14a", 128, 128, 3, 1, 1},
    {"motion_conv3_trans_14a", 512, 128, 1, 1, 0},   {"motion_conv_expand_trans_14a", 512, 128, 1, 1, 0},
    {"motion_conv1_trans_14b", 128, 512, 1, 1, 0},   {"motion_conv2_trans_14b", 128, 128, 3, 1, 1},
    {"motion_conv3_trans_14b", 512, 128, 3, 1, 1},   {"motion_conv_trans", 256, 832, 3, 1, 1},
    {"motion_conv1_trans", 256, 256, 1, 1, 0},       {"motion_conv2_trans", 256, 256, 3, 1, 1},
    {"motion_conv3_trans", 1024, 256, 1, 1, 0},      {"motion_conv_branch_trans", 1024, 256, 1, 1, 0}};
// (tile_cfg, splitk) per fusion conv measured fastest by tools/tune_conv.py at P = 384 pairs
// (BASELINE config 2: B = 64, L = 7) on MI355X; other sizes use conv2d_auto_plan.
// (round 2: re-tuned in situ, tools/tune_forward.py --precision fp32, after the fp32 K loop went lean -- with the addressing
// VALU gone the 64x64 tile wins almost everywhere: 6.13 -> 5.89 ms; profiles/r02/tune_fp32_lean.txt)
// (round 3: after the K loops changed again -- LDS-DMA tiles, fused bottleneck chains at 28 -- the conv2d_auto_plan rule
// re-derived from the B = 42 sweep beats this table at P = 384 (5.221 vs 5.253 ms, same box) and an in-situ sweep started
// from it finds nothing better (profiles/r03/tune_b64_fp32_from_heuristic.txt): exact fp32 has no P = 384 table any more;
// {-1, 0} = automatic)
// (main 1x1, branch 1x1) pairs whose outputs are summed: RGB_OFF.py:663-666, :768-770, :839-841
struct MergedSpec { const char* name; int main_id, branch_id; };
const MergedSpec kMerged[3] = {{"merged_28a", C3_28A, CB_28A}, {"merged_14a", C3_14A, CE_14A}, {"merged_7", C3_7, CB_7}};
// test-time shape of the reference eval scripts: 10 crops x 25 segments -> P = 240 (test_rgb_off.py:24-25)
// (round 2: both tables re-tuned in situ with the buffer-addressed loaders, tools/tune_forward.py --batch 10 --length 25 --wide:
// fp32 3.64 -> 3.54 ms, bf16x3 1.77 -> 1.68 ms; profiles/r02/tune_p240_*.txt; splits the sweep reports beyond the slab space of
// the workspace run unsplit and are written as 1 here)
const int kTunedP240[kNumConvs][2] = {   // round 3: the two entries an in-situ sweep from the automatic plans still moves (3.427 -> 3.359 ms)
    {3, 2}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0},
    {4, 4}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0},
    {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}};
// plans of the three merged 1x1 convs (kMerged order): [P == 240][conv] = {tile_cfg, splitk}
const int kMergedPlan[2][3][2] = {
    {{3, 1}, {3, 1}, {3, 1}},                                         // P = 384 (and the default)
    {{1, 1}, {3, 1}, {3, 1}}};                                        // P = 240
struct HeadSpec { const char* key; int C; };
const HeadSpec kHeads[3] = {{"fc_action_motion", 1024}, {"fc_action_motion_28", 256}, {"fc_action_motion_14", 512}};
const char* kSobelKey = "sobel_edge_diagonal.conv.weight";

enum SlotKind { SK_GEN_W, SK_GEN_B, SK_DOWN_W, SK_DOWN_B, SK_DW_W, SK_DW_B, SK_SOBEL, SK_CONV_W, SK_CONV_B, SK_FC_W, SK_FC_B };
struct Slot {
  std::string key;
  std::vector<int64_t> shape;
  SlotKind kind;
  int idx;
  bool set = false;
};

thread_local std::string g_err;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct offk_handle {
  offk_config cfg;
  int N = 0, P = 0;
  std::vector<Slot> slots;
  std::map<std::string, int> index;
  // packed device weights
  float* pw_w[kNumSites] = {};   // [160][C]
  float* pw_wt16[kNumSites] = {};   // the same matrix in the operand order of the fused units kernel's 16-pixel form (fp32)
  float* pw_wt16s[kNumSites] = {};  // ... as three bf16 planes for the split-fp32 form (OFFK_PRECISION_F32SPLIT; 1.5 x the floats)
  bool split_gemm = false;          // a split-fp32 handle runs the Winograd GEMMs with Co % 128 == 0 in split-fp32 too (OFFK_SPLIT_GEMM=0: fp32 pipe)
  int split_gemm_skip = 0;          // (tuning builds: OFFK_SPLIT_GEMM_SKIP, bit k = wino_u[k] stays on the fp32 pipe)
  bool f32split = false;            // created with OFFK_PRECISION_F32SPLIT: cfg.precision is OFFK_PRECISION_FP32 inside the library, the
                                    // kernels that have a split form take it
  bool pw_dirty = true;
  float* pw_b[kNumSites] = {};   // [160]
  float* dw_w[kNumSites] = {};   // [9][32]
  float* dw_b[kNumSites] = {};   // [32] or null
  // unit parameters bound in place (offk_bind_weight): the caller's own tensors, reference layouts, read at launch time
  const float* bnd_gen_w[kNumSites] = {};   // [128][C]
  const float* bnd_gen_b[kNumSites] = {};   // [128]
  const float* bnd_down_w[kNumSites] = {};  // [32][C]
  const float* bnd_down_b[kNumSites] = {};  // [32]
  const float* bnd_dw_w[kNumSites] = {};    // [32][1][3][3]
  const float* bnd_dw_b[kNumSites] = {};    // [32]
  float* sobel_w = nullptr;      // shared [9][32] (diag variant)
  bool sobel_taps4 = false;      // the loaded Sobel weight is zero outside the four taps of util.py:61 -> K2's four-tap path
  float* conv_w[kNumConvs] = {};
  float* conv_b[kNumConvs] = {};
  float* fc_w[3] = {};
  float* fc_b[3] = {};
  // residual-branch 1x1 convs merged into their sibling: out = W3*t + Wb*x == [W3|Wb] * [t|x] (K-concatenated)
  float* merged_w[3] = {};
  float* merged_ws[3] = {};      // split-fp32 handles: the plane images of merged_w[1], [2] and of motion_conv1_trans_14b's weights
  float* c1_14b_ws = nullptr;    // (the 1x1 convs on 7x7 maps that run on wino_gemm_split.hip's kernel with its conv epilogue)
  float* merged_b[3] = {};
  int merged_cfg[3] = {3, 3, 3}, merged_sk[3] = {1, 1, 1};   // kMergedPlan at offk_create
  bool merged_dirty = true;

  // training side (offk_off_units_backward): workspace superset, K1b chunking, gradient-buffer layout
  float* zero_page = nullptr;    // 256 B of zeros (target of masked-out loads)
  bool fused_units = true;       // forward: K1 fused with the temporal difference (OFFK_FUSED_UNITS=0 at offk_create: K1 + K2)
  bool winograd = true;          // fp32: Winograd F(4x4, 3x3) for the three 3x3 / stride 1 convs on 7x7 maps (winograd.hip); OFFK_WINOGRAD=0: direct
  float* wino_u7 = nullptr;      // transformed weights of C_T28 in the four groups of winograd7.hip (225 x Co x Ci floats)
  bool wino_7x7 = true;          // the 7x7 / stride 2 conv of fusion@28 in polyphase Winograd form F(5x5, 4x4) (OFFK_WINOGRAD_7X7=0: direct)
  int wino7_min_p = 12;          // ... from this many pairs (OFFK_WINOGRAD_7X7=<n> with n > 1 at offk_create: tools)
  float* wino_u7s = nullptr;     // the same for wino_u7 (Co = 64: the 64-channel form of the kernel)
  float* wino_us[6] = {};        // split-fp32 handles: the plane images of wino_u (wino_gemm_split.hip), 6 bytes per element; nullptr: fp32 GEMMs
  float* wino_u[6] = {};         // transformed weights [121][Co][Ci] of C3_14B, C_T7, C2_7, C2_14A, C2_14B; 400 x Co x Ci floats of C_T14 (polyphase 5x5 / 2)
  bool wino_dirty = true;
  float* mid_ws[2] = {};         // split-fp32 handles (OFFK_SPLIT_MID=0: off): plane images of the 1x1 convs inside wino_mid (motion_conv1_trans_14a, motion_conv1_trans)
  float* chain_ws[3][4] = {};    // split-fp32 handles (OFFK_SPLIT_CHAIN=0: off): plane images of the chains' c1 / c2 (packed K order) / c3 weights
                                 // (chain_split.hip), [chain 28a, 28b, 28c][conv]; [0][3]: motion_conv_branch_28a, which such a handle runs as a
                                 // 1x1 conv of its own in front of chain 28a (its output is the chain's residual)
  float* chain_u2[3] = {};       // F(2x2, 3x3) weights [16][64][64] of the chains' 3x3 convs C2_28A / B / C (chain_fused.hip, OFFK_CHAIN_WINO)
  bool chain_wino = false;
  int wino5_min_p = 40;          // ... from this many pairs (OFFK_WINOGRAD_5X5=<n> with n > 1 at offk_create: tools)
  bool wino_5x5 = true;          // the 5x5 / stride 2 conv of fusion@14 in polyphase Winograd form (OFFK_WINOGRAD_5X5=0: direct)
  bool wino_gemm = true;         // the batched GEMMs of a Winograd conv as one persistent launch (wino_gemm.hip); false: the generic 1x1 kernel
  bool wino_mid = true;          // fp32 + Winograd: output transform + 1x1 conv + input transform between two Winograd convs in one launch
                                 // (wino_mid.hip); OFFK_WINO_MID=0 at offk_create: three launches
  bool chain = true;             // fp32: one launch per bottleneck chain of fusion@28 (chain_fused.hip); OFFK_CHAIN=0 at offk_create: three convs
  int chain_min_p = 72;          // ... from this many pairs (OFFK_CHAIN=<n> with n > 1 at offk_create: tests / tools)
  size_t train_ws_bytes = 0;
  int wg_kpb = 0;                // 32-pixel K-tiles one pw_wgrad block walks
  std::map<std::string, std::pair<size_t, size_t>> grad_slots;   // key -> (offset, count) in floats
  size_t grad_floats = 0;

  int conv_cfg[kNumConvs];       // tile plan per fusion conv (-1 = automatic)
  int conv_splitk[kNumConvs];    // K-split per fusion conv (0 = automatic)
  size_t splitk_floats = 0;      // size of the "splitk" workspace region
  float* cur_splitk = nullptr;   // that region inside the workspace of the forward being enqueued
  float* cur_pool_part = nullptr;   // set around the one conv launch whose epilogue also emits pooled partial sums (heads)
  bool fold_pool = true;         // 7- / 14-head: average pool in the producing conv's epilogue + MFMA FC (OFFK_FOLD_POOL=0: pool + fc kernels)
  std::vector<void*> allocs;
  // workspace plan
  std::map<std::string, std::pair<size_t, size_t>> regions;
  size_t ws_bytes = 0;
  // profiling: 0 off, 1 per-stage events, 2 per-launch trace (one event in front of every launch group, by name)
  int profiling = 0;
  std::vector<hipEvent_t> events;   // groups of OFFK_NUM_STAGES + 1
  size_t ev_used = 0;
  std::vector<hipEvent_t> tr_events;
  std::vector<int> tr_marks;        // per recorded event: index into tr_names, -1 = end of a forward
  std::vector<std::string> tr_names;
  mutable std::string err;
};

namespace {

int fail(offk_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}
int fail_hip(offk_handle* h, hipError_t e, const char* what) {
  return fail(h, OFFK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, expr)                                   \
  do {                                                     \
    hipError_t e__ = (expr);                               \
    if (e__ != hipSuccess) return fail_hip(h, e__, #expr); \
  } while (0)

// per-launch trace (offk_set_profiling(h, 2)): an event on `st` in front of the launch group `name`; name == nullptr closes
// the forward.  The time of a group is the distance to the next mark on the same stream.
constexpr size_t kTraceMaxEvents = 1 << 16;
int trace_mark(offk_handle* h, hipStream_t st, const char* name) {
  if (h->profiling != 2 || h->tr_marks.size() >= kTraceMaxEvents) return OFFK_OK;
  int id = -1;
  if (name) {
    for (size_t i = 0; i < h->tr_names.size() && id < 0; ++i)
      if (h->tr_names[i] == name) id = (int)i;
    if (id < 0) { id = (int)h->tr_names.size(); h->tr_names.push_back(name); }
  }
  if (h->tr_marks.size() >= h->tr_events.size()) {
    hipEvent_t e;
    HIP_TRY(h, hipEventCreate(&e));
    h->tr_events.push_back(e);
  }
  HIP_TRY(h, hipEventRecord(h->tr_events[h->tr_marks.size()], st));
  h->tr_marks.push_back(id);
  return OFFK_OK;
}

int dev_alloc(offk_handle* h, float** p, size_t nfloats) {
  void* q = nullptr;
  HIP_TRY(h, hipMalloc(&q, nfloats * sizeof(float)));
  HIP_TRY(h, hipMemset(q, 0, nfloats * sizeof(float)));
  h->allocs.push_back(q);
  *p = static_cast<float*>(q);
  return OFFK_OK;
}

void add_slot(offk_handle* h, const std::string& key, std::vector<int64_t> shape, SlotKind kind, int idx) {
  Slot s;
  s.key = key; s.shape = std::move(shape); s.kind = kind; s.idx = idx;
  h->index[key] = (int)h->slots.size();
  h->slots.push_back(std::move(s));
}

void add_region(offk_handle* h, const std::string& name, size_t nfloats) {
  size_t off = align_up(h->ws_bytes, 256);
  h->regions[name] = std::make_pair(off, nfloats * sizeof(float));
  h->ws_bytes = off + nfloats * sizeof(float);
}

float* region(const offk_handle* h, void* ws, const char* name) {
  auto it = h->regions.find(name);
  return reinterpret_cast<float*>(static_cast<char*>(ws) + it->second.first);
}

// S-blocks of the units' backward: 7-row strips at 28x28, whole planes below (two LDS tiles per block; shorter
// strips for a third resident block per CU measured slower: 1.39 ms -> 1.41 .. 1.54 ms for the whole backward)
void ub_plan(int H, int* strips, int* rows) {
  *rows = H >= 28 ? 7 : H;
  *strips = (H + *rows - 1) / *rows;
}
constexpr int kUbTpix = 8;     // pixels per T-block of the backward: one task per thread (as K2, sobel_tdiff.hip st_tpix)

void plan_workspace(offk_handle* h) {
  const size_t N = h->N, P = h->P;
  for (int s = 0; s < kNumSites; ++s) {
    size_t hw = (size_t)kSites[s].H * kSites[s].H;
    add_region(h, std::string("G_") + kSites[s].name, N * hw * kGenCh);
    add_region(h, std::string("D_") + kSites[s].name, P * hw * kDownCh);
  }
  add_region(h, "fusion_28", P * 784 * 320);
  add_region(h, "fusion_14", P * 196 * 1056);
  add_region(h, "fusion_7", P * 49 * 832);
  // fusion@28 temporaries, 14x14 maps
  add_region(h, "xt_28", P * 196 * 128);   // [t2 (c2 output) | x0 (pre-ReLU 7x7 output)]: input of the merged c3+branch conv
  add_region(h, "t1_28", P * 196 * 64);
  add_region(h, "sa_28", P * 196 * 256);
  add_region(h, "sb_28", P * 196 * 256);
  // fusion@14 temporaries, 7x7 maps
  add_region(h, "xu_14", P * 49 * 256);    // [u2 | x1]
  add_region(h, "u1_14", P * 49 * 128);
  add_region(h, "sa_14", P * 49 * 512);
  // fusion@7
  add_region(h, "xv_7", P * 49 * 512);     // [v2 | x2]
  add_region(h, "v1_7", P * 49 * 256);
  add_region(h, "sum_7", P * 49 * 1024);
  // per-pair logits when consensus averages them afterwards
  add_region(h, "logit_7", P * (size_t)h->cfg.num_classes);
  add_region(h, "logit_14", P * (size_t)h->cfg.num_classes);
  add_region(h, "logit_28", P * (size_t)h->cfg.num_classes);
  add_region(h, "pooled_7", P * 1024);
  add_region(h, "pooled_14", P * 512);
  add_region(h, "pooled_28", P * 256);
  add_region(h, "poolpart_28", (size_t)4 * P * 256);   // 28-head: max-pooled cells summed per block of pool rows (maxpool_rows_kernel)
  // average pools of the 7- and 14-heads folded into the producing conv: per 32-row slab, two partial column sums
  add_region(h, "poolpart_7", ((P * 49 + 31) / 32) * 2 * 1024);
  add_region(h, "poolpart_14", ((P * 49 + 31) / 32) * 2 * 512);
  // Winograd path of the 3x3 convs at 7x7 (fp32): transformed input [121][P][Ci <= 832], GEMM output [121][P][Co <= 512],
  // per-tile sums of sum_14b for the 14-head
  if (h->cfg.precision == OFFK_PRECISION_FP32) {
    // widest: the polyphase 7x7 / 2 conv (9 P tiles x 225 x 320 floats: 1.0 GB at P = 384), then the polyphase 5x5 / 2 conv (400 P x 1056)
    add_region(h, "wino_v", std::max((size_t)kWinoUnits4 * P * 1056, (size_t)kWino7Tiles * P * kWino7Units * 320));
    add_region(h, "wino_m", (size_t)kWinoPoints * P * 512);    // (7x7: 64 x 9 P x 64 is smaller)
    add_region(h, "poolpart_14t", (size_t)4 * P * 512);
  }
  // split-K partial slabs: room for 8 slices of the widest large-K conv output (7x7: [P*196, 64], 3x3 @7: [P*49, 256]) -- up
  // to 64 at small P, where the plans split deeper (conv2d_auto_plan); a conv whose plan needs more gets as many as fit
  h->splitk_floats = (size_t)std::min<size_t>(64, std::max<size_t>(8, 1536 / P)) * P * 196 * 64;
  add_region(h, "splitk", h->splitk_floats);
  h->ws_bytes = align_up(h->ws_bytes, 256);

  // ---- training side: backward regions behind the forward layout (offk_train_workspace_bytes) ----
  const size_t fwd_bytes = h->ws_bytes;
  long long work = 0;   // (K-tile, channel slab) steps of the grouped weight-gradient GEMM
  for (int s = 0; s < kNumSites; ++s) {
    const int hw = kSites[s].H * kSites[s].H;
    work += (long long)N * ((hw + 31) / 32) * ((kSites[s].C + 127) / 128);
  }
  h->wg_kpb = (int)std::max<long long>(4, (work + 1535) / 1536);
  for (int s = 0; s < kNumSites; ++s) {
    const size_t hw = (size_t)kSites[s].H * kSites[s].H;
    const std::string n = kSites[s].name;
    int strips, rows;
    ub_plan(kSites[s].H, &strips, &rows);
    const size_t kt_total = N * ((hw + 31) / 32), nchunks = (kt_total + h->wg_kpb - 1) / h->wg_kpb;
    const size_t cpad = (size_t)((kSites[s].C + 127) / 128) * 128;
    add_region(h, "dG_" + n, N * hw * kGenCh);
    add_region(h, "dD_" + n, P * hw * kDownCh);
    add_region(h, "dwp_" + n, P * strips * 10 * kDownCh);
    add_region(h, "wgs_" + n, nchunks * kUnitCh * cpad);
    add_region(h, "wgb_" + n, nchunks * kUnitCh);
    auto slot = [&](const std::string& key, size_t count) {
      h->grad_slots[key] = std::make_pair(h->grad_floats, count);
      h->grad_floats += count;
    };
    slot("motion_conv_gen_" + n + ".weight", (size_t)kGenCh * kSites[s].C);
    slot("motion_conv_gen_" + n + ".bias", kGenCh);
    slot("motion_spatial_down_" + n + ".weight", (size_t)kDownCh * kSites[s].C);
    slot("motion_spatial_down_" + n + ".bias", kDownCh);
    if (h->cfg.variant != OFFK_VARIANT_DIAG_SOBEL) {
      slot("motion_spatial_grad_" + n + ".weight", (size_t)kDownCh * 9);
      slot("motion_spatial_grad_" + n + ".bias", kDownCh);
    }
  }
  h->train_ws_bytes = align_up(h->ws_bytes, 256);
  h->ws_bytes = fwd_bytes;
}

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int want) {
    if (hipGetDevice(&prev) == hipSuccess && prev != want) changed = hipSetDevice(want) == hipSuccess;
  }
  ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

int check_ready(offk_handle* h) {
  for (const Slot& s : h->slots)
    if (!s.set) return fail(h, OFFK_ERR_MISSING_WEIGHT, "weight not set: " + s.key);
  return OFFK_OK;
}

bool slot_set(const offk_handle* h, const std::string& key) {
  auto it = h->index.find(key);
  return it != h->index.end() && h->slots[it->second].set;
}

int site_weights_ready(offk_handle* h, int site, bool need_pw, bool need_dw) {
  const std::string n = kSites[site].name;
  if (need_pw)
    for (const char* k : {"motion_conv_gen_", "motion_spatial_down_"})
      for (const char* sfx : {".weight", ".bias"})
        if (!slot_set(h, std::string(k) + n + sfx)) return fail(h, OFFK_ERR_MISSING_WEIGHT, std::string("weight not set: ") + k + n + sfx);
  if (need_dw) {
    if (h->cfg.variant == OFFK_VARIANT_DIAG_SOBEL) {
      if (!slot_set(h, kSobelKey)) return fail(h, OFFK_ERR_MISSING_WEIGHT, std::string("weight not set: ") + kSobelKey);
    } else {
      for (const char* sfx : {".weight", ".bias"})
        if (!slot_set(h, "motion_spatial_grad_" + n + sfx)) return fail(h, OFFK_ERR_MISSING_WEIGHT, "weight not set: motion_spatial_grad_" + n + sfx);
    }
  }
  return OFFK_OK;
}

void pw_weight_ptrs(const offk_handle* h, int site, const float** w, const float** w_down, const float** b,
                    const float** b_down) {
  const float* own = h->pw_w[site];
  *w = h->bnd_gen_w[site] ? h->bnd_gen_w[site] : own;
  *w_down = h->bnd_down_w[site] ? h->bnd_down_w[site] : own + (size_t)kGenCh * kSites[site].C;
  *b = h->bnd_gen_b[site] ? h->bnd_gen_b[site] : h->pw_b[site];
  *b_down = h->bnd_down_b[site] ? h->bnd_down_b[site] : h->pw_b[site] + kGenCh;
}

void fill_pw_site(const offk_handle* h, int site, const offk_feat_parts& fp, float* G, float* D, PwSite* o) {
  for (int q = 0; q < 4; ++q) { o->xp[q] = q < fp.n_parts ? fp.data[q] : nullptr; o->cp[q] = q < fp.n_parts ? fp.channels[q] : 0; }
  o->nparts = fp.n_parts;
  pw_weight_ptrs(h, site, &o->w, &o->w_down, &o->bias, &o->bias_down);
  o->G = G; o->D = D;
  o->C = kSites[site].C; o->HW = kSites[site].H * kSites[site].H; o->M = h->N * o->HW;
  o->blk_begin = 0;
}
void fill_st_site(const offk_handle* h, int site, const float* G, const float* D, float* M, int m_cs, int m_coff, StSite* o) {
  o->G = G; o->D = D;
  // one unit = [S 32 | T 128] of a channels-last row (RGB_OFF.py:616); the kernel takes the two halves as separate views
  o->Ms = M; o->s_cs = m_cs; o->s_coff = m_coff; o->Mt = M; o->t_cs = m_cs; o->t_coff = m_coff + kDownCh;
  const bool sobel = h->cfg.variant == OFFK_VARIANT_DIAG_SOBEL;
  o->dw = sobel ? h->sobel_w : (h->bnd_dw_w[site] ? h->bnd_dw_w[site] : h->dw_w[site]);
  o->dw_ref = !sobel && h->bnd_dw_w[site] != nullptr;
  o->db = sobel ? nullptr : (h->bnd_dw_b[site] ? h->bnd_dw_b[site] : h->dw_b[site]);
  o->H = kSites[site].H;
  st_plan(o->H, &o->strips, &o->rows);
  st_recips(o->H, &o->wrecip, &o->twrecip);
  o->tchunks = st_tchunks(o->H);
  o->s_begin = 0; o->t_begin = 0;
}

struct DropCfg { unsigned thresh = 0; float scale = 1.f; unsigned long long seed = 0; };
int make_drop(offk_handle* h, unsigned long long seed, double p, DropCfg* d) {
  if (!(p >= 0.0) || p >= 1.0) return fail(h, OFFK_ERR_INVALID, "dropout probability must be in [0, 1)");
  d->thresh = (unsigned)llround(p * 65536.0);
  d->scale = (float)(1.0 / (1.0 - p));
  d->seed = seed;
  return OFFK_OK;
}

int run_sobel_tdiff_all(offk_handle* h, hipStream_t st, void* ws, int algo, const DropCfg& drop = DropCfg()) {
  StParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.nsites = kNumSites; sp.B = h->cfg.batch; sp.L = h->cfg.length;
  sp.drop_thresh = drop.thresh; sp.drop_scale = drop.scale;
  sp.zeros = h->zero_page;
  const char* fus[3] = {"fusion_28", "fusion_14", "fusion_7"};
  sp.taps4 = h->cfg.variant == OFFK_VARIANT_DIAG_SOBEL && h->sobel_taps4;
  int sblk = 0, tblk = 0;
  for (int s = 0; s < kNumSites; ++s) {
    fill_st_site(h, s, region(h, ws, (std::string("G_") + kSites[s].name).c_str()),
                 region(h, ws, (std::string("D_") + kSites[s].name).c_str()), region(h, ws, fus[kSiteFusion[s]]),
                 kFusionC[kSiteFusion[s]], kSiteCoff[s], &sp.s[s]);
    sp.s[s].s_begin = sblk;
    sp.s[s].drop_base = drop_stream_base(drop.seed, s);
    sblk += h->P * sp.s[s].strips;
    if (algo >= 4) sp.s[s].tchunks = st_tchunks_flat(kSites[s].H, h->cfg.length);
    sp.s[s].t_begin = tblk;
    tblk += h->cfg.batch * sp.s[s].tchunks;
  }
  sp.total_s = sblk; sp.total_t = tblk;
  HIP_TRY(h, sobel_tdiff_launch(sp, algo, st));
  return OFFK_OK;
}

offk_feat_parts whole_map(int site, const float* p) {
  offk_feat_parts fp;
  memset(&fp, 0, sizeof(fp));
  fp.n_parts = 1; fp.channels[0] = kSites[site].C; fp.data[0] = p;
  return fp;
}

int check_parts(offk_handle* h, const offk_feat_parts parts[]) {
  for (int s = 0; s < kNumSites; ++s) {
    const offk_feat_parts& fp = parts[s];
    if (fp.n_parts < 1 || fp.n_parts > 4) return fail(h, OFFK_ERR_INVALID, "feature map must come as 1..4 channel groups");
    int sum = 0;
    for (int q = 0; q < fp.n_parts; ++q) {
      if (!fp.data[q]) return fail(h, OFFK_ERR_INVALID, "null feature map");
      if (fp.channels[q] <= 0 || fp.channels[q] % 32) return fail(h, OFFK_ERR_INVALID, "channel groups must be multiples of 32 channels");
      sum += fp.channels[q];
    }
    if (sum != kSites[s].C) return fail(h, OFFK_ERR_INVALID, std::string("channel groups of site ") + kSites[s].name + " do not add up to C");
  }
  return OFFK_OK;
}

int finalize_pw(offk_handle* h, hipStream_t st) {
  if (!h->pw_dirty) return OFFK_OK;
  // the operand-order image the fused units kernel reads straight into registers (pw_tdiff16_kernel; split-fp32: the plane image too)
  for (int s = 0; s < kNumSites; ++s) HIP_TRY(h, pw_pack_direct16_launch(h->pw_w[s], kSites[s].C, h->pw_wt16[s], st));
  if (h->f32split)
    for (int s = 0; s < kNumSites; ++s) HIP_TRY(h, pw_pack_split16_launch(h->pw_w[s], kSites[s].C, h->pw_wt16s[s], st));
  h->pw_dirty = false;
  return OFFK_OK;
}

int run_off_units(offk_handle* h, hipStream_t st, const offk_feat_parts feats[], void* ws, hipEvent_t* ev,
                  const DropCfg& drop = DropCfg()) {
  { int rc = finalize_pw(h, st); if (rc != OFFK_OK) return rc; }
  PwParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.nsites = kNumSites; pp.L = h->cfg.length; pp.P = h->P; pp.slice_mode = h->cfg.slice_mode;
  pp.nhwc = h->cfg.feat_layout == OFFK_FEAT_NHWC;
  pp.zeros = h->zero_page;
  int blk = 0;
  for (int i = 0; i < kNumSites; ++i) {
    int s = kPwOrder[i];
    fill_pw_site(h, s, feats[s], region(h, ws, (std::string("G_") + kSites[s].name).c_str()),
                 region(h, ws, (std::string("D_") + kSites[s].name).c_str()), &pp.s[i]);
    pp.s[i].blk_begin = blk;
    blk += pw_blocks_for(pp.s[i].M);
  }
  pp.total_blocks = blk;
  if (ev) HIP_TRY(h, hipEventRecord(ev[0], st));
  { int rc = trace_mark(h, st, "units:pw_reduce (K1)"); if (rc != OFFK_OK) return rc; }
  HIP_TRY(h, pw_reduce_launch(pp, st));
  if (ev) HIP_TRY(h, hipEventRecord(ev[1], st));

  { int rc = trace_mark(h, st, "units:sobel_tdiff (K2)"); if (rc != OFFK_OK) return rc; }
  { int rc = run_sobel_tdiff_all(h, st, ws, 0, drop); if (rc != OFFK_OK) return rc; }
  if (ev) HIP_TRY(h, hipEventRecord(ev[2], st));
  return OFFK_OK;
}

// Inference path of the units: K1 fused with the temporal difference (pw_tdiff.hip), then the S-blocks of K2 alone.
int run_off_units_fused(offk_handle* h, hipStream_t st, const offk_feat_parts feats[], void* ws, hipEvent_t* ev) {
  { int rc = finalize_pw(h, st); if (rc != OFFK_OK) return rc; }
  PtParams pt;
  memset(&pt, 0, sizeof(pt));
  pt.nsites = kNumSites; pt.B = h->cfg.batch; pt.L = h->cfg.length; pt.P = h->P; pt.slice_mode = h->cfg.slice_mode;
  pt.tgroups = pt_tgroups(h->cfg.length);
  pt.zeros = h->zero_page;
  // the operand-order weight image is the library's own copy: not with contraction weights bound in place
  pt.bdirect = 1;
  pt.f32split = h->f32split;
  for (int s = 0; s < kNumSites; ++s)
    if (h->bnd_gen_w[s] || h->bnd_down_w[s]) pt.bdirect = 0;
  const char* fus[3] = {"fusion_28", "fusion_14", "fusion_7"};
  int blk = 0;
  for (int i = 0; i < kNumSites; ++i) {
    const int s = kPwOrder[i];
    PtSite& o = pt.s[i];
    const offk_feat_parts& fp = feats[s];
    for (int q = 0; q < 4; ++q) { o.xp[q] = q < fp.n_parts ? fp.data[q] : nullptr; o.cp[q] = q < fp.n_parts ? fp.channels[q] : 0; }
    o.nparts = fp.n_parts;
    pw_weight_ptrs(h, s, &o.w, &o.w_down, &o.bias, &o.bias_down);
    o.wt = nullptr;
    o.wt16 = h->pw_wt16[s];
    o.wt16s = h->pw_wt16s[s];
    o.D = region(h, ws, (std::string("D_") + kSites[s].name).c_str());
    o.M = region(h, ws, fus[kSiteFusion[s]]);
    o.m_cs = kFusionC[kSiteFusion[s]]; o.m_coff = kSiteCoff[s];
    o.C = kSites[s].C; o.HW = kSites[s].H * kSites[s].H;
  }
  (void)blk;     // block layout (chunks, leftover blocks, blk_begin, total_blocks): pw_tdiff_launch
  if (ev) HIP_TRY(h, hipEventRecord(ev[0], st));
  { int rc = trace_mark(h, st, "units:pw_tdiff (K1T)"); if (rc != OFFK_OK) return rc; }
  HIP_TRY(h, pw_tdiff_launch(pt, st));
  if (ev) HIP_TRY(h, hipEventRecord(ev[1], st));
  { int rc = trace_mark(h, st, "units:sobel S-blocks (K2 spatial half)"); if (rc != OFFK_OK) return rc; }
  { int rc = run_sobel_tdiff_all(h, st, ws, 3); if (rc != OFFK_OK) return rc; }   // S-blocks only: M[.., coff .. coff+32)
  if (ev) HIP_TRY(h, hipEventRecord(ev[2], st));
  return OFFK_OK;
}

struct View { const float* p; int cs, coff; };

int conv_raw(offk_handle* h, hipStream_t st, const char* name, int Co, int Ci, int K, int stride, int pad, const float* w,
             const float* bias, int cfg, int sk, int n_img, int H, View x, const float* res, int res_cs, int res_coff,
             int flags, float* y, int y_cs, int y_coff, const void* w_planes = nullptr) {
  if (w_planes && K == 1 && stride == 1 && pad == 0 && !res && !(flags & OFFK_CONV_RELU_IN_)) {
    // a 1x1 conv of a split-fp32 handle: wino_gemm_split.hip's kernel with its conv epilogue (bias, ReLU, the folded pool)
    WinoGemmArgs a{};
    a.x = x.p; a.w = w; a.y = y; a.M = n_img * H * H; a.Co = Co; a.ngroups = 1; a.g_batch[0] = 1; a.g_K[0] = Ci;
    a.w_planes = w_planes;
    a.epilogue = 1; a.x_rs = x.cs; a.x_coff = x.coff; a.y_rs = y_cs; a.y_coff = y_coff; a.bias = bias;
    a.relu = (flags & (OFFK_CONV_RELU_PRE_ | OFFK_CONV_RELU_POST_)) ? 1 : 0;
    a.pool_part = h->cur_pool_part; a.pool_hw = h->cur_pool_part ? H * H : 0;
    if (wino_gemm_split_supported(a)) {
      { int rc = trace_mark(h, st, name); if (rc != OFFK_OK) return rc; }
      hipError_t e = wino_gemm_split_launch(a, st);
      if (e != hipSuccess) return fail_hip(h, e, name);
      return OFFK_OK;
    }
  }
  ConvDesc d;
  d.x = x.p; d.x_cs = x.cs; d.x_coff = x.coff; d.n_img = n_img; d.H = H; d.W = H; d.Ci = Ci;
  d.w = w; d.bias = bias; d.Co = Co; d.KH = K; d.KW = K; d.stride = stride; d.pad = pad;
  d.res = res; d.res_cs = res_cs; d.res_coff = res_coff; d.flags = flags;
  d.y = y; d.y_cs = y_cs; d.y_coff = y_coff;
  d.tile_cfg = cfg; d.splitk = sk;
  d.partial = h->cur_splitk; d.partial_floats = h->splitk_floats;
  d.precision = h->cfg.precision;
  d.pool_part = h->cur_pool_part; d.pool_hw = h->cur_pool_part ? H * H / (stride * stride) : 0;
  const char* why = nullptr;
  { int rc = trace_mark(h, st, name); if (rc != OFFK_OK) return rc; }
  hipError_t e = conv2d_launch(d, st, &why);
  if (e != hipSuccess) return fail(h, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, std::string(name) + ": " + (why ? why : hipGetErrorString(e)));
  return OFFK_OK;
}

int conv(offk_handle* h, hipStream_t st, ConvId id, int n_img, int H, View x, const float* res, int res_cs, int res_coff,
         int flags, float* y, int y_cs, int y_coff) {
  const ConvSpec& c = kConvs[id];
  const float* w = h->conv_w[id];
  return conv_raw(h, st, c.key, c.Co, c.Ci, c.K, c.stride, c.pad, w, h->conv_b[id], h->conv_cfg[id], h->conv_splitk[id], n_img,
                  H, x, res, res_cs, res_coff, flags, y, y_cs, y_coff, id == C1_14B ? h->c1_14b_ws : nullptr);
}

// main 1x1 + branch 1x1 as ONE conv over the channel-concatenated input [t | x]
int conv_merged(offk_handle* h, hipStream_t st, int m, int n_img, int H, View x, int flags, float* y, int y_cs, int y_coff) {
  const ConvSpec& a = kConvs[kMerged[m].main_id];
  const ConvSpec& b = kConvs[kMerged[m].branch_id];
  const float* w = h->merged_w[m];
  return conv_raw(h, st, kMerged[m].name, a.Co, a.Ci + b.Ci, 1, 1, 0, w, h->merged_b[m], h->merged_cfg[m], h->merged_sk[m], n_img,
                  H, x, nullptr, 0, 0, flags, y, y_cs, y_coff, h->merged_ws[m]);
}

// (re)build the merged weights after any conv weight changed: [Co][Ci_main | Ci_branch], bias = b_main + b_branch
int finalize_merged(offk_handle* h, hipStream_t st) {
  if (!h->merged_dirty) return OFFK_OK;
  for (int m = 0; m < 3; ++m) {
    const ConvSpec& a = kConvs[kMerged[m].main_id];
    const ConvSpec& b = kConvs[kMerged[m].branch_id];
    const size_t K = (size_t)a.Ci + b.Ci;
    HIP_TRY(h, hipMemcpy2DAsync(h->merged_w[m], K * 4, h->conv_w[kMerged[m].main_id], (size_t)a.Ci * 4, (size_t)a.Ci * 4, a.Co,
                                hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpy2DAsync(h->merged_w[m] + a.Ci, K * 4, h->conv_w[kMerged[m].branch_id], (size_t)b.Ci * 4, (size_t)b.Ci * 4,
                                a.Co, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, vec_add_launch(h->conv_b[kMerged[m].main_id], h->conv_b[kMerged[m].branch_id], h->merged_b[m], a.Co, st));
    if (h->merged_ws[m]) HIP_TRY(h, wino_pack_split_launch(h->merged_w[m], h->merged_ws[m], a.Co, (int)K, 1, st));
  }
  if (h->c1_14b_ws) HIP_TRY(h, wino_pack_split_launch(h->conv_w[C1_14B], h->c1_14b_ws, kConvs[C1_14B].Co, kConvs[C1_14B].Ci, 1, st));
  h->merged_dirty = false;
  return OFFK_OK;
}
int finalize_wino(offk_handle* h, hipStream_t st) {
  if (!h->winograd || !h->wino_dirty) return OFFK_OK;
  const ConvId wid[6] = {C3_14B, C_T7, C2_7, C2_14A, C2_14B, C_T14};
  for (int k = 0; k < 6; ++k) {
    const ConvSpec& c = kConvs[wid[k]];
    HIP_TRY(h, wino_weight_launch(h->conv_w[wid[k]], c.Co, c.Ci, k == 5 ? 4 : 1, h->wino_u[k], st));
    if (!h->wino_us[k]) continue;
    WinoGroup grp[4];
    const int ngrp = wino_groups(k == 5 ? 4 : 1, 1, c.Ci, c.Co, grp);
    for (int g = 0; g < ngrp; ++g)
      HIP_TRY(h, wino_pack_split_launch(h->wino_u[k] + grp[g].u_off, reinterpret_cast<char*>(h->wino_us[k]) + grp[g].u_off * 6, c.Co,
                                        grp[g].kmul * c.Ci, grp[g].batch, st));
  }
  if (h->wino_u7) HIP_TRY(h, wino7_weight_launch(h->conv_w[C_T28], kConvs[C_T28].Co, kConvs[C_T28].Ci, h->wino_u7, st));
  if (h->wino_u7 && h->wino_u7s) {
    const ConvSpec& c = kConvs[C_T28];
    WinoGroup grp[4];
    const int ngrp = wino7_groups(1, c.Ci, c.Co, grp);
    for (int g = 0; g < ngrp; ++g)
      HIP_TRY(h, wino_pack_split_launch(h->wino_u7 + grp[g].u_off, reinterpret_cast<char*>(h->wino_u7s) + grp[g].u_off * 6, c.Co,
                                        grp[g].kmul * c.Ci, grp[g].batch, st));
  }
  {
    const ConvId c2[3] = {C2_28A, C2_28B, C2_28C};
    for (int k = 0; k < 3; ++k)
      if (h->chain_u2[k]) HIP_TRY(h, chain_wino_weight_launch(h->conv_w[c2[k]], h->chain_u2[k], st));
    const ConvId c1[3] = {C1_28A, C1_28B, C1_28C}, c3[3] = {C3_28A, C3_28B, C3_28C};
    for (int k = 0; k < 3; ++k) {
      if (!h->chain_ws[k][0]) continue;
      HIP_TRY(h, wino_pack_split_launch(h->conv_w[c1[k]], h->chain_ws[k][0], 64, kConvs[c1[k]].Ci, 1, st));
      HIP_TRY(h, wino_pack_split_launch(h->conv_w[c2[k]], h->chain_ws[k][1], 64, 576, 1, st));
      HIP_TRY(h, wino_pack_split_launch(h->conv_w[c3[k]], h->chain_ws[k][2], 256, 64, 1, st));
    }
    if (h->chain_ws[0][3]) HIP_TRY(h, wino_pack_split_launch(h->conv_w[CB_28A], h->chain_ws[0][3], 256, 64, 1, st));
    const ConvId midc[2] = {C1_14A, C1_7};
    for (int k = 0; k < 2; ++k)
      if (h->mid_ws[k]) HIP_TRY(h, wino_pack_split_launch(h->conv_w[midc[k]], h->mid_ws[k], kConvs[midc[k]].Co, kConvs[midc[k]].Ci, 1, st));
  }
  h->wino_dirty = false;
  return OFFK_OK;
}
#define TRY(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != OFFK_OK) return rc__; \
  } while (0)

}  // namespace

extern "C" {

int offk_abi_version(void) { return OFFK_ABI_VERSION; }

const char* offk_last_error(const offk_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int offk_create(const offk_config* cfg, offk_handle** out) {
  if (!cfg || !out) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: null argument");
  *out = nullptr;
  if (cfg->batch < 1 || cfg->length < 2) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: need batch >= 1 and length >= 2");
  if (cfg->variant != OFFK_VARIANT_RGB_LEARNED_DW && cfg->variant != OFFK_VARIANT_DIAG_SOBEL) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad variant");
  if (cfg->slice_mode != OFFK_SLICE_REFERENCE_FLAT && cfg->slice_mode != OFFK_SLICE_PER_CLIP) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad slice_mode");
  if (cfg->consensus != OFFK_CONSENSUS_NONE && cfg->consensus != OFFK_CONSENSUS_AVG) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad consensus");
  if (cfg->feat_layout != OFFK_FEAT_NCHW && cfg->feat_layout != OFFK_FEAT_NHWC) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad feat_layout");
  if (cfg->precision != OFFK_PRECISION_FP32 && cfg->precision != OFFK_PRECISION_F32SPLIT)
    return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad precision (0 = fp32, 2 = f32split; 1 was the two-plane bf16x3 mode, retired in ABI v9)");
  if (cfg->num_classes < 1 || cfg->num_classes > 4096) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: bad num_classes");
  if ((long long)cfg->batch * cfg->length * 784 * 320 > 0x7fffffffLL) return fail(nullptr, OFFK_ERR_INVALID, "offk_create: batch*length too large for one call; shard the clips");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, OFFK_ERR_NO_DEVICE, "offk_create: HIP device not available");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || !strstr(prop.gcnArchName, "gfx950"))
    return fail(nullptr, OFFK_ERR_NO_DEVICE, "offk_create: device is not gfx950 (MI355X); this library has no other code path");

  offk_handle* h = new offk_handle();
  h->cfg = *cfg;
  offk_config cfg_in = *cfg;
  if (cfg_in.precision == OFFK_PRECISION_F32SPLIT) { h->f32split = true; h->cfg.precision = cfg_in.precision = OFFK_PRECISION_FP32; }
  cfg = &cfg_in;        // below: the library's own view (split-fp32 = the fp32 plans and buffers + the split kernels' weight images)
  h->N = cfg->batch * cfg->length;
  h->P = cfg->batch * (cfg->length - 1);
  for (int c = 0; c < kNumConvs; ++c) {
    const int (*tab)[2] = h->P == 240 ? kTunedP240 : nullptr;      // (P = 384: the automatic plans, see above)
    h->conv_cfg[c] = tab ? tab[c][0] : -1;
    h->conv_splitk[c] = tab ? tab[c][1] : 0;
  }
  for (int m = 0; m < 3; ++m) {
    const int (*mp)[2] = kMergedPlan[h->P == 240];
    h->merged_cfg[m] = mp[m][0];
    h->merged_sk[m] = mp[m][1];
  }
  DeviceGuard guard(cfg->device);
  int rc = OFFK_OK;
  for (int s = 0; s < kNumSites && rc == OFFK_OK; ++s) {
    const std::string n = kSites[s].name;
    const int C = kSites[s].C;
    add_slot(h, "motion_conv_gen_" + n + ".weight", {kGenCh, C, 1, 1}, SK_GEN_W, s);
    add_slot(h, "motion_conv_gen_" + n + ".bias", {kGenCh}, SK_GEN_B, s);
    add_slot(h, "motion_spatial_down_" + n + ".weight", {kDownCh, C, 1, 1}, SK_DOWN_W, s);
    add_slot(h, "motion_spatial_down_" + n + ".bias", {kDownCh}, SK_DOWN_B, s);
    rc = dev_alloc(h, &h->pw_w[s], (size_t)kUnitCh * C);
    if (rc == OFFK_OK && cfg->precision == OFFK_PRECISION_FP32) rc = dev_alloc(h, &h->pw_wt16[s], (size_t)kUnitCh * C);
    if (rc == OFFK_OK && h->f32split) rc = dev_alloc(h, &h->pw_wt16s[s], (size_t)kUnitCh * C * 3 / 2);
    if (rc == OFFK_OK) rc = dev_alloc(h, &h->pw_b[s], kUnitCh);
    if (cfg->variant == OFFK_VARIANT_RGB_LEARNED_DW && rc == OFFK_OK) {
      add_slot(h, "motion_spatial_grad_" + n + ".weight", {kDownCh, 1, 3, 3}, SK_DW_W, s);
      add_slot(h, "motion_spatial_grad_" + n + ".bias", {kDownCh}, SK_DW_B, s);
      rc = dev_alloc(h, &h->dw_w[s], 9 * kDownCh);
      if (rc == OFFK_OK) rc = dev_alloc(h, &h->dw_b[s], kDownCh);
    }
  }
  if (cfg->variant == OFFK_VARIANT_DIAG_SOBEL && rc == OFFK_OK) {
    add_slot(h, kSobelKey, {kDownCh, 1, 3, 3}, SK_SOBEL, 0);
    rc = dev_alloc(h, &h->sobel_w, 9 * kDownCh);
  }
  for (int c = 0; c < kNumConvs && rc == OFFK_OK; ++c) {
    const ConvSpec& cs = kConvs[c];
    add_slot(h, std::string(cs.key) + ".weight", {cs.Co, cs.Ci, cs.K, cs.K}, SK_CONV_W, c);
    add_slot(h, std::string(cs.key) + ".bias", {cs.Co}, SK_CONV_B, c);
    rc = dev_alloc(h, &h->conv_w[c], (size_t)cs.Co * cs.Ci * cs.K * cs.K);
    if (rc == OFFK_OK) rc = dev_alloc(h, &h->conv_b[c], cs.Co);
  }
  for (int m = 0; m < 3 && rc == OFFK_OK; ++m) {
    const ConvSpec& a = kConvs[kMerged[m].main_id];
    const ConvSpec& b = kConvs[kMerged[m].branch_id];
    const size_t n = (size_t)a.Co * (a.Ci + b.Ci);
    rc = dev_alloc(h, &h->merged_w[m], n);
    if (rc == OFFK_OK) rc = dev_alloc(h, &h->merged_b[m], a.Co);
  }
  for (int k = 0; k < 3 && rc == OFFK_OK; ++k) {
    add_slot(h, std::string(kHeads[k].key) + ".weight", {cfg->num_classes, kHeads[k].C}, SK_FC_W, k);
    add_slot(h, std::string(kHeads[k].key) + ".bias", {cfg->num_classes}, SK_FC_B, k);
    rc = dev_alloc(h, &h->fc_w[k], (size_t)cfg->num_classes * kHeads[k].C);
    if (rc == OFFK_OK) rc = dev_alloc(h, &h->fc_b[k], cfg->num_classes);

  }
  if (rc != OFFK_OK) {
    g_err = h->err;
    offk_destroy(h);
    return rc;
  }
  if (dev_alloc(h, &h->zero_page, 64) != OFFK_OK) {
    g_err = h->err;
    offk_destroy(h);
    return OFFK_ERR_HIP;
  }
  // The path switches of the product library, all read HERE and nowhere else (INTEGRATION.md lists them): each names one
  // algorithm choice of the exact-fp32 forward; "0" = off, a number > 1 = use it from that many frame pairs P = B (L - 1).
  //   OFFK_FUSED_UNITS   K1 fused with the temporal difference (pw_tdiff.hip); 0: K1 + K2 (what training always runs)
  //   OFFK_WINOGRAD      Winograd forms of the k x k fusion convs (winograd.hip, winograd7.hip); 0: direct implicit GEMMs everywhere
  //   OFFK_WINOGRAD_5X5  ... of the 5x5 / stride 2 conv (default: from P = 40)     OFFK_WINOGRAD_7X7  ... of the 7x7 / stride 2 conv (from P = 12)
  //   OFFK_CHAIN         one launch per bottleneck chain of fusion@28 (chain_fused.hip; from P = 72)
  //   OFFK_FOLD_POOL     7- / 14-head average pools taken in the producing conv's epilogue; 0: pool + fc kernels
  //   OFFK_WINO_MID      what sits between two Winograd convs on 7x7 maps (output transform, 1x1 conv, input transform) in one launch
  //                      (wino_mid.hip); 0: three launches
  //   OFFK_CHAIN_WINO    the 3x3 conv inside a bottleneck chain in Winograd F(2x2, 3x3) form (chain_fused.hip); 0: direct (also with OFFK_WINOGRAD=0)
  //   OFFK_WINO_GEMM     the batched GEMMs of a Winograd conv as one persistent launch (wino_gemm.hip); 0: one block of the generic 1x1 kernel per
  //                      tile (bit-identical).  The handle-less stage entry points read it once per process.
  { const char* e = getenv("OFFK_FUSED_UNITS"); h->fused_units = !(e && *e == '0'); }
  { const char* e = getenv("OFFK_CHAIN"); h->chain = !(e && *e == '0'); if (e && atoi(e) > 1) h->chain_min_p = atoi(e);
    // split-fp32 handles: chain14_split_kernel from the first pair on (a block of it alone on a CU is ~25 us per chain where the fp32 kernel's
    // sixteen phases are 45 us: B = 1 .. 4 0.432 / 0.482 / 0.565 -> 0.425 / 0.477 / 0.555 ms, B = 8 level; profiles/r06/small_batch_gates.txt)
    else if (h->f32split) { const char* c = getenv("OFFK_SPLIT_CHAIN"); if (!(c && *c == '0')) h->chain_min_p = 1; } }
  { const char* e = getenv("OFFK_FOLD_POOL"); h->fold_pool = !(e && *e == '0'); }
  { const char* e = getenv("OFFK_WINO_MID"); h->wino_mid = !(e && *e == '0'); }
  { const char* e = getenv("OFFK_WINO_GEMM"); h->wino_gemm = !(e && *e == '0'); }
  { const char* e = getenv("OFFK_WINOGRAD"); h->winograd = !(e && *e == '0') && cfg->precision == OFFK_PRECISION_FP32; }
  { const char* e = getenv("OFFK_WINOGRAD_5X5"); h->wino_5x5 = !(e && *e == '0'); if (e && atoi(e) > 1) h->wino5_min_p = atoi(e); }
  { const char* e = getenv("OFFK_WINOGRAD_7X7"); h->wino_7x7 = !(e && *e == '0'); if (e && atoi(e) > 1) h->wino7_min_p = atoi(e); }
  if (h->winograd) {
    const ConvId wid[6] = {C3_14B, C_T7, C2_7, C2_14A, C2_14B, C_T14};
    for (int k = 0; k < 6; ++k)
      if (dev_alloc(h, &h->wino_u[k], (size_t)(k == 5 ? kWinoUnits4 : kWinoPoints) * kConvs[wid[k]].Co * kConvs[wid[k]].Ci) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    { const char* e = getenv("OFFK_SPLIT_GEMM"); h->split_gemm = h->f32split && !(e && *e == '0'); }
#ifdef OFFK_TUNING_KNOBS
    { const char* e = getenv("OFFK_SPLIT_GEMM_SKIP"); if (e) h->split_gemm_skip = atoi(e); }
#endif
    if (h->split_gemm)
      for (int k = 0; k < 6; ++k) {
        const size_t elems = (size_t)(k == 5 ? kWinoUnits4 : kWinoPoints) * kConvs[wid[k]].Co * kConvs[wid[k]].Ci;
        if (kConvs[wid[k]].Co % 64 == 0 && kConvs[wid[k]].Ci >= 64 && !(h->split_gemm_skip & (1 << k)) &&
            dev_alloc(h, &h->wino_us[k], (elems * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
      }
    if (h->split_gemm && !(h->split_gemm_skip & 128)) {      // the 1x1 convs on 7x7 maps
      for (int m = 1; m < 3; ++m) {
        const size_t elems = (size_t)kConvs[kMerged[m].main_id].Co * (kConvs[kMerged[m].main_id].Ci + kConvs[kMerged[m].branch_id].Ci);
        if (dev_alloc(h, &h->merged_ws[m], (elems * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
      }
      if (dev_alloc(h, &h->c1_14b_ws, ((size_t)kConvs[C1_14B].Co * kConvs[C1_14B].Ci * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    }
    if (h->wino_7x7 && h->split_gemm && !(h->split_gemm_skip & 64) &&
        dev_alloc(h, &h->wino_u7s, ((size_t)kWino7Units * kConvs[C_T28].Co * kConvs[C_T28].Ci * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    if (h->wino_7x7 && dev_alloc(h, &h->wino_u7, (size_t)kWino7Units * kConvs[C_T28].Co * kConvs[C_T28].Ci) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    { const char* e = getenv("OFFK_CHAIN_WINO"); h->chain_wino = h->chain && !(e && *e == '0'); }
    if (h->chain_wino)
      for (int k = 0; k < 3; ++k)
        if (dev_alloc(h, &h->chain_u2[k], (size_t)16 * 64 * 64) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    { const char* e = getenv("OFFK_SPLIT_CHAIN"); if (h->chain && h->winograd && h->f32split && !(e && *e == '0')) {      // (h->winograd: finalize_wino packs them)
      const size_t elems[4] = {(size_t)64 * 256, (size_t)64 * 576, (size_t)256 * 64, (size_t)256 * 64};
      for (int k = 0; k < 3; ++k)
        for (int q = 0; q < (k == 0 ? 4 : 3); ++q)
          if (dev_alloc(h, &h->chain_ws[k][q], (elems[q] * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    } }
    { const char* e = getenv("OFFK_SPLIT_MID"); if (h->wino_mid && h->winograd && h->f32split && !(e && *e == '0')) {
      const ConvId midc[2] = {C1_14A, C1_7};
      for (int k = 0; k < 2; ++k)
        if (dev_alloc(h, &h->mid_ws[k], ((size_t)kConvs[midc[k]].Co * kConvs[midc[k]].Ci * 3 + 1) / 2) != OFFK_OK) { g_err = h->err; offk_destroy(h); return OFFK_ERR_HIP; }
    } }
  }
  plan_workspace(h);
  *out = h;
  return OFFK_OK;
}

int offk_destroy(offk_handle* h) {
  if (!h) return OFFK_OK;
  DeviceGuard guard(h->cfg.device);
  for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->tr_events) (void)hipEventDestroy(e);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return OFFK_OK;
}

int offk_set_weight(offk_handle* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!h || !key || !data || !shape) return fail(h, OFFK_ERR_INVALID, "offk_set_weight: null argument");
  std::string k(key);
  if (k.compare(0, 7, "module.") == 0) k = k.substr(7);
  auto it = h->index.find(k);
  if (it == h->index.end()) return fail(h, OFFK_ERR_UNKNOWN_KEY, "offk_set_weight: not an OFF sub-network key for this variant: " + k);
  Slot& s = h->slots[it->second];
  bool ok = ndim == (int)s.shape.size();
  size_t n = 1;
  for (int i = 0; ok && i < ndim; ++i) { ok = shape[i] == s.shape[i]; n *= (size_t)s.shape[i]; }
  if (!ok) return fail(h, OFFK_ERR_INVALID, "offk_set_weight: shape mismatch for " + k);
  DeviceGuard guard(h->cfg.device);
  // A load-time call, blocking by contract: wait for everything enqueued on the device first -- a kernel that still reads
  // the copy being replaced, or (device source) the kernel that is still producing `data` on some non-blocking stream.
  // Parameters that change every step are not pushed through here: offk_bind_weight.
  HIP_TRY(h, hipDeviceSynchronize());
  const size_t bytes = n * sizeof(float);
  auto copy = [&](float* dst) -> int {
    HIP_TRY(h, hipMemcpy(dst, data, bytes, hipMemcpyDefault));
    return OFFK_OK;
  };
  auto staged = [&](auto&& pack) -> int {   // copy to a temp device buffer, then run a packing kernel
    void* tmp = nullptr;
    HIP_TRY(h, hipMalloc(&tmp, bytes));
    hipError_t e = hipMemcpy(tmp, data, bytes, hipMemcpyDefault);
    if (e == hipSuccess) e = pack(static_cast<const float*>(tmp));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail_hip(h, e, "offk_set_weight pack");
    return OFFK_OK;
  };
  int rc = OFFK_OK;
  const int i = s.idx;
  switch (s.kind) {
    case SK_GEN_W: rc = copy(h->pw_w[i]); break;
    case SK_DOWN_W: rc = copy(h->pw_w[i] + (size_t)kGenCh * kSites[i].C); break;
    case SK_GEN_B: rc = copy(h->pw_b[i]); break;
    case SK_DOWN_B: rc = copy(h->pw_b[i] + kGenCh); break;
    case SK_DW_W: rc = staged([&](const float* t) { return repack_dw_launch(t, h->dw_w[i], nullptr); }); break;
    case SK_DW_B: rc = copy(h->dw_b[i]); break;
    case SK_SOBEL: {
      rc = staged([&](const float* t) { return repack_dw_launch(t, h->sobel_w, nullptr); });
      // frozen parameter (util.py:72), but it travels in checkpoints: take the four-tap path only for weights that
      // really are zero at the corners and the centre
      std::vector<float> host(n);
      if (rc == OFFK_OK && hipMemcpy(host.data(), data, bytes, hipMemcpyDefault) == hipSuccess) {
        bool four = true;
        for (int c = 0; c < kDownCh && four; ++c)
          for (int tap : {0, 2, 4, 6, 8}) four = four && host[(size_t)c * 9 + tap] == 0.f;
        h->sobel_taps4 = four;
      } else {
        h->sobel_taps4 = false;
      }
      break;
    }
    case SK_CONV_W:
      if (kConvs[i].K == 1) rc = copy(h->conv_w[i]);
      else rc = staged([&](const float* t) { return pack_conv_weight_launch(t, kConvs[i].Co, kConvs[i].Ci, kConvs[i].K, kConvs[i].K, h->conv_w[i], nullptr); });
      break;
    case SK_CONV_B: rc = copy(h->conv_b[i]); break;
    case SK_FC_W: rc = copy(h->fc_w[i]); break;
    case SK_FC_B: rc = copy(h->fc_b[i]); break;
  }
  if (rc == OFFK_OK) {
    s.set = true;
    switch (s.kind) {       // an explicit copy replaces an earlier in-place binding of the same key
      case SK_GEN_W: h->bnd_gen_w[i] = nullptr; break;
      case SK_GEN_B: h->bnd_gen_b[i] = nullptr; break;
      case SK_DOWN_W: h->bnd_down_w[i] = nullptr; break;
      case SK_DOWN_B: h->bnd_down_b[i] = nullptr; break;
      case SK_DW_W: h->bnd_dw_w[i] = nullptr; break;
      case SK_DW_B: h->bnd_dw_b[i] = nullptr; break;
      default: break;
    }
  }
  if (s.kind == SK_CONV_W || s.kind == SK_CONV_B) h->merged_dirty = true;
  if (s.kind == SK_CONV_W) h->wino_dirty = true;
  if (s.kind == SK_GEN_W || s.kind == SK_DOWN_W) h->pw_dirty = true;
  return rc;
}

int offk_bind_weight(offk_handle* h, const char* key, const float* device_data, const int64_t* shape, int ndim) {
  if (!h || !key || !device_data || !shape) return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: null argument");
  std::string k(key);
  if (k.compare(0, 7, "module.") == 0) k = k.substr(7);
  auto it = h->index.find(k);
  if (it == h->index.end()) return fail(h, OFFK_ERR_UNKNOWN_KEY, "offk_bind_weight: not an OFF sub-network key for this variant: " + k);
  Slot& s = h->slots[it->second];
  // the tensor is read in place by every later launch (buffer descriptors sized from the key's shape): check it IS that shape
  bool shape_ok = ndim == (int)s.shape.size();
  size_t need = sizeof(float);
  for (int i = 0; shape_ok && i < ndim; ++i) { shape_ok = shape[i] == s.shape[i]; need *= (size_t)s.shape[i]; }
  if (!shape_ok) return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: shape mismatch for " + k);
  if ((reinterpret_cast<uintptr_t>(device_data) & 15) != 0) return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: pointer must be 16-byte aligned: " + k);
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, device_data) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != h->cfg.device) {
    (void)hipGetLastError();
    return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: needs a device pointer on the handle's device: " + k);
  }
  {   // ... and that the allocation behind the pointer really holds that many bytes from there on
    hipDeviceptr_t base = nullptr;
    size_t asz = 0;
    if (hipMemGetAddressRange(&base, &asz, const_cast<float*>(device_data)) == hipSuccess) {
      const size_t off = (size_t)(reinterpret_cast<const char*>(device_data) - reinterpret_cast<const char*>(base));
      if (off > asz || asz - off < need) return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: allocation too small for " + k);
    } else {
      (void)hipGetLastError();
    }
  }
  const int i = s.idx;
  switch (s.kind) {
    case SK_GEN_W: h->bnd_gen_w[i] = device_data; break;
    case SK_GEN_B: h->bnd_gen_b[i] = device_data; break;
    case SK_DOWN_W: h->bnd_down_w[i] = device_data; break;
    case SK_DOWN_B: h->bnd_down_b[i] = device_data; break;
    case SK_DW_W: h->bnd_dw_w[i] = device_data; break;
    case SK_DW_B: h->bnd_dw_b[i] = device_data; break;
    default:
      return fail(h, OFFK_ERR_INVALID, "offk_bind_weight: only the OFF units' parameters (motion_conv_gen_*, motion_spatial_down_*, "
                                       "motion_spatial_grad_*) can be bound in place: " + k);
  }
  s.set = true;
  return OFFK_OK;
}

int offk_missing_weights(const offk_handle* h, char* buf, size_t buflen) {
  if (!h) return OFFK_ERR_INVALID;
  int missing = 0;
  for (const Slot& s : h->slots)
    if (!s.set) {
      if (missing == 0 && buf && buflen) snprintf(buf, buflen, "%s", s.key.c_str());
      ++missing;
    }
  return missing;
}

size_t offk_workspace_bytes(const offk_handle* h) { return h ? h->ws_bytes : 0; }

int offk_workspace_region(const offk_handle* h, const char* name, size_t* offset_bytes, size_t* nbytes) {
  if (!h || !name) return OFFK_ERR_INVALID;
  auto it = h->regions.find(name);
  if (it == h->regions.end()) { h->err = std::string("unknown workspace region: ") + name; return OFFK_ERR_INVALID; }
  if (offset_bytes) *offset_bytes = it->second.first;
  if (nbytes) *nbytes = it->second.second;
  return OFFK_OK;
}

int offk_set_profiling(offk_handle* h, int enable) {
  if (!h) return OFFK_ERR_INVALID;
  if (enable < 0 || enable > 2) return fail(h, OFFK_ERR_INVALID, "offk_set_profiling: 0 off, 1 per-stage events, 2 per-launch trace");
  h->profiling = enable;
  return OFFK_OK;
}

int offk_launch_times(offk_handle* h, char* names, size_t names_len, double* ms, int64_t* calls, int max_entries, int reset) {
  if (!h || max_entries < 0 || (max_entries > 0 && (!ms || !calls))) return fail(h, OFFK_ERR_INVALID, "offk_launch_times: bad argument");
  DeviceGuard guard(h->cfg.device);
  const int n = (int)h->tr_names.size();
  std::vector<double> sum(n, 0.0);
  std::vector<int64_t> cnt(n, 0);
  if (!h->tr_marks.empty()) HIP_TRY(h, hipEventSynchronize(h->tr_events[h->tr_marks.size() - 1]));
  for (size_t i = 0; i + 1 < h->tr_marks.size(); ++i) {
    const int id = h->tr_marks[i];
    if (id < 0) continue;
    float t = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&t, h->tr_events[i], h->tr_events[i + 1]));
    sum[id] += t;
    cnt[id] += 1;
  }
  std::string all;
  for (int i = 0; i < n && i < max_entries; ++i) {
    ms[i] = sum[i];
    calls[i] = cnt[i];
    all += h->tr_names[i];
    all += '\n';
  }
  if (names && names_len) snprintf(names, names_len, "%s", all.c_str());
  if (reset) { h->tr_marks.clear(); h->tr_names.clear(); }
  return n;
}

int offk_stage_times(offk_handle* h, double ms[OFFK_NUM_STAGES], int64_t calls[OFFK_NUM_STAGES], int reset) {
  if (!h || !ms || !calls) return fail(h, OFFK_ERR_INVALID, "offk_stage_times: null argument");
  DeviceGuard guard(h->cfg.device);
  const size_t per = OFFK_NUM_STAGES + 1;
  for (int s = 0; s < OFFK_NUM_STAGES; ++s) { ms[s] = 0.0; calls[s] = 0; }
  for (size_t g = 0; g < h->ev_used; ++g) {
    hipEvent_t* ev = &h->events[g * per];
    HIP_TRY(h, hipEventSynchronize(ev[OFFK_NUM_STAGES]));
    for (int s = 0; s < OFFK_NUM_STAGES; ++s) {
      float t = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&t, ev[s], ev[s + 1]));
      ms[s] += t;
      calls[s] += 1;
    }
  }
  if (reset) h->ev_used = 0;
  return OFFK_OK;
}

int offk_pw_reduce(offk_handle* h, void* stream, int site, const float* feat, float* G, float* D) {
  if (!h || site < 0 || site >= kNumSites || !feat || !G || !D) return fail(h, OFFK_ERR_INVALID, "offk_pw_reduce: bad argument");
  TRY(site_weights_ready(h, site, true, false));
  DeviceGuard guard(h->cfg.device);
  PwParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.nsites = 1; pp.L = h->cfg.length; pp.P = h->P; pp.slice_mode = h->cfg.slice_mode;
  pp.nhwc = h->cfg.feat_layout == OFFK_FEAT_NHWC;
  pp.zeros = h->zero_page;
  TRY(finalize_pw(h, static_cast<hipStream_t>(stream)));
  fill_pw_site(h, site, whole_map(site, feat), G, D, &pp.s[0]);
  pp.total_blocks = pw_blocks_for(pp.s[0].M);
  HIP_TRY(h, pw_reduce_launch(pp, static_cast<hipStream_t>(stream)));
  return OFFK_OK;
}

int offk_sobel_tdiff(offk_handle* h, void* stream, int site, const float* G, const float* D, float* M, int m_cstride,
                     int m_coff, int algo) {
  if (!h || site < 0 || site >= kNumSites || !G || !D || !M) return fail(h, OFFK_ERR_INVALID, "offk_sobel_tdiff: bad argument");
  if (m_cstride % 4 || m_coff % 4 || m_coff < 0 || m_coff + kUnitCh > m_cstride || algo < 0 || algo > 5)
    return fail(h, OFFK_ERR_INVALID, "offk_sobel_tdiff: need 16-byte aligned channel slice of 160 channels inside m_cstride, algo in 0..5");
  TRY(site_weights_ready(h, site, false, true));
  DeviceGuard guard(h->cfg.device);
  StParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.nsites = 1; sp.B = h->cfg.batch; sp.L = h->cfg.length;
  sp.zeros = h->zero_page; sp.drop_scale = 1.f;
  sp.taps4 = h->cfg.variant == OFFK_VARIANT_DIAG_SOBEL && h->sobel_taps4;
  fill_st_site(h, site, G, D, M, m_cstride, m_coff, &sp.s[0]);
  if (algo >= 4) sp.s[0].tchunks = st_tchunks_flat(kSites[site].H, h->cfg.length);
  sp.total_s = h->P * sp.s[0].strips;
  sp.total_t = h->cfg.batch * sp.s[0].tchunks;
  HIP_TRY(h, sobel_tdiff_launch(sp, algo, static_cast<hipStream_t>(stream)));
  return OFFK_OK;
}

int offk_sobel_tdiff_all(offk_handle* h, void* stream, void* workspace, int algo) {
  if (!h || !workspace || algo < 0 || algo > 5) return fail(h, OFFK_ERR_INVALID, "offk_sobel_tdiff_all: bad argument");
  for (int s = 0; s < kNumSites; ++s) {
    int rc = site_weights_ready(h, s, false, true);
    if (rc != OFFK_OK) return rc;
  }
  DeviceGuard guard(h->cfg.device);
  return run_sobel_tdiff_all(h, static_cast<hipStream_t>(stream), workspace, algo);
}

int offk_off_units(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace) {
  if (!h || !feats || !workspace) return fail(h, OFFK_ERR_INVALID, "offk_off_units: null argument");
  for (int s = 0; s < kNumSites; ++s) {
    if (!feats[s]) return fail(h, OFFK_ERR_INVALID, "offk_off_units: null feature map");
    TRY(site_weights_ready(h, s, true, true));
  }
  DeviceGuard guard(h->cfg.device);
  offk_feat_parts parts[kNumSites];
  for (int s = 0; s < kNumSites; ++s) parts[s] = whole_map(s, feats[s]);
  return run_off_units(h, static_cast<hipStream_t>(stream), parts, workspace, nullptr);
}

int offk_off_units_fused(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace) {
  if (!h || !feats || !workspace) return fail(h, OFFK_ERR_INVALID, "offk_off_units_fused: null argument");
  for (int s = 0; s < kNumSites; ++s) {
    if (!feats[s]) return fail(h, OFFK_ERR_INVALID, "offk_off_units_fused: null feature map");
    TRY(site_weights_ready(h, s, true, true));
  }
  DeviceGuard guard(h->cfg.device);
  offk_feat_parts parts[kNumSites];
  for (int s = 0; s < kNumSites; ++s) parts[s] = whole_map(s, feats[s]);
  if (h->fused_units && h->cfg.feat_layout != OFFK_FEAT_NHWC) return run_off_units_fused(h, static_cast<hipStream_t>(stream), parts, workspace, nullptr);
  return run_off_units(h, static_cast<hipStream_t>(stream), parts, workspace, nullptr);
}

int offk_forward(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], float* out7, float* out14,
                 float* out28, void* workspace) {
  if (!h || !feats) return fail(h, OFFK_ERR_INVALID, "offk_forward: null argument");
  offk_feat_parts parts[kNumSites];
  for (int s = 0; s < kNumSites; ++s) {
    if (!feats[s]) return fail(h, OFFK_ERR_INVALID, "offk_forward: null feature map");
    parts[s] = whole_map(s, feats[s]);
  }
  return offk_forward_parts(h, stream, parts, out7, out14, out28, workspace);
}

int offk_forward_parts(offk_handle* h, void* stream, const offk_feat_parts feats[OFFK_NUM_SITES], float* out7, float* out14,
                       float* out28, void* workspace) {
  if (!h || !feats || !out7 || !out14 || !workspace) return fail(h, OFFK_ERR_INVALID, "offk_forward: null argument");
  TRY(check_parts(h, feats));
  TRY(check_ready(h));
  DeviceGuard guard(h->cfg.device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  void* ws = workspace;
  const int P = h->P, ncls = h->cfg.num_classes;

  hipEvent_t* ev = nullptr;
  if (h->profiling == 1) {
    const size_t per = OFFK_NUM_STAGES + 1;
    if ((h->ev_used + 1) * per > h->events.size() && h->events.size() < 4096 * per) {
      for (size_t i = 0; i < per; ++i) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        h->events.push_back(e);
      }
    }
    if ((h->ev_used + 1) * per <= h->events.size()) ev = &h->events[h->ev_used++ * per];
  }

  if (h->fused_units && h->cfg.feat_layout != OFFK_FEAT_NHWC) TRY(run_off_units_fused(h, st, feats, ws, ev));
  else TRY(run_off_units(h, st, feats, ws, ev));

  h->cur_splitk = region(h, ws, "splitk");
  float* F28 = region(h, ws, "fusion_28");
  float* F14 = region(h, ws, "fusion_14");
  float* F7 = region(h, ws, "fusion_7");
  const int RI = OFFK_CONV_RELU_IN_, RP = OFFK_CONV_RELU_PRE_, RO = OFFK_CONV_RELU_POST_;

  TRY(finalize_merged(h, st));
  TRY(finalize_wino(h, st));
  const bool cons = h->cfg.consensus == OFFK_CONSENSUS_AVG;
  float* l7 = cons ? region(h, ws, "logit_7") : out7;
  float* l14 = cons ? region(h, ws, "logit_14") : out14;
  float* l28 = cons ? region(h, ws, "logit_28") : out28;
  const int n = P;                 // every launch below covers all P pairs (the buffers are pair-major)
  hipStream_t s = st;
  // pool_kernel + fc_kernel: the heads of the paths that cannot fold the average pool into the producing conv (LDS-patch tiles
  // of the bf16x3 plans, OFFK_FOLD_POOL=0)
  auto run_head = [&](int k, const float* x, int x_cs, int x_coff, int Hh, int C, int maxpool, const char* pooled_name,
                      float* logits) -> int {
    float* pooled = region(h, ws, pooled_name);
    { int rc = trace_mark(h, s, k == 0 ? "head_7 (pool + fc)" : k == 1 ? "head_28 (pool + fc)" : "head_14 (pool + fc)"); if (rc != OFFK_OK) return rc; }
    hipError_t e = pool_launch(x, x_cs, x_coff, n, Hh, Hh, C, maxpool, pooled, s);
    if (e == hipSuccess) e = fc_launch(pooled, n, C, h->fc_w[k], h->fc_b[k], ncls, logits, s);
    if (e != hipSuccess) return fail_hip(h, e, "head");
    return OFFK_OK;
  };
  // the folded-pool FCs of the heads are collected and launched together behind the last stage (heads.hip: fc_pooled_multi_kernel)
  FcPooledJobs fcj{};
  fcj.n_img = n; fcj.ncls = ncls;
  auto fc_pooled_later = [&](const float* part, int hw, int tiles, int C, int k, float* logits) {
    fcj.job[fcj.njobs++] = FcPooledJob{part, hw, tiles, C, h->fc_w[k], h->fc_b[k], logits};
  };
  float *xt = region(h, ws, "xt_28"), *t1 = region(h, ws, "t1_28");
  float *sa = region(h, ws, "sa_28"), *sb = region(h, ws, "sb_28");
  float *xu = region(h, ws, "xu_14"), *u1 = region(h, ws, "u1_14"), *s14 = region(h, ws, "sa_14");   // xu = [u2 | x1]
  float *xv = region(h, ws, "xv_7"), *v1 = region(h, ws, "v1_7"), *s7 = region(h, ws, "sum_7");   // xv = [v2 | x2]
  // the batched GEMMs of a conv on a Winograd path: ONE launch of the 1x1 kernel, the K groups ride on gridDim.y (four launches
  // left the short groups alone on the chip: slower than not skipping their zero products)
  auto wino_gemms = [&](const char* key, const WinoGroup* grp, int ngrp, int npoints, int rows, int Ci, int Co, const float* V,
                        const float* U, float* M) -> int {
    const char* why = nullptr;
    const void* planes = nullptr;
    for (int k = 0; k < 6; ++k) if (U == h->wino_u[k]) planes = h->wino_us[k];
    if (U == h->wino_u7) planes = h->wino_u7s;
    hipError_t e = wino_gemms_launch(grp, ngrp, npoints, rows, Ci, Co, V, U, M, h->wino_gemm, s, &why, planes);
    if (e != hipSuccess) return fail(h, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, std::string(key) + " (winograd): " + (why ? why : hipGetErrorString(e)));
    return OFFK_OK;
  };
  // 3x3 / stride 1 conv on 7x7 maps (phases = 1) or the 5x5 / stride 2 conv on 14x14 maps in polyphase form (phases = 4) on the
  // Winograd path: input transform, batched GEMMs, output transform with the conv's epilogue (winograd.hip)
  const bool wino = h->winograd;
  float* const wino_V = wino ? region(h, ws, "wino_v") : nullptr;
  float* const wino_M = wino ? region(h, ws, "wino_m") : nullptr;
  // the three steps of a conv on that path: V = B^T x B (wino_V), M = V U per point (wino_M), y = epilogue(A^T M A)
  auto wino_in = [&](ConvId id, View x) -> int {
    const ConvSpec& c = kConvs[id];
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: input transform]").c_str()));
    HIP_TRY(h, wino_input_launch(x.p, x.cs, x.coff, n, c.Ci, c.K == 5 ? 4 : 1, wino_V, s));
    return OFFK_OK;
  };
  auto wino_mm = [&](ConvId id, int uidx) -> int {
    const ConvSpec& c = kConvs[id];
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: 121 GEMMs]").c_str()));
    WinoGroup grp[4];
    const int ngrp = wino_groups(c.K == 5 ? 4 : 1, n, c.Ci, c.Co, grp);
    return wino_gemms(c.key, grp, ngrp, kWinoPoints, n, c.Ci, c.Co, wino_V, h->wino_u[uidx], wino_M);
  };
  auto wino_out = [&](ConvId id, const float* res, int res_cs, int res_coff, int flags, float* y, int y_cs, int y_coff, float* pool_t) -> int {
    const ConvSpec& c = kConvs[id];
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: output transform]").c_str()));
    HIP_TRY(h, wino_output_launch(wino_M, n, c.Co, c.K == 5 ? 4 : 1, h->conv_b[id], res, res_cs, res_coff, flags, y, y_cs, y_coff, pool_t, s));
    return OFFK_OK;
  };
  auto wino_conv = [&](ConvId id, int uidx, View x, const float* res, int res_cs, int res_coff, int flags, float* y, int y_cs,
                       int y_coff, float* pool_t) -> int {
    TRY(wino_in(id, x));
    TRY(wino_mm(id, uidx));
    return wino_out(id, res, res_cs, res_coff, flags, y, y_cs, y_coff, pool_t);
  };
  // What sits between two convs on that path, in ONE launch (wino_mid.hip): the output transform (+ bias, ReLU) of conv `a` from
  // wino_M, optionally the 1x1 conv `mid` (+ bias, ReLU), the input transform of the conv behind into wino_V.  xa: where the
  // activation of conv `a` is ALSO stored (the merged convs read x1 / x2 from there later); nullptr: nowhere.
  auto wino_between = [&](ConvId a, const ConvId* mid, float* xa, int xa_cs, int xa_coff, const char* name) -> int {
    WinoMidArgs m;
    m.M = wino_M; m.bias_in = h->conv_b[a]; m.phases_in = kConvs[a].K == 5 ? 4 : 1;
    m.x = xa; m.x_cs = xa_cs; m.x_coff = xa_coff;
    m.w1 = mid ? h->conv_w[*mid] : nullptr; m.b1 = mid ? h->conv_b[*mid] : nullptr;
    m.Cin = kConvs[a].Co; m.Cmid = mid ? kConvs[*mid].Co : kConvs[a].Co; m.n_img = n;
    m.V = wino_V;
    if (mid) m.w1p = *mid == C1_14A ? h->mid_ws[0] : *mid == C1_7 ? h->mid_ws[1] : nullptr;
    TRY(trace_mark(h, s, name));
    HIP_TRY(h, wino_mid_launch(m, s));
    return OFFK_OK;
  };
  const bool mid = wino && h->wino_mid;
  // ---- fusion @28 -> 14x14 (RGB_OFF.py:655-685) -----------------------------------
  // xt = [t2 | x0] per pixel: c3(t2) + branch(x0) (:663-666) is then ONE 1x1 conv over 128 channels
  // :657 x0, pre-ReLU kept for the branch.  fp32: polyphase Winograd F(5x5, 4x4) (winograd7.hip) -- input transform, 64 batched
  // GEMMs in four K groups as ONE launch of the 1x1 kernel, output transform
  // (from P = 12 pairs -- B = 2: 0.495 against 0.503 ms, B = 8: 0.813 against 0.874, B = 64: 3.88 against 4.24; B = 1: equal)
  if (h->wino_u7 && wino && h->wino_7x7 && P >= h->wino7_min_p) {
    const ConvSpec& c = kConvs[C_T28];
    const int T = kWino7Tiles * n;
    // (the input transform INSIDE the GEMM kernel was built and measured in round 4 -- tools/experiments/winograd7_fused.hip, out of the
    //  product build since round 5: 0.84 ms against 0.54 ms for these two launches; profiles/r04/wino7_fused_attempt.txt)
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: input transform]").c_str()));
    HIP_TRY(h, wino7_input_launch(F28, 320, 0, n, c.Ci, wino_V, s));
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: 64 GEMMs]").c_str()));
    WinoGroup grp[4];
    const int ngrp = wino7_groups(T, c.Ci, c.Co, grp);
    TRY(wino_gemms(c.key, grp, ngrp, kWino7Points, T, c.Ci, c.Co, wino_V, h->wino_u7, wino_M));
    TRY(trace_mark(h, s, (std::string(c.key) + " [winograd: output transform]").c_str()));
    HIP_TRY(h, wino7_output_launch(wino_M, n, c.Co, h->conv_b[C_T28], 0, xt, 128, 64, s));
  } else {
    TRY(conv(h, s, C_T28, n, 28, View{F28, 320, 0}, nullptr, 0, 0, 0, xt, 128, 64));
  }
  // 1x1 -> 3x3 -> 1x1 (+ residual) as ONE launch per chain (a block owns half an image, t1 / t2 stay in LDS)
  bool split_branch = false;
  auto chain = [&](const char* name, const float* x, int x_cs, int x_coff, int Cin, int relu_in, ConvId c1, ConvId c2,
                   const float* w3, const float* b3, int K3, const float* res, float* y, int y_cs, int y_coff) -> int {
    ChainArgs a;
    a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.Cin = Cin; a.relu_in = relu_in;
    a.w1 = h->conv_w[c1]; a.b1 = h->conv_b[c1]; a.w2 = h->conv_w[c2]; a.b2 = h->conv_b[c2];
    a.u2 = h->chain_wino ? h->chain_u2[c2 == C2_28A ? 0 : c2 == C2_28B ? 1 : 2] : nullptr;
    a.w3 = w3; a.b3 = b3; a.K3 = K3;
    a.res = res; a.res_cs = 256; a.res_coff = 0;
    a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
    a.n_img = n; a.relu_out = 1;
    const unsigned long long xb = ((unsigned long long)n * 196 * x_cs - x_coff) * 4ull;
    a.x_bytes = xb < 0x7fffffffull ? (unsigned)xb : 0u;
    { int rc = trace_mark(h, s, name); if (rc != OFFK_OK) return rc; }
    const char* why = nullptr;
    const int ck = c2 == C2_28A ? 0 : c2 == C2_28B ? 1 : 2;
    a.w1p = h->chain_ws[ck][0]; a.w2p = h->chain_ws[ck][1]; a.w3p = h->chain_ws[ck][2];
    if (split_branch) { a.wbp = h->chain_ws[0][3]; a.bbr = h->conv_b[CB_28A]; }
    hipError_t e = chain14_split_supported(a) ? chain14_split_launch(a, s, &why) : chain14_launch(a, s, &why);
    if (e != hipSuccess) return fail(h, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, std::string(name) + ": " + (why ? why : hipGetErrorString(e)));
    return OFFK_OK;
  };
  // (from P = 72 pairs: a chain block walks its three convs alone -- 45 us per launch however few blocks there are; B = 8: three
  //  convs per chain 0.885 ms per forward against 0.90, B = 16: 1.37 against 1.345; OFFK_CHAIN=<pairs> moves the gate)
  const bool chained = h->chain && P >= h->chain_min_p && (unsigned long long)n * 196 * 256 * 4ull < 0x7fffffffull;
  if (chained) {
    if (h->chain_ws[0][3]) {
      // split-fp32 (chain_split.hip, BR form): c3 with K3 = 64, the branch 1x1 on the pre-ReLU chain input inside the kernel (its output
      // passes through y as the chain's residual) -- RGB_OFF.py:663-667; in the fp32 kernel the branch is merged into c3's K
      split_branch = true;
      TRY(chain("chain_28a = motion_conv1_trans_28a + motion_conv2_trans_28a + merged_28a", xt, 128, 64, 64, 1, C1_28A, C2_28A, h->conv_w[C3_28A],
                h->conv_b[C3_28A], 64, nullptr, sa, 256, 0));
      split_branch = false;
    } else {
      TRY(chain("chain_28a = motion_conv1_trans_28a + motion_conv2_trans_28a + merged_28a", xt, 128, 64, 64, 1, C1_28A, C2_28A, h->merged_w[0], h->merged_b[0], 128,
                nullptr, sa, 256, 0));                                                              // :658-667
    }
    TRY(chain("chain_28b = motion_conv1_trans_28b + motion_conv2_trans_28b + motion_conv3_trans_28b", sa, 256, 0, 256, 0, C1_28B, C2_28B, h->conv_w[C3_28B], h->conv_b[C3_28B], 64, sa, sb, 256, 0));   // :670-676
    TRY(chain("chain_28c = motion_conv1_trans_28c + motion_conv2_trans_28c + motion_conv3_trans_28c", sb, 256, 0, 256, 0, C1_28C, C2_28C, h->conv_w[C3_28C], h->conv_b[C3_28C], 64, sb, F14, 1056, 800)); // :679-685 -> cat at :760
  } else {
    TRY(conv(h, s, C1_28A, n, 14, View{xt, 128, 64}, nullptr, 0, 0, RI | RP, t1, 64, 0));         // :658-660
    TRY(conv(h, s, C2_28A, n, 14, View{t1, 64, 0}, nullptr, 0, 0, RP, xt, 128, 0));               // :661-662 t2
    TRY(conv_merged(h, s, 0, n, 14, View{xt, 128, 0}, RO, sa, 256, 0));                            // :663-667
    TRY(conv(h, s, C1_28B, n, 14, View{sa, 256, 0}, nullptr, 0, 0, RP, t1, 64, 0));               // :670-671
    TRY(conv(h, s, C2_28B, n, 14, View{t1, 64, 0}, nullptr, 0, 0, RP, xt, 128, 0));               // :672-673
    TRY(conv(h, s, C3_28B, n, 14, View{xt, 128, 0}, sa, 256, 0, RO, sb, 256, 0));                  // :674-676
    TRY(conv(h, s, C1_28C, n, 14, View{sb, 256, 0}, nullptr, 0, 0, RP, t1, 64, 0));               // :679-680
    TRY(conv(h, s, C2_28C, n, 14, View{t1, 64, 0}, nullptr, 0, 0, RP, xt, 128, 0));               // :681-682
    TRY(conv(h, s, C3_28C, n, 14, View{xt, 128, 0}, sb, 256, 0, RO, F14, 1056, 800));              // :683-685 -> cat at :760
  }
  if (ev) HIP_TRY(h, hipEventRecord(ev[3], s));
  if (out28) {   // 28-head (:782-787): only reads sum_28c
    if (h->fold_pool) {     // pool-row partial sums + the FC as an MFMA GEMM (heads.hip) instead of pool_kernel + fc_kernel
      float* pp = region(h, ws, "poolpart_28");
      TRY(trace_mark(h, s, "head_28 (max pool rows)"));
      HIP_TRY(h, maxpool_rows_launch(F14, 1056, 800, n, 14, 14, 256, pp, s));
      fc_pooled_later(pp, 49, 1, 256, 1, l28);
    } else {
      TRY(run_head(1, F14, 1056, 800, 14, 256, 1, "pooled_28", l28));
    }
  }
  // ---- fusion @14 -> 7x7 (RGB_OFF.py:759-780) ---------------------------------------
  // (from P = 40 pairs: below, its 132-K-tile GEMMs have too few row tiles to fill the chip and the split-K direct conv wins --
  // B = 1: 0.435 vs 0.50 ms, B = 4: 0.672 vs 0.695, B = 8: 0.892 vs 0.874, B = 12: 1.157 vs 1.12; OFFK_WINOGRAD_5X5=<pairs> moves the gate)
  const bool w5 = wino && h->wino_5x5 && P >= h->wino5_min_p;
  if (w5 && mid) {      // :762-767: x1 = relu(conv5x5(F14)) -> c1_14a -> c2_14a, the 1x1 conv between the two Winograd GEMM launches
    const ConvId c1 = C1_14A;
    TRY(wino_in(C_T14, View{F14, 1056, 0}));
    TRY(wino_mm(C_T14, 5));
    TRY(wino_between(C_T14, &c1, xu, 256, 128, "motion_conv_trans_14 out + motion_conv1_trans_14a + motion_conv2_trans_14a in [winograd: between]"));
    TRY(wino_mm(C2_14A, 3));
    TRY(wino_out(C2_14A, nullptr, 0, 0, RP, xu, 256, 0, nullptr));                                 // :766-767 u2
  } else {
    if (w5) TRY(wino_conv(C_T14, 5, View{F14, 1056, 0}, nullptr, 0, 0, RP, xu, 256, 128, nullptr));   // :762-763 x1 (polyphase)
    else TRY(conv(h, s, C_T14, n, 14, View{F14, 1056, 0}, nullptr, 0, 0, RP, xu, 256, 128));      // :762-763 x1
    TRY(conv(h, s, C1_14A, n, 7, View{xu, 256, 128}, nullptr, 0, 0, RP, u1, 128, 0));             // :764-765
    if (wino) TRY(wino_conv(C2_14A, 3, View{u1, 128, 0}, nullptr, 0, 0, RP, xu, 256, 0, nullptr));  // :766-767 u2
    else TRY(conv(h, s, C2_14A, n, 7, View{u1, 128, 0}, nullptr, 0, 0, RP, xu, 256, 0));          // :766-767 u2
  }
  TRY(conv_merged(h, s, 1, n, 7, View{xu, 256, 0}, RO, s14, 512, 0));                              // :768-771
  TRY(conv(h, s, C1_14B, n, 7, View{s14, 512, 0}, nullptr, 0, 0, RP, u1, 128, 0));                // :773-774
  if (mid) {            // :775-780: c2_14b's output feeds c3_14b alone -- output transform, ReLU and input transform in one launch
    TRY(wino_in(C2_14B, View{u1, 128, 0}));
    TRY(wino_mm(C2_14B, 4));
    TRY(wino_between(C2_14B, nullptr, nullptr, 0, 0, "motion_conv2_trans_14b out + motion_conv3_trans_14b in [winograd: between]"));
  } else if (wino) TRY(wino_conv(C2_14B, 4, View{u1, 128, 0}, nullptr, 0, 0, RP, xu, 256, 0, nullptr));  // :775-776
  else TRY(conv(h, s, C2_14B, n, 7, View{u1, 128, 0}, nullptr, 0, 0, RP, xu, 256, 0));            // :775-776
  // (fold: the conv's epilogue also leaves per-slab column sums of sum_14b: the 14-head's average pool)
  auto generic = [](int cfg) { return cfg != 6 && cfg != 7 && cfg != 10; };      // the LDS-patch kernels have no pooling epilogue
  const bool fold = h->fold_pool && generic(h->conv_cfg[C3_14B]) && generic(h->merged_cfg[2]);
  float* pp14 = region(h, ws, "poolpart_14");
  float* pp7 = region(h, ws, "poolpart_7");
  float* pp14t = wino ? region(h, ws, "poolpart_14t") : nullptr;
  const bool fold14t = wino && h->fold_pool;
  if (mid) {
    TRY(wino_mm(C3_14B, 0));
    TRY(wino_out(C3_14B, s14, 512, 0, RP | RO, F7, 832, 320, fold14t ? pp14t : nullptr));          // :777-780 -> cat at :832
  } else if (wino) {
    TRY(wino_conv(C3_14B, 0, View{xu, 256, 0}, s14, 512, 0, RP | RO, F7, 832, 320, fold14t ? pp14t : nullptr));   // :777-780 -> cat at :832
  } else {
    h->cur_pool_part = fold ? pp14 : nullptr;
    int rc_ = conv(h, s, C3_14B, n, 7, View{xu, 256, 0}, s14, 512, 0, RP | RO, F7, 832, 320);   // :777-780 -> cat at :832
    h->cur_pool_part = nullptr;
    if (rc_ != OFFK_OK) return rc_;
  }
  if (ev) HIP_TRY(h, hipEventRecord(ev[4], s));
  // 14-head (:789-793): only reads sum_14b
  if (fold14t) {
    fc_pooled_later(pp14t, 49, 1, 512, 2, l14);
  } else if (fold && !wino) {
    fc_pooled_later(pp14, 49, 0, 512, 2, l14);
  } else {
    TRY(run_head(2, F7, 832, 320, 7, 512, 0, "pooled_14", l14));
  }
  // ---- fusion @7 (RGB_OFF.py:831-841) -------------------------------------------------
  if (mid) {            // :833-838: x2 = relu(conv3x3(F7)) -> c1 -> c2, as fusion@14's first three convs
    const ConvId c1 = C1_7;
    TRY(wino_in(C_T7, View{F7, 832, 0}));
    TRY(wino_mm(C_T7, 1));
    TRY(wino_between(C_T7, &c1, xv, 512, 256, "motion_conv_trans out + motion_conv1_trans + motion_conv2_trans in [winograd: between]"));
    TRY(wino_mm(C2_7, 2));
    TRY(wino_out(C2_7, nullptr, 0, 0, RP, xv, 512, 0, nullptr));                                   // :837-838 v2
  } else {
    if (wino) TRY(wino_conv(C_T7, 1, View{F7, 832, 0}, nullptr, 0, 0, RP, xv, 512, 256, nullptr));  // :833-834 x2
    else TRY(conv(h, s, C_T7, n, 7, View{F7, 832, 0}, nullptr, 0, 0, RP, xv, 512, 256));          // :833-834 x2
    TRY(conv(h, s, C1_7, n, 7, View{xv, 512, 256}, nullptr, 0, 0, RP, v1, 256, 0));               // :835-836
    if (wino) TRY(wino_conv(C2_7, 2, View{v1, 256, 0}, nullptr, 0, 0, RP, xv, 512, 0, nullptr));  // :837-838 v2
    else TRY(conv(h, s, C2_7, n, 7, View{v1, 256, 0}, nullptr, 0, 0, RP, xv, 512, 0));            // :837-838 v2
  }
  h->cur_pool_part = fold ? pp7 : nullptr;
  { int rc_ = conv_merged(h, s, 2, n, 7, View{xv, 512, 0}, 0, s7, 1024, 0);                        // :839-841 (no ReLU)
    h->cur_pool_part = nullptr;
    if (rc_ != OFFK_OK) return rc_; }
  if (ev) HIP_TRY(h, hipEventRecord(ev[5], s));
  // ---- 7-head (:843-847)
  if (fold) {
    fc_pooled_later(pp7, 49, 0, 1024, 0, l7);
  } else {
    TRY(run_head(0, s7, 1024, 0, 7, 1024, 0, "pooled_7", l7));
  }
  if (fcj.njobs > 0) {
    TRY(trace_mark(h, s, "heads (fc on folded pools, one launch)"));
    HIP_TRY(h, fc_pooled_multi_launch(fcj, s));
  }
  if (cons) {
    const int B = h->cfg.batch, T = h->cfg.length - 1;
    TRY(trace_mark(h, st, "consensus (K6)"));
    const float* const cx[3] = {l7, l14, out28 ? l28 : l14};                                      // Flow_OFF.py:874-876, one launch
    float* const co[3] = {out7, out14, out28 ? out28 : out14};
    HIP_TRY(h, consensus_multi_launch(cx, co, out28 ? 3 : 2, B, T, ncls, st));
  }
  if (ev) HIP_TRY(h, hipEventRecord(ev[6], st));
  TRY(trace_mark(h, st, nullptr));
  return OFFK_OK;
}

// ---- training side of the OFF units -----------------------------------
size_t offk_train_workspace_bytes(const offk_handle* h) { return h ? h->train_ws_bytes : 0; }

size_t offk_unit_grad_floats(const offk_handle* h) { return h ? h->grad_floats : 0; }

int offk_unit_grad_slot(const offk_handle* h, const char* key, size_t* offset_floats, size_t* count) {
  if (!h || !key) return OFFK_ERR_INVALID;
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);
  auto it = h->grad_slots.find(k);
  if (it == h->grad_slots.end()) { h->err = "no unit gradient for key: " + k; return OFFK_ERR_INVALID; }
  if (offset_floats) *offset_floats = it->second.first;
  if (count) *count = it->second.second;
  return OFFK_OK;
}

int offk_off_units_train(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace,
                         uint64_t drop_seed, double drop_p) {
  if (!h || !feats || !workspace) return fail(h, OFFK_ERR_INVALID, "offk_off_units_train: null argument");
  DropCfg drop;
  TRY(make_drop(h, drop_seed, drop_p, &drop));
  for (int s = 0; s < kNumSites; ++s) {
    if (!feats[s]) return fail(h, OFFK_ERR_INVALID, "offk_off_units_train: null feature map");
    TRY(site_weights_ready(h, s, true, true));
  }
  DeviceGuard guard(h->cfg.device);
  offk_feat_parts parts[kNumSites];
  for (int s = 0; s < kNumSites; ++s) parts[s] = whole_map(s, feats[s]);
  return run_off_units(h, static_cast<hipStream_t>(stream), parts, workspace, nullptr, drop);
}

int offk_off_units_backward(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES],
                            const offk_grad_view gm[OFFK_NUM_SITES], void* workspace, uint64_t drop_seed, double drop_p,
                            float* grads, int accumulate) {
  if (!h || !feats || !gm || !workspace || !grads) return fail(h, OFFK_ERR_INVALID, "offk_off_units_backward: null argument");
  if (h->cfg.feat_layout == OFFK_FEAT_NHWC) return fail(h, OFFK_ERR_INVALID, "offk_off_units_backward: NCHW feature maps only");
  DropCfg drop;
  TRY(make_drop(h, drop_seed, drop_p, &drop));
  for (int s = 0; s < kNumSites; ++s) {
    if (!feats[s] || !gm[s].data) return fail(h, OFFK_ERR_INVALID, "offk_off_units_backward: null feature map or gradient");
    if (gm[s].cstride < gm[s].coff + kUnitCh || (gm[s].cstride & 3) || (gm[s].coff & 3) || gm[s].coff < 0)
      return fail(h, OFFK_ERR_INVALID, "offk_off_units_backward: gradient view needs 16-byte aligned 160 channels inside cstride");
    TRY(site_weights_ready(h, s, false, true));
  }
  DeviceGuard guard(h->cfg.device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  void* ws = workspace;
  const bool learned_dw = h->cfg.variant != OFFK_VARIANT_DIAG_SOBEL;
  auto reg = [&](const char* prefix, int s) { return region(h, ws, (std::string(prefix) + kSites[s].name).c_str()); };

  // K2b: dM -> dGpre, dD, depthwise partials
  UbParams up;
  memset(&up, 0, sizeof(up));
  up.nsites = kNumSites; up.B = h->cfg.batch; up.L = h->cfg.length; up.tpix = kUbTpix;
  up.drop_thresh = drop.thresh; up.drop_scale = drop.scale; up.zeros = h->zero_page;
  int sblk = 0, tblk = 0;
  int nsblocks[kNumSites];
  for (int s = 0; s < kNumSites; ++s) {
    UbSite& u = up.s[s];
    u.G = reg("G_", s); u.D = reg("D_", s);
    u.dw = learned_dw ? (h->bnd_dw_w[s] ? h->bnd_dw_w[s] : h->dw_w[s]) : h->sobel_w;
    u.dw_ref = learned_dw && h->bnd_dw_w[s] != nullptr;
    u.gm = gm[s].data; u.gm_cs = gm[s].cstride; u.gm_coff = gm[s].coff;
    u.dG = reg("dG_", s); u.dD = reg("dD_", s);
    u.dw_part = learned_dw ? reg("dwp_", s) : nullptr;
    u.drop_base = drop_stream_base(drop.seed, s);
    u.H = kSites[s].H;
    ub_plan(u.H, &u.strips, &u.rows);
    u.tchunks = (u.H * u.H + kUbTpix - 1) / kUbTpix;
    u.s_begin = sblk; u.t_begin = tblk;
    nsblocks[s] = h->P * u.strips;
    sblk += nsblocks[s];
    tblk += h->cfg.batch * u.tchunks;
  }
  up.total_s = sblk; up.total_t = tblk;
  HIP_TRY(h, units_bwd_launch(up, st));

  // K1b: weight-gradient GEMM, per-chunk slabs
  WgParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.nsites = kNumSites; wp.L = h->cfg.length; wp.P = h->P; wp.slice_mode = h->cfg.slice_mode; wp.kt_per_blk = h->wg_kpb;
  wp.precision = 0;
  wp.dbg = 0;   // ablation bits of the tools build only
  wp.zeros = h->zero_page;
  WrParams rp;
  memset(&rp, 0, sizeof(rp));
  rp.nsites = kNumSites; rp.accumulate = accumulate ? 1 : 0;
  int blk = 0;
  for (int i = 0; i < kNumSites; ++i) {
    const int s = kPwOrder[i];
    WgSite& w = wp.s[i];
    w.xp[0] = feats[s]; w.cp[0] = kSites[s].C; w.nparts = 1;
    w.dG = reg("dG_", s); w.dD = reg("dD_", s); w.slab = reg("wgs_", s); w.bpart = reg("wgb_", s);
    w.C = kSites[s].C; w.HW = kSites[s].H * kSites[s].H;
    w.tpf = (w.HW + 31) / 32; w.kt_total = h->N * w.tpf;
    w.ntiles = (w.C + 127) / 128; w.nchunks = (w.kt_total + h->wg_kpb - 1) / h->wg_kpb;
    w.blk_begin = blk;
    blk += w.nchunks * w.ntiles;
    WrSite& r = rp.s[i];
    r.slab = w.slab; r.bpart = w.bpart; r.dw_part = learned_dw ? reg("dwp_", s) : nullptr;
    const std::string n = kSites[s].name;
    auto dst = [&](const std::string& key) { return grads + h->grad_slots[key].first; };
    r.gen_w = dst("motion_conv_gen_" + n + ".weight"); r.gen_b = dst("motion_conv_gen_" + n + ".bias");
    r.down_w = dst("motion_spatial_down_" + n + ".weight"); r.down_b = dst("motion_spatial_down_" + n + ".bias");
    if (learned_dw) { r.dw_w = dst("motion_spatial_grad_" + n + ".weight"); r.dw_b = dst("motion_spatial_grad_" + n + ".bias"); }
    r.C = w.C; r.cpad = w.ntiles * 128; r.nchunks = w.nchunks; r.nsblocks = nsblocks[s];
  }
  wp.total_blocks = blk;
  HIP_TRY(h, pw_wgrad_launch(wp, st));
  HIP_TRY(h, wgrad_reduce_launch(rp, st));
  return OFFK_OK;
}

int offk_segment_consensus_backward(void* stream, const float* grad_out, int B, int T, int C, float* grad_in) {
  if (!grad_out || !grad_in || B < 1 || T < 1 || C < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_segment_consensus_backward: bad argument");
  hipError_t e = consensus_bwd_launch(grad_out, B, T, C, grad_in, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(nullptr, OFFK_ERR_HIP, hipGetErrorString(e));
  return OFFK_OK;
}

int offk_conv2d(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int Ci, const float* w,
                const float* bias, int Co, int KH, int KW, int stride, int pad, const float* res, int res_cstride,
                int res_coff, int flags, float* y, int y_cstride, int y_coff) {
  if (!x || !w || !y || n_img < 1 || H < 1 || W < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_conv2d: bad argument");
  ConvDesc d;
  d.x = x; d.x_cs = x_cstride; d.x_coff = x_coff; d.n_img = n_img; d.H = H; d.W = W; d.Ci = Ci;
  d.w = w; d.bias = bias; d.Co = Co; d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad;
  d.res = res; d.res_cs = res_cstride; d.res_coff = res_coff; d.flags = flags; d.y = y; d.y_cs = y_cstride; d.y_coff = y_coff;
  const char* why = nullptr;
  hipError_t e = conv2d_launch(d, static_cast<hipStream_t>(stream), &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  return OFFK_OK;
}

int offk_conv2d_ex(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int Ci, const float* w,
                   const float* bias, int Co, int KH, int KW, int stride, int pad, const float* res, int res_cstride,
                   int res_coff, int flags, float* y, int y_cstride, int y_coff, int tile_cfg, int splitk, float* partial,
                   size_t partial_floats, int precision) {
  if (!x || !w || !y || n_img < 1 || H < 1 || W < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_conv2d_ex: bad argument");
  ConvDesc d;
  d.x = x; d.x_cs = x_cstride; d.x_coff = x_coff; d.n_img = n_img; d.H = H; d.W = W; d.Ci = Ci;
  d.w = w; d.bias = bias; d.Co = Co; d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad;
  d.res = res; d.res_cs = res_cstride; d.res_coff = res_coff; d.flags = flags; d.y = y; d.y_cs = y_cstride; d.y_coff = y_coff;
  d.tile_cfg = tile_cfg; d.splitk = splitk; d.partial = partial; d.partial_floats = partial_floats;
  d.precision = precision;
  const char* why = nullptr;
  hipError_t e = conv2d_launch(d, static_cast<hipStream_t>(stream), &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  return OFFK_OK;
}

int offk_bottleneck_chain14(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Cin, int relu_in,
                            const float* w1, const float* b1, const float* w2_packed, const float* b2, const float* w3,
                            const float* b3, int K3, const float* res, int res_cstride, int res_coff, float* y, int y_cstride,
                            int y_coff) {
  if (!x || !w1 || !b1 || !w2_packed || !b2 || !w3 || !b3 || !y || n_img < 1)
    return fail(nullptr, OFFK_ERR_INVALID, "offk_bottleneck_chain14: bad argument");
  ChainArgs a;
  a.u2 = nullptr;     // (the stage entry point runs the direct 3x3: it has no place for transformed weights)
  a.x = x; a.x_cs = x_cstride; a.x_coff = x_coff; a.Cin = Cin; a.relu_in = relu_in ? 1 : 0;
  a.w1 = w1; a.b1 = b1; a.w2 = w2_packed; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.K3 = K3;
  a.res = res; a.res_cs = res_cstride; a.res_coff = res_coff; a.y = y; a.y_cs = y_cstride; a.y_coff = y_coff;
  a.n_img = n_img; a.relu_out = 1;
  const unsigned long long xb = ((unsigned long long)n_img * 196 * x_cstride - x_coff) * 4ull;
  a.x_bytes = xb < 0x7fffffffull ? (unsigned)xb : 0u;
  const char* why = nullptr;
  hipError_t e = chain14_launch(a, static_cast<hipStream_t>(stream), &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  return OFFK_OK;
}

int offk_bottleneck_chain14_split(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Cin, int relu_in,
                                  const float* w1, const float* b1, const float* w2_packed, const float* b2, const float* w3,
                                  const float* b3, const float* branch_w, const float* branch_b, const float* res, int res_cstride,
                                  int res_coff, float* y, int y_cstride, int y_coff, void* scratch, size_t scratch_bytes) {
  const char* who = "offk_bottleneck_chain14_split";
  if (!x || !w1 || !b1 || !w2_packed || !b2 || !w3 || !b3 || !y || !scratch || n_img < 1 || (Cin != 64 && Cin != 256) || (branch_w && !branch_b))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument");
  const size_t e1 = (size_t)64 * Cin, e2 = (size_t)64 * 576, e3 = (size_t)256 * 64;
  if (scratch_bytes < 6 * (e1 + e2 + 2 * e3)) return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* sp = static_cast<char*>(scratch);
  ChainArgs a;
  a.u2 = nullptr;
  a.x = x; a.x_cs = x_cstride; a.x_coff = x_coff; a.Cin = Cin; a.relu_in = relu_in ? 1 : 0;
  a.w1 = w1; a.b1 = b1; a.w2 = w2_packed; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.K3 = 64;
  a.res = res; a.res_cs = res_cstride; a.res_coff = res_coff; a.y = y; a.y_cs = y_cstride; a.y_coff = y_coff;
  a.n_img = n_img; a.relu_out = 1;
  const unsigned long long xb = ((unsigned long long)n_img * 196 * x_cstride - x_coff) * 4ull;
  a.x_bytes = xb < 0x7fffffffull ? (unsigned)xb : 0u;
  a.w1p = sp; a.w2p = sp + 6 * e1; a.w3p = sp + 6 * (e1 + e2);
  hipError_t e = wino_pack_split_launch(w1, const_cast<void*>(a.w1p), 64, Cin, 1, st);
  if (e == hipSuccess) e = wino_pack_split_launch(w2_packed, const_cast<void*>(a.w2p), 64, 576, 1, st);
  if (e == hipSuccess) e = wino_pack_split_launch(w3, const_cast<void*>(a.w3p), 256, 64, 1, st);
  if (e == hipSuccess && branch_w) {
    a.wbp = sp + 6 * (e1 + e2 + e3); a.bbr = branch_b;
    e = wino_pack_split_launch(branch_w, const_cast<void*>(a.wbp), 256, 64, 1, st);
  }
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  const char* why = nullptr;
  e = chain14_split_launch(a, st, &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, std::string(who) + ": " + (why ? why : hipGetErrorString(e)));
  return OFFK_OK;
}

namespace {
// shared body of the two Winograd entry points: phases = 1 (3x3 / stride 1 on 7x7) or 4 (polyphase 5x5 / stride 2 on 14x14)
int winograd_entry(const char* who, void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci, int phases,
                   const float* w_packed, const float* bias, int Co, const float* res, int res_cstride, int res_coff, int flags,
                   float* y, int y_cstride, int y_coff, float* scratch, size_t scratch_floats, float* pool_part) {
  if (!x || !w_packed || !y || !scratch || n_img < 1 || Ci < 32 || (Ci & 31) || Co < 64 || (Co & 63) || (flags & OFFK_CONV_RELU_IN_))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument (Ci % 32 == 0, Co % 64 == 0, no RELU_IN)");
  const size_t T = (size_t)n_img, units = phases == 4 ? kWinoUnits4 : kWinoPoints;      // rows per batch entry; row-Ci units of U / V
  const size_t need = units * Ci * ((size_t)Co + T) + (size_t)kWinoPoints * T * Co;
  if (scratch_floats < need) return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* U = scratch;
  float* V = U + units * Co * Ci;
  float* M = V + units * T * Ci;
  hipError_t e = wino_weight_launch(w_packed, Co, Ci, phases, U, st);
  if (e == hipSuccess) e = wino_input_launch(x, x_cstride, x_coff, n_img, Ci, phases, V, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  WinoGroup grp[4];
  const int ngrp = wino_groups(phases, (long long)T, Ci, Co, grp);
  {
    const char* why = nullptr;
    e = wino_gemms_launch(grp, ngrp, kWinoPoints, (int)T, Ci, Co, V, U, M, wino_gemm_stage_default(), st, &why);
    if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  }
  e = wino_output_launch(M, n_img, Co, phases, bias, res, res_cstride, res_coff, flags, y, y_cstride, y_coff, pool_part, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  return OFFK_OK;
}
}  // namespace

int offk_winograd_conv3x3(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci, const float* w_packed,
                          const float* bias, int Co, const float* res, int res_cstride, int res_coff, int flags, float* y,
                          int y_cstride, int y_coff, float* scratch, size_t scratch_floats, float* pool_part) {
  return winograd_entry("offk_winograd_conv3x3", stream, x, x_cstride, x_coff, n_img, Ci, 1, w_packed, bias, Co, res, res_cstride,
                        res_coff, flags, y, y_cstride, y_coff, scratch, scratch_floats, pool_part);
}

int offk_winograd_conv5x5s2(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci, const float* w_packed,
                            const float* bias, int Co, const float* res, int res_cstride, int res_coff, int flags, float* y,
                            int y_cstride, int y_coff, float* scratch, size_t scratch_floats) {
  return winograd_entry("offk_winograd_conv5x5s2", stream, x, x_cstride, x_coff, n_img, Ci, 4, w_packed, bias, Co, res, res_cstride,
                        res_coff, flags, y, y_cstride, y_coff, scratch, scratch_floats, nullptr);
}

int offk_winograd_conv7x7s2(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci, const float* w_packed,
                            const float* bias, int Co, int flags, float* y, int y_cstride, int y_coff, float* scratch,
                            size_t scratch_floats) {
  const char* who = "offk_winograd_conv7x7s2";
  if (!x || !w_packed || !y || !scratch || n_img < 1 || Ci < 32 || (Ci & 31) || Co < 64 || (Co & 63) || (flags & OFFK_CONV_RELU_IN_))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument (Ci % 32 == 0, Co % 64 == 0, no RELU_IN)");
  const size_t T = (size_t)kWino7Tiles * n_img;
  const size_t need = (size_t)kWino7Units * Ci * (Co + T) + (size_t)kWino7Points * T * Co;
  if (scratch_floats < need) return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* U = scratch;
  float* V = U + (size_t)kWino7Units * Co * Ci;
  float* M = V + (size_t)kWino7Units * T * Ci;
  hipError_t e = wino7_weight_launch(w_packed, Co, Ci, U, st);
  if (e == hipSuccess) e = wino7_input_launch(x, x_cstride, x_coff, n_img, Ci, V, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  WinoGroup grp[4];
  const int ngrp = wino7_groups((long long)T, Ci, Co, grp);
  const char* why = nullptr;
  e = wino_gemms_launch(grp, ngrp, kWino7Points, (int)T, Ci, Co, V, U, M, wino_gemm_stage_default(), st, &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  e = wino7_output_launch(M, n_img, Co, bias, flags, y, y_cstride, y_coff, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  return OFFK_OK;
}

int offk_winograd_between(void* stream, const float* M, const float* bias_in, int phases_in, int n_img, int Cin, float* x,
                          int x_cstride, int x_coff, const float* w1, const float* b1, int Cmid, float* V) {
  const char* who = "offk_winograd_between";
  if (!M || !V || n_img < 1 || (w1 && !b1) || (x && (x_cstride < x_coff + Cin || (x_cstride & 3) || (x_coff & 3) || x_coff < 0)))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument");
  if (!wino_mid_supported(Cin, Cmid, w1 != nullptr, phases_in))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": shape not built ((Cin, Cmid) = (128, 128) / (256, 256); phases_in 1, or 4 with Cin 128)");
  WinoMidArgs m;
  m.M = M; m.bias_in = bias_in; m.phases_in = phases_in; m.x = x; m.x_cs = x_cstride; m.x_coff = x_coff;
  m.w1 = w1; m.b1 = b1; m.Cin = Cin; m.Cmid = Cmid; m.n_img = n_img; m.V = V;
  hipError_t e = wino_mid_launch(m, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  return OFFK_OK;
}

int offk_winograd_between_ex(void* stream, const float* M, const float* bias_in, int phases_in, int n_img, int Cin, float* x,
                             int x_cstride, int x_coff, const float* w1, const float* b1, int Cmid, float* V, int precision,
                             void* scratch, size_t scratch_bytes) {
  const char* who = "offk_winograd_between_ex";
  if (precision == OFFK_PRECISION_FP32) return offk_winograd_between(stream, M, bias_in, phases_in, n_img, Cin, x, x_cstride, x_coff, w1, b1, Cmid, V);
  if (precision != OFFK_PRECISION_F32SPLIT || !w1 || !b1 || !scratch || scratch_bytes < (size_t)Cmid * Cin * 6)
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": split-fp32 needs the 1x1 conv and Cmid * Cin * 6 bytes of scratch");
  if (!M || !V || n_img < 1 || (x && (x_cstride < x_coff + Cin || (x_cstride & 3) || (x_coff & 3) || x_coff < 0)))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument");
  if (!wino_mid_supported(Cin, Cmid, true, phases_in))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": shape not built ((Cin, Cmid) = (128, 128) / (256, 256); phases_in 1, or 4 with Cin 128)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = wino_pack_split_launch(w1, scratch, Cmid, Cin, 1, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  WinoMidArgs m;
  m.M = M; m.bias_in = bias_in; m.phases_in = phases_in; m.x = x; m.x_cs = x_cstride; m.x_coff = x_coff;
  m.w1 = w1; m.b1 = b1; m.Cin = Cin; m.Cmid = Cmid; m.n_img = n_img; m.V = V; m.w1p = scratch;
  e = wino_mid_launch(m, st);
  if (e != hipSuccess) return fail_hip(nullptr, e, who);
  return OFFK_OK;
}

int offk_batched_gemm_nt(void* stream, const float* x, const float* w, float* y, int batch, int M, int K, int Co, int precision,
                         void* scratch, size_t scratch_bytes) {
  const char* who = "offk_batched_gemm_nt";
  if (!x || !w || !y || batch < 1 || M < 1 || K < 32 || (K & 31) || Co < 64 || (Co & 63))
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": bad argument (K % 32 == 0, Co % 64 == 0)");
  if (precision != OFFK_PRECISION_FP32 && precision != OFFK_PRECISION_F32SPLIT)
    return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": precision must be OFFK_PRECISION_FP32 or OFFK_PRECISION_F32SPLIT");
  hipStream_t st = static_cast<hipStream_t>(stream);
  WinoGroup grp{batch, 1, 0, 0, 0};
  const void* planes = nullptr;
  if (precision == OFFK_PRECISION_F32SPLIT) {
    if (K < 64) return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": the split-fp32 form takes K >= 64");
    if (!scratch || scratch_bytes < (size_t)batch * Co * K * 6)
      return fail(nullptr, OFFK_ERR_INVALID, std::string(who) + ": scratch smaller than batch * Co * K * 6 bytes (the plane image of w)");
    hipError_t e = wino_pack_split_launch(w, scratch, Co, K, batch, st);
    if (e != hipSuccess) return fail_hip(nullptr, e, who);
    planes = scratch;
  }
  const char* why = nullptr;
  hipError_t e = wino_gemms_launch(&grp, 1, batch, M, K, Co, x, w, y, true, st, &why, planes);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, std::string(who) + ": " + (why ? why : hipGetErrorString(e)));
  return OFFK_OK;
}

int offk_set_conv_plan(offk_handle* h, const char* conv_key, int tile_cfg, int splitk) {
  if (!h || !conv_key) return fail(h, OFFK_ERR_INVALID, "offk_set_conv_plan: null argument");
  for (int c = 0; c < kNumConvs; ++c)
    if (!strcmp(kConvs[c].key, conv_key)) {
      h->conv_cfg[c] = tile_cfg < 0 ? -1 : tile_cfg;
      h->conv_splitk[c] = splitk < 1 ? 0 : splitk;
      return OFFK_OK;
    }
  for (int m = 0; m < 3; ++m)
    if (!strcmp(kMerged[m].name, conv_key)) {
      h->merged_cfg[m] = tile_cfg < 0 ? -1 : tile_cfg;
      h->merged_sk[m] = splitk < 1 ? 0 : splitk;
      return OFFK_OK;
    }
  return fail(h, OFFK_ERR_UNKNOWN_KEY, std::string("offk_set_conv_plan: unknown conv ") + conv_key);
}

int offk_pack_conv_weight(void* stream, const float* w_oihw, int Co, int Ci, int KH, int KW, float* w_ohwi) {
  if (!w_oihw || !w_ohwi || Co < 1 || Ci < 32 || (Ci & 31) || KH < 1 || KW < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_pack_conv_weight: bad argument (Ci % 32 == 0)");
  hipError_t e = pack_conv_weight_launch(w_oihw, Co, Ci, KH, KW, w_ohwi, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, "offk_pack_conv_weight");
  return OFFK_OK;
}

int offk_head(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int C, int maxpool,
              const float* fc_w, const float* fc_b, int num_classes, float* out) {
  if (!x || !fc_w || !fc_b || !out) return fail(nullptr, OFFK_ERR_INVALID, "offk_head: null argument");
  const char* why = nullptr;
  hipError_t e = head_launch(x, x_cstride, x_coff, n_img, H, W, C, maxpool, fc_w, fc_b, num_classes, out,
                             static_cast<hipStream_t>(stream), &why);
  if (e != hipSuccess) return fail(nullptr, why ? OFFK_ERR_INVALID : OFFK_ERR_HIP, why ? why : hipGetErrorString(e));
  return OFFK_OK;
}

int offk_segment_consensus(void* stream, const float* x, int B, int T, int C, float* out) {
  if (!x || !out || B < 1 || T < 1 || C < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_segment_consensus: bad argument");
  hipError_t e = consensus_launch(x, B, T, C, out, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, "offk_segment_consensus");
  return OFFK_OK;
}

int offk_score_fusion(void* stream, const float* const* scores, const float* weights, int n_sets, int videos, int crops,
                      int classes, float* fused, int32_t* pred) {
  if (!scores || !weights || !fused || n_sets < 1 || n_sets > 8 || videos < 1 || crops < 1 || classes < 1)
    return fail(nullptr, OFFK_ERR_INVALID, "offk_score_fusion: bad argument (1..8 score sets)");
  for (int i = 0; i < n_sets; ++i)
    if (!scores[i]) return fail(nullptr, OFFK_ERR_INVALID, "offk_score_fusion: null score set");
  hipError_t e = score_fusion_launch(scores, weights, n_sets, videos, crops, classes, fused, pred, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, "offk_score_fusion");
  return OFFK_OK;
}

int offk_nchw_to_nhwc(void* stream, const float* src, int n_img, int C, int HW, float* dst) {
  if (!src || !dst || n_img < 1 || n_img > 65535 || C < 1 || HW < 1) return fail(nullptr, OFFK_ERR_INVALID, "offk_nchw_to_nhwc: bad argument");
  hipError_t e = nchw_to_nhwc_launch(src, n_img, C, HW, dst, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, "offk_nchw_to_nhwc");
  return OFFK_OK;
}

int offk_nhwc_to_nchw(void* stream, const float* src, int cstride, int coff, int n_img, int C, int HW, float* dst) {
  if (!src || !dst || n_img < 1 || n_img > 65535 || C < 1 || HW < 1 || coff < 0 || coff + C > cstride)
    return fail(nullptr, OFFK_ERR_INVALID, "offk_nhwc_to_nchw: bad argument");
  hipError_t e = nhwc_to_nchw_launch(src, cstride, coff, n_img, C, HW, dst, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail_hip(nullptr, e, "offk_nhwc_to_nchw");
  return OFFK_OK;
}

}  // extern "C"
