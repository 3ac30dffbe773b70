/*
 * offk.h -- C ABI of liboffk.so: the MI355X-native OFF sub-network forward.
 *
 * The reference (JoeHEZHAO/Optical-Flow-Guided-Feature-Pytorch) has NO plugin /
 * operator / FFI boundary for this path: the OFF sub-network is inline Python inside
 * BNInception_OFF.RGB_OFF_forward (RGB_OFF.py:360-860, OFF part :596-847) and
 * BNInception_OFF.forward (Flow_OFF.py:370-887, OFF part :606-876).  The cut this
 * library replaces is therefore the set of nine local tensors
 * inception_{3a,3b,3c,4a,4b,4c,4d,5a,5b}_output_out (RGB_OFF.py:395,418,435,458,481,
 * 504,527,567,590) on the way in and fc_action_motion{,_14,_28} (RGB_OFF.py:787,793,
 * 847) on the way out.  Each entry point below cites the reference lines it stands for.
 *
 * Conventions
 *  - plain C types only; every function returns 0 (OFFK_OK) or a negative error code
 *    and never throws; offk_last_error() gives the message.
 *  - the caller owns every data buffer (device pointers, e.g. torch tensor.data_ptr());
 *    the library owns the handle and its packed weight copies only.
 *  - all work is enqueued asynchronously on the hipStream_t passed as `void* stream`
 *    (NULL = default stream); there is no implicit synchronisation and no other stream: every launch of a call goes to
 *    `stream` in order, so the call can be stream-captured (ABI v8: the handle-owned side stream of v5 - v7 is gone).
 *  - offk_forward runs the units as ONE kernel that fuses the 1x1 reduces with the temporal difference (the gen output G
 *    never goes to HBM) plus the spatial half of K2; OFFK_FUSED_UNITS=0 in the environment at offk_create selects the
 *    two-kernel form (K1 then K2), which offk_off_units / offk_off_units_train always use (the backward needs G).  Both
 *    give the same values (to 2e-6 in fp32: the fused kernel sums k in another grouping).
 *  - a handle is not thread-safe; distinct handles are independent.
 *  - fp32 everywhere.  Boundary tensors are NCHW contiguous exactly as the reference
 *    backbone produces them; INTERNAL activations (workspace, stage entry points) are
 *    channels-last: [rows = image*H*W + y*W + x][channels], see DESIGN.md.
 */
#ifndef OFFK_H_
#define OFFK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFFK_ABI_VERSION 10
#define OFFK_NUM_SITES 9 /* 3a 3b 3c 4a 4b 4c 4d 5a 5b */

enum offk_status {
  OFFK_OK = 0,
  OFFK_ERR_INVALID = -1,        /* bad argument / shape */
  OFFK_ERR_HIP = -2,            /* a HIP runtime call failed */
  OFFK_ERR_MISSING_WEIGHT = -3, /* forward called before every weight was set */
  OFFK_ERR_UNKNOWN_KEY = -4,    /* state_dict key not part of the OFF sub-network */
  OFFK_ERR_NO_DEVICE = -5       /* no gfx950 device visible */
};

enum offk_variant {
  OFFK_VARIANT_RGB_LEARNED_DW = 0, /* RGB_OFF.py: learned depthwise 3x3 + bias (:268,611) */
  OFFK_VARIANT_DIAG_SOBEL = 1      /* Flow_OFF.py:51,622 / RGB_OFF_v2.py:58: util.SobelFilter_Diagonal */
};
enum offk_slice_mode {
  OFFK_SLICE_REFERENCE_FLAT = 0, /* spatial branch = X[:B*(L-1)] on the flat frame axis (RGB_OFF.py:609) */
  OFFK_SLICE_PER_CLIP = 1        /* drop each clip's last frame (what the comment at :608 says) */
};
enum offk_consensus {
  OFFK_CONSENSUS_NONE = 0, /* RGB_OFF.py:849-860: per-pair logits [B*(L-1), classes] */
  OFFK_CONSENSUS_AVG = 1   /* Flow_OFF.py:867-876 / basic_ops.py:19-21: mean over L-1 -> [B, classes] */
};
enum offk_feat_layout {
  OFFK_FEAT_NCHW = 0, /* what the reference backbone's torch.cat produces */
  OFFK_FEAT_NHWC = 1  /* channels_last physical layout of the same logical tensor */
};

enum offk_precision {
  OFFK_PRECISION_FP32 = 0,  /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate */
  /* (1 was OFFK_PRECISION_BF16X3 until ABI v8: each fp32 operand as TWO bf16 planes, three products, ~1e-5 of the fp32 path;
   *  retired in ABI v9 -- offk_create rejects it -- in favour of the exact three-plane mode below) */
  OFFK_PRECISION_F32SPLIT = 2 /* fp32 arithmetic on the bf16 matrix pipe: each fp32 operand cut into THREE bf16 planes
                               (8 + 8 + 8 significand bits = the fp32 value exactly), the six plane products above
                               2^-24 of the leading one on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (units kernel:
                               the leading product and the five small ones in two running accumulators, added once; batched
                               GEMMs: summed per 32-k step from zero and added to the accumulator once per step).  Measured
                               error against fp64 no larger than the fp32 pipe's own on every parity distribution
                               (DESIGN.md); kernels that have no split form run exactly as in OFFK_PRECISION_FP32.
                               Non-finite inputs: the cut of +-Inf leaves Inf - Inf = NaN in the lower planes, so every
                               output such an operand touches is NaN where the fp32 pipe gives +-Inf (or NaN): non-finite
                               in both modes, but not the same non-finite value.  Tiny inputs: a plane below the bf16
                               normal range is a bf16 subnormal, which the MFMA honours (tools/probe_split_mfma.hip part
                               2); a plane below 2^-133 (the last bf16 subnormal) is lost -- only operands below 2^-109
                               in magnitude have such planes (tests/test_gpu_split.py::test_split_nonfinite_and_tiny_inputs). */
};

typedef struct offk_config {
  int32_t batch;       /* B clips            (BNInception_OFF.batch,  RGB_OFF.py:35) */
  int32_t length;      /* L segments / clip  (BNInception_OFF.length, RGB_OFF.py:36), >= 2 */
  int32_t variant;     /* enum offk_variant */
  int32_t slice_mode;  /* enum offk_slice_mode */
  int32_t consensus;   /* enum offk_consensus */
  int32_t num_classes; /* 101 (RGB_OFF.py:332-334) */
  int32_t feat_layout; /* enum offk_feat_layout */
  int32_t device;      /* HIP device ordinal the handle lives on */
  int32_t precision;   /* enum offk_precision: arithmetic of the contractions (1x1 reduce, fusion convs) */
} offk_config;

typedef struct offk_handle offk_handle;

int offk_abi_version(void);

/* Message for the last error on `h` (or, with h == NULL, the last error of a failed
 * offk_create / handle-less call on this thread).  Never NULL. */
const char* offk_last_error(const offk_handle* h);

/* Replaces: BNInception_OFF.__init__ motion_* declarations (RGB_OFF.py:265-358). */
int offk_create(const offk_config* cfg, offk_handle** out);
int offk_destroy(offk_handle* h);

/* Replaces: load_state_dict for the motion_*, fc_action_motion* and sobel_edge_diagonal
 * keys (model_utils.py:188-216, Flow_OFF.py:1398-1413).  `key` is the reference
 * state_dict key (an optional "module." prefix is ignored, test_flow_off.py:52-58);
 * `data` may be a host or a device pointer; shape is checked against the reference's.
 * The library keeps its own packed device copy.  Blocking, load-time call: it first waits for all work on the device
 * (so no kernel still reads the copy being replaced and a device-side `data` is complete whatever stream produced it). */
int offk_set_weight(offk_handle* h, const char* key, const float* data, const int64_t* shape, int ndim);
/* Training-side alternative for the parameters that change every optimizer step (train_off.py:39-45 leaves exactly the
 * OFF units' tensors trainable: motion_conv_gen_<s>, motion_spatial_down_<s>, motion_spatial_grad_<s>, weight and bias):
 * bind the caller's own parameter storage instead of copying it.  `device_data` is the tensor in the reference layout of
 * the key ([128,C,1,1], [128], [32,C,1,1], [32], [32,1,3,3], [32]; contiguous, 16-byte aligned, on the handle's device);
 * every later launch reads it in place, so an in-place optimizer update needs NO call at all -- it only has to be ordered
 * before the next launch, which it is when both are enqueued on the same stream.  The caller keeps the storage alive and
 * re-binds if the tensor is re-allocated.  A later offk_set_weight of the same key replaces the binding by a copy.
 * Binding ANY gen / down weight moves all nine sites off the operand-order weight images of the fused units kernels (the library's
 * own packed copies: the fp32 form's and the split-fp32 form's) onto the register-staged fp32 kernel that reads the bound tensors.
 * `shape` / `ndim` are checked against the key's reference shape, and the allocation behind `device_data` must hold that
 * many bytes (hipMemGetAddressRange): every later launch sizes its buffer descriptors from the key, not from the tensor.
 * Not blocking, no allocation, no kernel launch. */
int offk_bind_weight(offk_handle* h, const char* key, const float* device_data, const int64_t* shape, int ndim);
/* Number of weights still unset (0 = ready); if buf != NULL the first missing key is copied there. */
int offk_missing_weights(const offk_handle* h, char* buf, size_t buflen);

/* Bytes of caller-provided device scratch offk_forward needs for this (B, L). */
size_t offk_workspace_bytes(const offk_handle* h);

/* Replaces: RGB_OFF.py:596-847 / Flow_OFF.py:606-876 (nine OFF units, fusion @28/@14/@7,
 * three heads, optional SegmentConsensus).
 *   feats[i]: site i feature map, [B*L, C_i, H_i, H_i] fp32 (layout per cfg.feat_layout)
 *   out7/out14/out28: [rows, num_classes] fp32, rows = B*(L-1) (consensus none) or B (avg);
 *   out28 may be NULL (the reference computes it but never returns it, RGB_OFF.py:860). */
int offk_forward(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES],
                 float* out7, float* out14, float* out28, void* workspace);

/* Same forward with every feature map handed over as its 1..4 channel groups in concat order --
 * the inception block's branch outputs BEFORE the torch.cat that builds inception_*_output_out
 * (RGB_OFF.py:395,418,435,458,481,504,527,567,590) -- so the backbone can skip that copy (31.6 MB
 * written + re-read per clip).  Group p of site i is [B*L, channels[p], H_i, H_i] (layout per
 * cfg.feat_layout), channels[p] % 32 == 0 (true for every BN-Inception branch), sum = C_i. */
typedef struct offk_feat_parts {
  int32_t n_parts;      /* 1..4 */
  int32_t channels[4];
  const float* data[4];
} offk_feat_parts;
int offk_forward_parts(offk_handle* h, void* stream, const offk_feat_parts parts[OFFK_NUM_SITES],
                       float* out7, float* out14, float* out28, void* workspace);

/* Named regions of the workspace after offk_forward (for stage-level parity tests):
 * "G_<site>", "D_<site>", "fusion_28", "fusion_14", "fusion_7", "sum_7".  All channels-last. */
int offk_workspace_region(const offk_handle* h, const char* name, size_t* offset_bytes, size_t* nbytes);

/* Per-stage device timing.  When enabled, offk_forward brackets each stage with HIP
 * events on the caller's stream; offk_stage_times synchronises those events and
 * returns accumulated milliseconds + launch counts since the last reset.
 * Stage order: 0 pw_reduce (K1), 1 sobel_tdiff (K2), 2 fusion_28, 3 fusion_14,
 * 4 fusion_7, 5 heads+consensus. */
#define OFFK_NUM_STAGES 6
int offk_set_profiling(offk_handle* h, int enable);   /* 0 off, 1 per-stage events, 2 per-launch trace (below) */
int offk_stage_times(offk_handle* h, double ms[OFFK_NUM_STAGES], int64_t calls[OFFK_NUM_STAGES], int reset);
/* Per-launch trace (offk_set_profiling(h, 2)): offk_forward records one HIP event on the caller's stream in front of every
 * launch group -- the units kernels, each fusion conv by its state_dict name (a split-K conv includes its reduction), each
 * head, the consensus.  Returns the number of distinct groups seen since
 * the last reset; the first min(n, max_entries) are written: names as one '\n'-separated string into `names`, accumulated
 * milliseconds and call counts into `ms` / `calls`.  An event between two short kernels costs a few microseconds of
 * command-processor time: use rocprofv3's kernel trace for absolute times of the small launches. */
int offk_launch_times(offk_handle* h, char* names, size_t names_len, double* ms, int64_t* calls, int max_entries, int reset);

/* ---- stage entry points (the same kernels offk_forward launches) ------------------ */

/* K1. Replaces motion_conv_gen_<s> + motion_relu_gen_<s> on all N frames and
 * motion_spatial_down_<s> on the sliced frames (RGB_OFF.py:597-598, 609-610), stacked so
 * the map is read once.  G: [N*HW, 128] (post-ReLU), D: [P*HW, 32] channels-last. */
int offk_pw_reduce(offk_handle* h, void* stream, int site, const float* feat, float* G, float* D);

/* K2. Replaces the temporal subtraction (RGB_OFF.py:599-604), the depthwise 3x3 /
 * diagonal Sobel (RGB_OFF.py:611; Flow_OFF.py:622 + util.py:52-77) and the concats
 * (RGB_OFF.py:616,656,760,832): writes [spatial 32 | temporal 128] into channels
 * [m_coff, m_coff+160) of M, a channels-last buffer with m_cstride channels per pixel.
 * algo: temporal difference by 0 = register rotation over t (default), 1 = t across lanes + wavefront shuffle,
 * 4 = flat shifted stream (T_row[r] = G_row[r + HW] - G_row[r] within a clip; every frame read twice, second time from cache);
 * 2 / 3 / 5 = diagnostics for bandwidth attribution (temporal half only with rotation / spatial half only / temporal
 * half only, flat form). */
int offk_sobel_tdiff(offk_handle* h, void* stream, int site, const float* G, const float* D,
                     float* M, int m_cstride, int m_coff, int algo);

/* K2 alone for all nine sites in ONE grouped launch (exactly what offk_forward enqueues), reading
 * the G_<site> / D_<site> workspace regions a previous offk_off_units / offk_forward filled and
 * writing the fusion buffers.  algo as in offk_sobel_tdiff.  Used by bench / tools to time K2. */
int offk_sobel_tdiff_all(offk_handle* h, void* stream, void* workspace, int algo);

/* K1+K2 for all nine sites into the workspace fusion buffers (two grouped launches). */
int offk_off_units(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace);
/* The units exactly as offk_forward runs them (ABI v9): the fused form when OFFK_FUSED_UNITS is on -- the 1x1 reduces with the
 * temporal difference in one kernel (T and S straight into the fusion_<28|14|7> regions, D_<site> filled, G_<site> NOT) --
 * in the handle's arithmetic (OFFK_PRECISION_F32SPLIT: the split-fp32 kernel).  For stage tests and profiling. */
int offk_off_units_fused(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace);

/* K4. Generic channels-last convolution (the fusion convs, RGB_OFF.py:657-685,762-780,
 * 833-841): y = post( pre(conv(in(x)) + bias) + res ).  x,y,res are channel-sliced views
 * (ptr, channels-per-pixel stride, first channel).  w is in the library's K order
 * [Co][Ci/32][KH*KW][32] (see offk_pack_conv_weight).  Requires Ci % 32 == 0, Co % 64 == 0,
 * strides/offsets % 4 == 0.  Tile shape and K-split are chosen automatically. */
enum offk_conv_flags { OFFK_CONV_RELU_IN = 1, OFFK_CONV_RELU_PRE = 2, OFFK_CONV_RELU_POST = 4,
                       /* (256 was OFFK_CONV_WINO7_FUSED in ABI v8: an experiment that lost, out of the library since ABI v9) */ };
int offk_conv2d(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int Ci,
                const float* w, const float* bias, int Co, int KH, int KW, int stride, int pad,
                const float* res, int res_cstride, int res_coff, int flags,
                float* y, int y_cstride, int y_coff);
/* Same with an explicit plan (tuning / micro-benchmarks): tile_cfg 0..5 = block tile 128x128,
 * 128x64, 256x64, 64x64, 64x128, 128x256 (pixels x channels), 6 / 7 = the LDS-patch kernel (k x k
 * convs whose 196-pixel output groups come from a 28x28 / four 14x14 / one 14x14 / four 7x7 input patch: 7x7s2@28,
 * 5x5s2@14, 3x3s1@14, 3x3s1@7; 128 / 64 output channels per block; splitk then splits the channel chunks)
 * (10 was its half-chunk bf16x3 form: retired in ABI v9), < 0 = automatic; splitk >= 1
 * K-slices whose fp32 partial slabs [splitk][M][Co] go to `partial` (summed in slice order by
 * a second launch, so results are bit-reproducible); splitk < 1 = automatic; precision must be
 * OFFK_PRECISION_FP32 (the stage entry has no split form). */
int offk_conv2d_ex(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int Ci,
                   const float* w, const float* bias, int Co, int KH, int KW, int stride, int pad,
                   const float* res, int res_cstride, int res_coff, int flags,
                   float* y, int y_cstride, int y_coff, int tile_cfg, int splitk, float* partial, size_t partial_floats,
                   int precision);
/* [Co][Ci][KH][KW] (PyTorch) -> [Co][Ci/32][KH*KW][32]; both device pointers. */
int offk_pack_conv_weight(void* stream, const float* w_oihw, int Co, int Ci, int KH, int KW, float* w_packed);
/* Override the plan offk_forward uses for one fusion conv (key = its state_dict name without
 * ".weight", e.g. "motion_conv_trans_28"); tile_cfg < 0 / splitk < 1 restore the automatic choice. */
int offk_set_conv_plan(offk_handle* h, const char* conv_key, int tile_cfg, int splitk);

/* K4c. One bottleneck chain of fusion@28 at 14x14 in one launch (exact fp32): t1 = relu(c1(in(x))), t2 = relu(c2_3x3(t1)),
 * y = relu(c3([t2 | x]) + b3 + res) -- RGB_OFF.py:658-667 (28a: K3 = 128, c3's weight = [motion_conv3_trans_28a |
 * motion_conv_branch_28a] over [t2 | pre-ReLU x], relu_in = 1, no residual) and :670-676 / :679-685 (28b / 28c: Cin = 256, K3 = 64,
 * res = x).  x: [n_img * 196][x_cstride] channels-last, Cin in {64, 256} channels at x_coff; w1 [64][Cin], w2 the packed
 * 3x3 weight [64][2][9][32] (offk_pack_conv_weight), w3 [256][K3]; res / y: 256 channels at res_coff / y_coff.  t1 and t2
 * never leave the chip (csrc/chain_fused.hip). */
int offk_bottleneck_chain14(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Cin, int relu_in,
                            const float* w1, const float* b1, const float* w2_packed, const float* b2,
                            const float* w3, const float* b3, int K3,
                            const float* res, int res_cstride, int res_coff, float* y, int y_cstride, int y_coff);
/* K4cs (ABI v10). The same chain in split-fp32 arithmetic on the bf16 matrix pipe (csrc/chain_split.hip: what an OFFK_PRECISION_F32SPLIT handle
 * runs in offk_forward; the 3x3 direct, every contraction as six bf16 plane products into two fp32 accumulators): c3 over t2 only (K3 = 64).
 * branch_w == NULL: the residual chains 28b / 28c (Cin = 256 or 64; res may be NULL).  branch_w != NULL (Cin = 64, res must be NULL): chain 28a --
 * y = relu(c3(t2) + b3 + branch_w . x + branch_b) with the branch 1x1 [256][64] on the chain input BEFORE relu_in's ReLU, RGB_OFF.py:657-667.
 * Weights as for offk_bottleneck_chain14 (fp32, device); scratch: device memory for their plane images, at least
 * 6 * (64 * Cin + 64 * 576 + 2 * 256 * 64) bytes (cut on every call: a stage entry point, not a fast path). */
int offk_bottleneck_chain14_split(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Cin, int relu_in,
                                  const float* w1, const float* b1, const float* w2_packed, const float* b2,
                                  const float* w3, const float* b3, const float* branch_w, const float* branch_b,
                                  const float* res, int res_cstride, int res_coff, float* y, int y_cstride, int y_coff,
                                  void* scratch, size_t scratch_bytes);
/* K4w. 3x3 / stride 1 / pad 1 convolution on 7x7 maps in Winograd form, fp32 arithmetic (csrc/winograd.hip: a map = four tiles,
 * F(4, 3) x F(3, 3) per axis, 121 points): the five such
 * convs of fusion@14 / @7 (RGB_OFF.py:766-767, 775-780, 833-834, 837-838).  Same epilogue and views as offk_conv2d (flags without
 * OFFK_CONV_RELU_IN); w_packed as for offk_conv2d.  scratch: 121 * (Co * Ci + n_img * (Ci + Co)) floats of device memory
 * (transformed weights, transformed input, GEMM output).  pool_part != NULL: [4 * n_img][Co] sums of the stored values of each
 * output tile (an image = 4 consecutive rows): the average pool of a head that follows. */
int offk_winograd_conv3x3(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci,
                          const float* w_packed, const float* bias, int Co,
                          const float* res, int res_cstride, int res_coff, int flags,
                          float* y, int y_cstride, int y_coff, float* scratch, size_t scratch_floats, float* pool_part);

/* The 5x5 / stride 2 / pad 2 conv on 14x14 maps (motion_conv_trans_14, RGB_OFF.py:762-763) through the same machinery in polyphase
 * form: four 7x7 phase images x 3x3 phase kernels concatenated along K.  x: [n_img * 196][x_cstride]; y: [n_img * 49][y_cstride];
 * w_packed: the packed 5x5 weight [Co][Ci/32][25][32]; scratch: 400 * Ci * (Co + n_img) + 121 * n_img * Co floats. */
int offk_winograd_conv5x5s2(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci,
                            const float* w_packed, const float* bias, int Co,
                            const float* res, int res_cstride, int res_coff, int flags,
                            float* y, int y_cstride, int y_coff, float* scratch, size_t scratch_floats);

/* The 7x7 / stride 2 / pad 3 conv on 28x28 maps (motion_conv_trans_28, RGB_OFF.py:657) in polyphase Winograd form F(5x5, 4x4):
 * four 14x14 phase images x 4x4 phase kernels concatenated along K, 9 output tiles of 5x5 per image, 64 points (winograd7.hip).
 * x: [n_img * 784][x_cstride]; y: [n_img * 196][y_cstride]; w_packed: the packed 7x7 weight [Co][Ci/32][49][32]; flags: ReLU
 * (PRE / POST; no residual input);
 * scratch: 225 * Ci * (Co + 9 * n_img) + 64 * 9 * n_img * Co floats. */
int offk_winograd_conv7x7s2(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int Ci,
                            const float* w_packed, const float* bias, int Co, int flags,
                            float* y, int y_cstride, int y_coff, float* scratch, size_t scratch_floats);

/* What sits BETWEEN two convolutions on the Winograd path at 7x7 maps, in one launch (wino_mid.hip; ABI v8).  Replaces, for the runs
 * 3x3 -> 1x1 -> 3x3 of RGB_OFF.py:762-767 (x1 -> motion_conv1_trans_14a -> motion_conv2_trans_14a) and :833-838 (x2 ->
 * motion_conv1_trans -> motion_conv2_trans), the output transform of the first conv, the 1x1 conv and the input transform of the
 * last one; with w1 == NULL, for :775-780 (motion_conv2_trans_14b -> motion_conv3_trans_14b), the two transforms alone.
 *   M      GEMM output of the conv in front in the Winograd domain, [121][n_img][Cin] (point order: phases_in = 1 a 3x3 / stride 1
 *          conv, 4 the polyphase 5x5 / stride 2 conv); bias_in [Cin] its bias (nullable); ReLU is applied behind it
 *   x      nullable: relu(A^T M A + bias_in) is also stored there, channels-last rows [n_img * 49][x_cstride] at x_coff
 *   w1     1x1 conv [Cmid][Cin] with bias b1 [Cmid] and ReLU behind it; NULL: none (Cmid == Cin)
 *   V      GEMM input of the conv behind, [121][n_img][Cmid]
 * Shapes built: (Cin, Cmid) = (128, 128) and (256, 256); without w1 also Cin = 128 / 256. */
int offk_winograd_between(void* stream, const float* M, const float* bias_in, int phases_in, int n_img, int Cin,
                          float* x, int x_cstride, int x_coff, const float* w1, const float* b1, int Cmid, float* V);
/* (ABI v10) The same with the 1x1 conv in the handle's arithmetic: precision OFFK_PRECISION_F32SPLIT runs stage B in split-fp32 on the bf16
 * matrix pipe (what a split-fp32 handle does in offk_forward; needs w1) -- scratch: Cmid * Cin * 6 bytes of device memory for w1's plane image;
 * OFFK_PRECISION_FP32 = offk_winograd_between. */
int offk_winograd_between_ex(void* stream, const float* M, const float* bias_in, int phases_in, int n_img, int Cin, float* x,
                             int x_cstride, int x_coff, const float* w1, const float* b1, int Cmid, float* V, int precision,
                             void* scratch, size_t scratch_bytes);

/* The batched GEMMs of a convolution on a Winograd path as a stage of their own (wino_gemm.hip / wino_gemm_split.hip; ABI v9):
 *   y[b] = x[b] . w[b]^T,  x [batch][M][K], w [batch][Co][K], y [batch][M][Co], all contiguous fp32; K % 32 == 0, Co % 64 == 0.
 * precision OFFK_PRECISION_FP32: the fp32 matrix pipe.  OFFK_PRECISION_F32SPLIT (K >= 64): split-fp32 arithmetic on the
 * bf16 pipe -- both operands cut into three bf16 planes, six plane products per multiply (the arithmetic of the split units kernel);
 * scratch (>= batch * Co * K * 6 bytes, device) receives the plane image of w.  What a handle created with that precision runs for the
 * fusion convolutions on a Winograd path (RGB_OFF.py:657, 762, 766, 775-777, 833, 837). */
int offk_batched_gemm_nt(void* stream, const float* x, const float* w, float* y, int batch, int M, int K, int Co, int precision,
                         void* scratch, size_t scratch_bytes);

/* K5. Replaces motion_pool_trans_28 / global_pool / squeeze / fc_action_motion*
 * (RGB_OFF.py:782-787, 789-793, 843-847): optional MaxPool(3,2,ceil) then global
 * average over the (pooled) map then Linear.  x: channel slice [x_coff, x_coff+C) of a
 * channels-last buffer with x_cstride channels per pixel. */
int offk_head(void* stream, const float* x, int x_cstride, int x_coff, int n_img, int H, int W, int C,
              int maxpool, const float* fc_w, const float* fc_b, int num_classes, float* out);

/* K6. Replaces ConsensusModule('avg') (basic_ops.py:12-46) as applied in
 * Flow_OFF.py:867-876: x [B, T, C] -> mean over T -> [B, C]. */
int offk_segment_consensus(void* stream, const float* x, int B, int T, int C, float* out);

/* K7. Late score fusion as the eval scripts do it after the forward (test_rgb_off.py:138:
 * np.mean over the 10 crops, weighted sum of the score sets; score_fusion.ipynb cell 8: six sets,
 * weights 1.0/1.5/1.6 (RGB 7x7 / TSN / 14x14) and 1.2/0.8/1.7 (Flow), then argmax):
 *   fused[v][c] = sum_i weights[i] * mean_k scores[i][v][k][c],   pred[v] = argmax_c fused[v][c]
 * scores: n_sets device pointers to [videos, crops, classes] fp32 (host array of pointers);
 * weights: n_sets host floats; pred may be NULL.  With crops = 1 this is the plain weighted sum
 * used for the modality_fuse return (Flow_OFF.py:881). */
int offk_score_fusion(void* stream, const float* const* scores, const float* weights, int n_sets, int videos,
                      int crops, int classes, float* fused, int32_t* pred);

/* ---- Training side of the OFF units ------------------------------
 * What train_off.py:126-151 needs from the path: the units' forward in training mode (nn.Dropout(p=0.8),
 * RGB_OFF.py:356, on every spatial gradient, :612 ...), and the gradients of the units' parameters --
 * the only unit-side tensors train_off.py:39-45 leaves trainable; the feature maps come from the frozen
 * backbone and get no gradient.  The fusion stages / heads between the units and the loss are ordinary
 * convolutions and stay with the caller's autograd; the cut is the gradient w.r.t. each unit output
 * motion_<site> = cat(S 32, T 128) (:616), i.e. the leading channels of the gradients of the three cat
 * results (:656, :760, :832).
 *
 * Dropout is reproducible instead of drawn from the framework's generator: element (pair, c, y, x) of site
 * s is kept iff a 16-bit field of splitmix64(stream(drop_seed, s), (pair*H*W + y*W + x)*8 + c/4) is
 * >= round(drop_p * 65536); kept values are scaled by 1/(1-drop_p).  offk_amd/synth.py (dropout_keep)
 * is the same function in numpy.  drop_p = 0 is eval mode. */
typedef struct offk_grad_view {
  const float* data;   /* channels-last rows [P*H*W][cstride] (e.g. a torch channels_last gradient tensor) */
  int32_t cstride;     /* channels per pixel of that buffer */
  int32_t coff;        /* first of the unit's 160 channels: [coff, coff+32) = S, [coff+32, coff+160) = T */
} offk_grad_view;

/* Workspace for the calls below: the offk_workspace_bytes layout followed by the backward regions
 * ("dG_<site>", "dD_<site>", partial-sum slabs); a superset, usable for offk_forward as well. */
size_t offk_train_workspace_bytes(const offk_handle* h);

/* offk_off_units in training mode: like offk_off_units, with the dropout above applied to S.  Leaves
 * G_<site> / D_<site> in the workspace for the backward. */
int offk_off_units_train(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES], void* workspace,
                         uint64_t drop_seed, double drop_p);

/* Gradient buffer: one flat fp32 device array the caller owns; offk_unit_grad_slot gives the offset and
 * element count of a parameter (reference state_dict key, e.g. "motion_conv_gen_3a.weight") in the
 * reference's own layout ([128,C,1,1], [128], [32,C,1,1], [32], [32,1,3,3], [32]). */
size_t offk_unit_grad_floats(const offk_handle* h);
int offk_unit_grad_slot(const offk_handle* h, const char* key, size_t* offset_floats, size_t* count);

/* Backward of the nine units.  Needs the G_<site> / D_<site> regions of `workspace` as the matching
 * offk_off_units(_train) / offk_forward call on the same feats left them, and the same drop_seed / drop_p.
 * gm[s]: gradient w.r.t. motion_<site>.  grads: the flat buffer above; accumulate != 0 adds to it (the
 * reference calls backward three times per step, train_off.py:141-143), otherwise it is overwritten.
 * Every reduction has a fixed order: results are bit-reproducible.  NCHW feature maps only. */
int offk_off_units_backward(offk_handle* h, void* stream, const float* const feats[OFFK_NUM_SITES],
                            const offk_grad_view gm[OFFK_NUM_SITES], void* workspace, uint64_t drop_seed, double drop_p,
                            float* grads, int accumulate);

/* Backward of offk_segment_consensus, basic_ops.py:29-33: grad_in[b*T + t][c] = grad_out[b][c] / T. */
int offk_segment_consensus_backward(void* stream, const float* grad_out, int B, int T, int C, float* grad_in);

/* NCHW <-> channels-last helpers (device pointers), used by tests and by callers that
 * want a reference-layout view of an internal buffer. */
int offk_nchw_to_nhwc(void* stream, const float* src, int n_img, int C, int HW, float* dst);
int offk_nhwc_to_nchw(void* stream, const float* src, int cstride, int coff, int n_img, int C, int HW, float* dst);

#ifdef __cplusplus
}
#endif
#endif /* OFFK_H_ */
