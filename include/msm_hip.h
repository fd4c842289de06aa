/*
 * msm_hip.h -- C ABI of libmsm_hip.so: the MI355X (gfx950) kernels behind the MSMFormer
 * inference hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - the caller allocates every output and the workspace; the library never allocates,
 *     never synchronises, and launches on the `stream` it is given (a hipStream_t passed as
 *     void* so that the header needs no HIP include);
 *   - every function returns MSM_OK (0) or a negative MSM_E_* code; msm_last_error_string()
 *     describes the last failure on the calling thread;
 *   - floating-point data is fp32 unless an entry point says otherwise: the native-op replacements also come in double
 *     (`_f64`: the reference op dispatches float and double, ops/src/cuda/ms_deform_attn_cuda.cu:69,139), the low-precision
 *     plan's entry points (`_bf16`, `_lp`, `_hm`) take or produce bf16 / fp16 tensors and weight streams as documented per
 *     function; token tensors are batch-major [B][L][E].
 *
 * Reference interfaces replaced (paths relative to the reference root, "OPS" =
 * MSMFormer/meanshiftformer/modeling/pixel_decoder/ops, "DEC" = .../transformer_decoder/
 * meanshiftformer_transformer_decoder.py, "AU" = .../transformer_decoder/attention_util.py,
 * "MS" = lib/utils/mean_shift.py):
 *   msm_msdeform_attn_fwd        <- MSDA.ms_deform_attn_forward, OPS/src/vision.cpp:19,
 *                                   OPS/src/ms_deform_attn.h:25-44, OPS/src/cuda/ms_deform_attn_cuda.cu:25-85
 *   msm_msdeform_attn_bwd        <- MSDA.ms_deform_attn_backward, OPS/src/vision.cpp:20,
 *                                   OPS/src/ms_deform_attn.h:46-66, OPS/src/cuda/ms_deform_attn_cuda.cu:88-158
 *   msm_msdeform_attn_enc_fwd    <- MSDeformAttn.forward lines OPS/modules/ms_deform_attn.py:101-118 fused
 *   msm_mask_logits_fwd          <- forward_prediction_heads einsum + attention-mask, DEC:668-680
 *   msm_hypersphere_attn_fwd     <- hypersphere_attention, AU:64-82 (+ head split/merge AU:364-375,424)
 *   msm_hypersphere_attn_bwd     <- its gradient under torch autograd (training step, tabletop_train_net_pretrained.py:209-246)
 *   msm_kv_project_f32           <- memory/key path of the cross-attention layers, DEC:575, DEC:251, AU:134-140
 *   msm_tokens_proj_nchw_f32     <- layer_1 GroupNorm + ReLU and the mask_features 1x1 convolution, MSD:349-358
 *   msm_dec_post_cross / msm_dec_post_self / msm_dec_heads
 *                                <- the row-local ops between the attention cores of a decoder layer,
 *                                   DEC:245-260, DEC:171-181, DEC:296-300, DEC:637-638, DEC:661-665
 *   msm_gemm_f32 / msm_layernorm_f32 / msm_groupnorm_* / msm_pos_embed_sine
 *                                <- the torch ops around them (F.linear, Conv2d 1x1/3x3, LayerNorm,
 *                                   GroupNorm, F.interpolate, PositionEmbeddingSine)
 *   msm_ms_*                     <- select_smart_seeds MS:128-189, seed_hill_climbing_ball MS:79-109,
 *                                   the assignment/relabel tail of mean_shift_smart_init MS:206-229
 *   msm_conv1x1_in_f32           <- input_proj / lateral 1x1 convolutions of the pixel decoder + GroupNorm moments, MSD:212-238
 *   msm_conv3x3_c64_f32          <- FPN output convolution layer_1 + the moments of its GroupNorm, MSD:264-279,349-351
 *   msm_label_stats              <- per-label loops of the two-stage harness, lib/fcn/test_dataset.py:62-131,183-198
 *   msm_label_image / msm_crop_resize / msm_paste_labels
 *                                <- combine_masks test_utils.py:93-112, crop_rois test_dataset.py:62-112, paste-back :160-177, batched
 *   msm_instance_postprocess     <- F.interpolate + instance_inference,
 *                                   MSMFormer/meanshiftformer/pretrained_meanshiftformer_model.py:337-343,461-497
 */
#ifndef MSM_HIP_H
#define MSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSM_OK 0
#define MSM_E_INVALID (-1)   /* bad argument / unsupported shape */
#define MSM_E_LAUNCH (-2)    /* hip launch error */
#define MSM_E_WORKSPACE (-3) /* workspace too small */

const char* msm_last_error_string(void);
#define MSM_ABI_VERSION 22   /* 22: msm_groupnorm_apply_f16 + msm_conv3x3_c64_f16h (the f16 plan's FPN level on a half token map), msm_conv1x1_in_multi_wide; 21: msm_dec_heads_mask (the next layer's attention mask as the heads kernel's epilogue), msm_l2_prefetch / msm_dec_set_prefetch, msm_dec_*_bf16x2 (hi + lo weight fragments), MSM_OPT_DEC_TILE32; 20: msm_ucn_embedding_tail; 19: backbone glue (msm_bias_act_nhwc, msm_nhwc_to_nchw_f32); 18: flags argument of msm_attn_mask_pooled (bit 1: IEEE-half operands); 17: msm_f32_to_f16_rows; 16: msm_mask_conv3x3_folded (the UCN mask step with the 3x3 mask_features convolution folded into the query embedding); 15: IEEE-half operand forms of the 16-bit plan (precision "f16": msm_dec_*_f16, msm_encoder_block_hm_fwd ffn_f16, fp16 keys in the low-precision attention); 14: cmat_width argument of the K/V projections (separable position constants), msm_conv3x3_c64_nchw_bf16, msm_encoder_prologue_hm_fwd; 13: flags argument of msm_ms_select_seeds_bf16 (persistent on-chip seeding over the bf16 copy), input projections on the bf16 matrix pipe (msm_conv1x1_in_lp, msm_conv1x1_in_multi_lp); 12: head-major bf16 activations between the encoder kernels of the bf16 plan (msm_encoder_block_hm_fwd, msm_msdeform_attn_enc_lp_fwd, msm_f32_to_f16); 11: mean-shift hill climb and the 3x3 FPN convolution with fp32 results on the bf16 matrix pipe (msm_ms_hill_climb_split, msm_groupnorm_apply_split + msm_conv3x3_c64_split), msm_topk_class_scores_gather, zero_buf arguments of msm_pool_mask_taps; 10: bf16-operand 3x3 convolution (msm_conv3x3_c64_bf16), attention masks at key resolution (msm_pool_mask_taps, msm_attn_mask_pooled); 9: float64 MSDeformAttn entry points (_f64), any channel count; 8: bf16 decoder tails, low-precision attention, bf16 K/V projection, split-fp32 encoder block; 7: msm_set_option replaces the environment switches; fused K/V attention, bf16 and backward entry points; 6: post-process workspace size; 5: embed stride / per-query bias of the mask step; 2: flags argument of the mask step, head-major value / packed-weight entry points; 3: msm_label_stats; 4: padded-frame post-process, GroupNorm moment / stride arguments, input-projection, prologue, 3x3 and batched K/V entry points */
int msm_abi_version(void);

/* Kernel-selection overrides for tools/ and tests/ (NOT read on the product path: every option defaults to
 * MSM_OPT_AUTO and the library then picks by shape).  Process-wide, set between launches; msm_set_option
 * returns MSM_OK or MSM_E_INVALID (unknown key), msm_get_option the current value.  The library reads no environment variable. */
#define MSM_OPT_AUTO (-1)
enum {
    MSM_OPT_MASK_NC = 0,        /* mask step wave tile: 1 = 2x16, 2 = 2x32 */
    MSM_OPT_MASKB_TARGET,       /* workgroups of the bf16 mask step */
    MSM_OPT_GEMM_TILE,          /* 0..4 tile configuration of msm_gemm_f32 */
    MSM_OPT_GEMM_SHALLOW,       /* 1: no deep-K tiles */
    MSM_OPT_ATTN_TARGET,        /* workgroup target of the split-K attention kernel */
    MSM_OPT_ATTN_KERNEL,        /* 3: split-K kernel + combine for every length (fallback of the query-split kernel) */
    MSM_OPT_ATTN_QK_MAX,        /* longest sequence the key-split kernel takes */
    MSM_OPT_ATTN_QKCFG,         /* 0 / 1 / 2: two / one / four (four waves) query blocks per workgroup in the key-split kernel */
    MSM_OPT_CONVIN_NT,          /* 1, 2, 4: pixel tiles per workgroup of the input projection; msm_conv1x1_in_multi_wide: K slices of every level (tuning: three decimal digits = one per level, e.g. 421) */
    MSM_OPT_POST_GENERIC,       /* 1: generic mask upsample instead of the 4x form */
    MSM_OPT_ENC_NO_COOP,        /* 1: fp32 encoder block without cooperative workgroups; 2: msm_encoder_block_hm_fwd as ONE 16-wave workgroup per CU (rounds 4-5; default since round 6: eight waves, two workgroups per CU) */
    MSM_OPT_MSDA_GENERIC,       /* 1: generic head-major MSDeformAttn gather; 2: the D = 8 kernel of round 2 (every lane repeats the tap geometry; fallback of the owner-record kernel); 3: that kernel with 8-query x 8-head workgroups */
    MSM_OPT_MS_CHUNK,           /* 1..8 seed blocks per hill-climb launch */
    MSM_OPT_MS_NO_PERSISTENT,   /* 1: one launch per seeding step */
    MSM_OPT_ATTN_FUSED_KV,      /* reserved (no effect) */
    MSM_OPT_KV_PIPE,            /* msm_kv_project_multi_bf16: 0 = fp32 MFMAs with only the store rounded (default: bf16 MFMAs) */
    MSM_OPT_MASK_KERNEL,        /* fp32 mask step: 5 = never the 4-query block on the 4x4x1 MFMA (fallback kernel) */
    MSM_OPT_MS_SPLIT_KERNEL,    /* msm_ms_hill_climb_split: 1 = X split inside the iteration kernel (fallback of the pre-split planes) */
    MSM_OPT_CONV3_WIDE,         /* msm_conv3x3_c64_f32 / _bf16: 0 = one 16-pixel block per wave, 16 waves per workgroup; 1 = two blocks, 8 waves (default: bf16 only) */
    MSM_OPT_DEC_TILE32,         /* msm_dec_*_f16: 1 = 32-row tiles (two 16-row MFMA tiles share every weight fragment), 0 = 16-row tiles (default: 32 from 4096 rows) */
    MSM_OPT_COUNT
};
int msm_set_option(int key, int value);
int msm_get_option(int key);

/* ---------------------------------------------------------------------------------------------
 * Generic fp32 MFMA GEMM:  C[b](m,n) = act( sum_k (A[b](m,k) + A2[b](m,k)) * W[b](n,k) + bias )
 *   A element (m,k) at A + b*a_sb + m*a_sm + k*a_sk; exactly one of a_sm/a_sk is 1
 *   (a_sk==1: row-major activations; a_sm==1: NCHW feature map read as [K][M]).
 *   a_mode 0: strided as above.  a_mode 2: implicit 3x3 convolution over an NHWC token map:
 *     A is [B][conv_h*conv_w][conv_c], M = conv_h*conv_w, K = 9*conv_c, k = tap*conv_c + c,
 *     zero padding 1 (replaces Conv2d(k=3,p=1), msdeformattn.py:268-277).
 *   A2 (nullable) uses the strides of A with its own batch stride a2_sb (0 = broadcast).
 *   W is [N][K] row-major (torch Linear / 1x1-conv weight), batch stride w_sb (0 = shared).
 *   C element (m,n) at C + b*c_sb + m*c_sm + n*c_sn (any strides).
 *   bias_mode 0 none, 1 bias[n], 2 bias[m], 3 bias[m*N + n] (an [M][N] matrix shared by the batch).
 *   act 0 none, 1 relu.
 *   split_k > 1: K is cut in split_k equal parts, part s writes raw sums (no bias/act) to
 *   C + s*c_ss; the consumer (msm_layernorm_f32) adds the parts.
 * ------------------------------------------------------------------------------------------- */
int msm_gemm_f32(const float* A, const float* A2, const float* W, const float* bias, float* C,
                 int M, int N, int K, int batch,
                 int64_t a_sm, int64_t a_sk, int64_t a_sb, int64_t a2_sb, int64_t w_sb,
                 int64_t c_sm, int64_t c_sn, int64_t c_sb, int64_t c_ss,
                 int a_mode, int conv_h, int conv_w, int conv_c,
                 int bias_mode, int act, int split_k, void* stream);

/* y = LayerNorm(x + sum_s parts[s] + bias; g1,b1);  if l2norm: y /= max(||y||,1e-12);
 * if g2: y2 = LayerNorm(y; g2,b2).  rows x E, E in {64,128,256,512}.  x/parts/bias/y2 nullable.
 * parts: n_parts slabs [rows][E] spaced part_stride floats.  (DEC:255-257,178-179,300-304,637-638,661) */
int msm_layernorm_f32(const float* x, const float* parts, int n_parts, int64_t part_stride,
                      const float* bias, const float* g1, const float* b1, int l2norm,
                      const float* g2, const float* b2, float* y, float* y2,
                      int rows, int E, float eps, void* stream);

/* GroupNorm over token maps x [B][HW][C] (NHWC), `groups` groups of C/groups channels.
 * stats: double [B][C][2] (sum, sum of squares), accumulated into; zeroed here first unless stats_cleared != 0. */
int msm_groupnorm_stats_f32(const float* x, double* stats, int stats_cleared, int B, int HW, int C, void* stream);
/* y = GN(x)*gamma+beta (+ bilinear_upsample(up) when up != NULL, align_corners=False, msdeformattn.py:348) (relu when
 * relu != 0).  x/y are [B][H*W][C]; up: image b is [uh*uw][C] at up + b*up_batch_stride floats (0 = dense), e.g. the
 * finest level inside the encoder's token buffer. */
int msm_groupnorm_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta,
                            const float* up, int uh, int uw, int64_t up_batch_stride, float* y,
                            int B, int H, int W, int C, int groups, float eps, int relu, void* stream);

/* The same result written as THREE bf16 planes, v = h + m + l exactly (planes + t * B*H*W*C elements, t = 0, 1, 2, each
 * [B][H*W][C]): the activation operand of msm_conv3x3_c64_split, split once by its producer. */
int msm_groupnorm_apply_split(const float* x, const double* stats, const float* gamma, const float* beta,
                              const float* up, int uh, int uw, int64_t up_batch_stride, uint16_t* planes,
                              int B, int H, int W, int C, int groups, float eps, int relu, void* stream);

/* The same result written as ONE plane of IEEE halves [B][H*W][C], clamped to the half range: the operand bits msm_conv3x3_c64_f16
 * rounds its input to, produced once (the "f16" plan's FPN level: half the bytes between the two kernels; consumer msm_conv3x3_c64_f16h). */
int msm_groupnorm_apply_f16(const float* x, const double* stats, const float* gamma, const float* beta,
                            const float* up, int uh, int uw, int64_t up_batch_stride, void* y_f16,
                            int B, int H, int W, int C, int groups, float eps, int relu, void* stream);

/* y [B][C][HW] (NCHW planes) = GN(x [B][HW][C]) * gamma + beta (relu when relu != 0), stats from msm_groupnorm_stats_f32 /
 * msm_conv3x3_c64_f32: the 64-channel activation the folded mask step contracts with (msm_mask_logits_fwd).
 * C <= 128, HW % 4 == 0. */
int msm_groupnorm_apply_nchw_f32(const float* x, const double* stats, const float* gamma, const float* beta, float* y,
                                 int B, int HW, int C, int groups, float eps, int relu, void* stream);

/* PositionEmbeddingSine(normalize=True) for one H x W map (position_encoding.py:29-52).
 * out element (c, y, x) at out + c*s_c + (y*W+x)*s_p; add_c (nullable, [2*npf]) is added per channel
 * (level embedding, msdeformattn.py:75). */
int msm_pos_embed_sine(float* out, int H, int W, int npf, int64_t s_c, int64_t s_p,
                       const float* add_c, float temperature, float scale, void* stream);

/* batched 2-D transpose: out[b][c][r] = in[b][r][c] */
int msm_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream);

/* y = x / max(||x||_2 over the C channels, eps) for an NCHW map [B][C][HW] (F.normalize(x, p=2, dim=1)): the UCN meta-arch's
 * normalisation of the backbone embedding, pretrained_meanshiftformer_model.py:298-300. */
int msm_l2_normalize_nchw_f32(const float* x, float* y, int B, int C, int HW, float eps, void* stream);

/* The location / softmax glue of the general MSDeformAttn.forward (OPS/modules/ms_deform_attn.py:101-109):
 *   attn_weight = softmax over L*P of logits [rows][M][L*P];  sampling_loc = reference_points[:, :, None, :, None, :] +
 *   offsets / (W_l, H_l) with offsets [rows][M][L][P][2], reference_points [rows][L][2] (rows = N*Lq), spatial_shapes int64 [L][2]. */
int msm_msda_locations(const float* offsets, const float* logits, const float* reference_points, const int64_t* spatial_shapes,
                       float* sampling_loc, float* attn_weight, int64_t rows, int M, int L, int P, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Query x pixel-embedding mask step with fused attention-mask derivation (DEC:668-680).
 *   mask_embed [B][Q][C], mask_feat [B][C][H*W] (NCHW, as produced by the pixel decoder).
 *   mask_out  (nullable) [B][Q][H*W] = einsum("bqc,bchw->bqhw").
 *   attn_out  (nullable) uint8 [B][Q][th*tw]: 1 where sigmoid(bilinear(mask -> th x tw)) < 0.5,
 *             i.e. the 2x2-tap average is negative; requires H/th == W/tw in {2,4,8}.
 *   row_any   (nullable with attn_out) int32 [B][Q]: set to 1 iff some key of the row is
 *             attendable (the reference resets all-masked rows, DEC:618); zeroed by this call.
 *   flags: MSM_MASK_SPARSE (1): rows of the mask that feed neither mask_out nor a tap are skipped;
 *          MSM_MASK_ROW_ANY_CLEARED (2): the caller already zeroed row_any (msm_dec_heads does), no fill is issued.
 *   embed_ld: floats between consecutive rows of mask_embed (0 = C): the C columns may be the head of a wider buffer.
 *   qbias (nullable): per-query constant added to every logit of the query, query (b, q) at qbias[(b*Q + q) * qbias_ld]
 *             (0 = 1).  With these two the step also serves the FOLDED form of the contraction: the mask features are
 *             a 1x1 convolution of the 64-channel FPN activation a (MSD:349-358), so
 *                 einsum(e, Wm a + bm) = einsum(e Wm, a) + e.bm
 *             is computed with C = 64 on `a` directly -- mask_embed = e Wm [B][Q][64], qbias = e.bm -- a quarter of
 *             the FLOPs and of the bytes of the literal order (modeling.FoldedMaskFeatures).
 * ------------------------------------------------------------------------------------------- */
#define MSM_MASK_SPARSE 1
#define MSM_MASK_ROW_ANY_CLEARED 2
#define MSM_MASK_F16 4              /* msm_mask_logits_bf16_fwd only: the packed features are IEEE halves (msm_pack_mask_features_f16), the product runs on the fp16 MFMA */
int msm_mask_logits_fwd(const float* mask_embed, const float* mask_feat, float* mask_out,
                        uint8_t* attn_out, int32_t* row_any,
                        int B, int Q, int C, int H, int W, int th, int tw, int flags,
                        int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream);

/* bf16 variant of the mask step (BASELINE configs 3 and 5; SURVEY 8d: HBM-bound at AI 71.6 FLOP/B): bf16 operands,
 * fp32 accumulation, same outputs and flags.  mask_feat_packed is the channel-quad packed bf16 form of the feature
 * map, [B][C/4][H*W][4] (bf16 bit patterns in uint16), written by msm_pack_mask_features_bf16 from fp32 NCHW;
 * mask_embed stays fp32 and is rounded to bf16 (nearest even) inside the kernel.  C % 16 == 0, C <= 256.
 * Precision "f16": msm_pack_mask_features_f16 writes the same layout with IEEE-half elements (clamped to the half range) and
 * the step is called with flags | MSM_MASK_F16 -- mask_embed is then rounded to fp16 and the product runs on
 * v_mfma_f32_16x16x16_f16 (same rate; 2^-12 instead of 2^-9 roundings on the one step whose sign is the output). */
int msm_pack_mask_features_bf16(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream);
int msm_pack_mask_features_f16(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream);
int msm_mask_logits_bf16_fwd(const float* mask_embed, const uint16_t* mask_feat_packed, float* mask_out,
                             uint8_t* attn_out, int32_t* row_any,
                             int B, int Q, int C, int H, int W, int th, int tw, int flags,
                             int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream);

/* The folded (64-channel) mask step in fp32 accuracy on the bf16 matrix pipe (precision mode f32_split): both operands as exact
 * three-term bf16 splits, six v_mfma_f32_16x16x32_bf16 per product with fp32 accumulation (the dropped terms are below 2^-26
 * of a product).  msm_pack_mask_features_split: fp32 NCHW [B][64][HW] -> [B][3 terms][8][HW][8] bf16, once per forward;
 * msm_mask_logits_split_fwd: arguments as msm_mask_logits_bf16_fwd, C must be 64. */
int msm_pack_mask_features_split(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream);
int msm_mask_logits_split_fwd(const float* mask_embed, const uint16_t* mask_feat_split, float* mask_out,
                              uint8_t* attn_out, int32_t* row_any, int B, int Q, int C, int H, int W, int th, int tw,
                              int flags, int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head hypersphere (vMF) attention core (AU:64-82) on already projected q/k/v:
 *   q [B][Lq][E], k,v [B][S][E] with per-batch strides (elements) q_sb, k_sb, v_sb and row
 *   stride ldq/ldk/ldv; head h uses columns [h*32, h*32+32); head_dim is fixed to 32.
 *   masked (nullable) uint8 [B][Lq][S] (1 = may not attend; shared by all heads, DEC:678);
 *   row_any (nullable) int32 [B][Lq]: rows with 0 ignore the mask (DEC:618).
 *   out [B][Lq][E] = per head normalize(softmax(kappa*q^.k^ + mask) v), heads concatenated.
 *   workspace: float, at least msm_hypersphere_attn_workspace(...) elements.
 * ------------------------------------------------------------------------------------------- */
int64_t msm_hypersphere_attn_workspace(int B, int Lq, int S, int heads);
int msm_hypersphere_attn_fwd(const float* q, const float* k, const float* v,
                             const uint8_t* masked, const int32_t* row_any, float* out,
                             int B, int Lq, int S, int heads,
                             int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb,
                             int64_t ldv, int64_t v_sb, float kappa,
                             float* workspace, int64_t workspace_elems, void* stream);
/* Low-precision form (BASELINE configs 3 / 5; the reference's counterpart is torch.autocast): q^, k^, the probabilities and V
 * enter the bf16 MFMAs as bf16 operands, accumulation / exp / row sums / normalisations stay fp32.  kv_format: 0 = k and v fp32
 * (the self-attention operands written by msm_dec_post_cross); 1 = both bf16 (as written by msm_kv_project_multi_bf16);
 * 2 (precision "f16") = k IEEE half and v bf16 (msm_kv_project_multi_bf16 with half_format = 1) with q^ / k^ on
 * v_mfma_f32_16x16x32_f16 -- kappa = 30 multiplies the cosine's rounding: 2 % of a softmax weight with bf16 operands, 0.25 % with
 * fp16; the probabilities (e^-60 .. 1) and V stay bf16 --; 3 = fp32 k / v with the fp16 score operands of 2.
 * ldk / k_sb / ldv / v_sb are in ELEMENTS of the storage type and must keep k rows 16-byte aligned.
 * Everything else as msm_hypersphere_attn_fwd (same workspace size). */
int msm_hypersphere_attn_lp_fwd(const float* q, const void* k, const void* v, int kv_format,
                                const uint8_t* masked, const int32_t* row_any, float* out,
                                int B, int Lq, int S, int heads,
                                int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb,
                                int64_t ldv, int64_t v_sb, float kappa,
                                float* workspace, int64_t workspace_elems, void* stream);

/* Long key sequences in the 16-bit plans: the folded K/V projection INSIDE the attention kernel (csrc/attention.hip,
 * hs_attn_fkv_kernel).  Replaces msm_kv_project_multi_bf16 + msm_hypersphere_attn_lp_fwd for the cross-attention of
 * PretrainedMeanShiftTransformerDecoder (meanshiftformer_transformer_decoder.py:697-1048; 307 200 keys per image) and the finest level
 * of the three-level decoder at 1280x960: [K | V](key) = x(key) W^T + row[y] + col[x] (attention_util.py:134-140 folded,
 * msm_kv_project_f32's separable form) is computed per 16-key block from the 64-channel fp16 feature -- 128 bytes per key instead of
 * 1024 bytes of bf16 K / V written and read back -- and never stored.
 *   x_f16    [B][H*W][64] IEEE half, token-major (the level feature; msm_f32_to_f16 of its token-major form)
 *   w_packed msm_attn_pack_kv_weights(w [2 * heads * 32][64] fp32 = [K rows | V rows]): fp16 MFMA fragments, heads * 8 KiB
 *   rowcol   [H + W][2 * heads * 32] fp32: the separable constants exactly as msm_kv_project_f32 takes them (cmat_width = W)
 *   col_v_t  [heads * 32][W] fp32: the V columns of the col table transposed (col[H + x][heads * 32 + d] -> col_v_t[d][x])
 *   score_format 1: q^ / k^ as bf16 operands, 2: as IEEE halves (precision "f16"); probabilities and V always bf16
 *   mask_bits (or NULL) msm_attn_pack_mask_bits(masked [B][Lq][S] bytes): the mask bit-packed and blocked, msm_attn_mask_bits_bytes(B, Lq, S)
 *            bytes = [B][ceil(Lq / 112)][S / 16][16 lj][8 m] uint16, bit k of word (lj, m) = masked[112 qc + 16 m + lj][16 kb + k] -- one
 *            16-byte load per lane and key block instead of seven 4-byte loads that use 16 bytes of each of 16 cache lines
 * W % 16 == 0.  q / row_any / out / workspace (msm_hypersphere_attn_workspace(B, Lq, H*W, heads)) as msm_hypersphere_attn_fwd. */
int msm_nchw_to_tokens_f16(const float* in, void* out, int B, int C, int HW, void* stream);   /* in [B][64][HW] fp32 -> out [B][HW][64] IEEE half (clamped): x_f16 of an NCHW level in one pass */
int msm_attn_pack_kv_weights(const float* w, void* packed, int heads, void* stream);
/* The UCN path's mask step (16-bit plans) with the 3x3 mask_features convolution folded into the query embedding.
 * Replaces, for a decoder that only needs the contraction: mask_features = Conv2d(64, 256, 3, padding 1)(x) (pixel_decoder/fpn.py:238-246,
 * 283-290) followed by einsum("bqc,bchw->bqhw", e, mask_features) and, for the attention mask, sigmoid(.) < 0.5 at mask resolution
 * (DEC:1012-1035).  Both are linear in x:  mask[b,q,(y,x)] = sum_{dy,dx,c} F[b,q,3 dy + dx,c] x[b,c,y+dy-1,x+dx-1] + F[b,q,576]  with
 * F[b,q,t,c] = sum_o e[b,q,o] W[o,c,t] and F[b,q,576] = e[b,q,:] . bias -- one small GEMM per prediction.  The (B, 256, H, W) tensor is never made.
 *   x_f16   [B][H*W][64] IEEE half, token-major (msm_nchw_to_tokens_f16: the tensor the fused K/V attention reads)
 *   F       fp32, row q of image b at F + b * f_sb + q * ldf: 576 filter values (k = 64 * tap + channel, tap = 3 * ky + kx as in the
 *           Conv2d weight) + the per-query constant at column 576; ldf % 4 == 0, ldf >= 577.  Rounded to IEEE half (clamped) in the kernel.
 *   exactly one of
 *   mask_bits  msm_attn_mask_bits_bytes(B, Q, H*W) bytes: bit = (mask < 0), bit-packed and blocked as msm_hypersphere_attn_fused_kv_fwd reads it,
 *              with row_any int32 [B][Q] = 1 where a row keeps an unmasked key (zeroed here unless row_any_cleared != 0)
 *   logits     fp32 [B][Q][H*W]
 * Q <= 112, W % 16 == 0, one image of x below 4 GiB.  fp16 operands on v_mfma_f32_16x16x32_f16, fp32 accumulation. */
int msm_mask_conv3x3_folded(const void* x_f16, const float* F, int64_t ldf, int64_t f_sb, void* mask_bits, int32_t* row_any, int row_any_cleared,
                            float* logits, int B, int Q, int H, int W, void* stream);
int64_t msm_attn_mask_bits_bytes(int B, int Lq, int S);
int msm_attn_pack_mask_bits(const uint8_t* masked, void* bits, int B, int Lq, int S, void* stream);
int msm_hypersphere_attn_fused_kv_fwd(const float* q, const void* x_f16, const void* w_packed, const float* rowcol, const float* col_v_t,
                                      int score_format, const void* mask_bits, const int32_t* row_any, float* out,
                                      int B, int Lq, int H, int W, int heads, int64_t ldq, int64_t q_sb, float kappa,
                                      float* workspace, int64_t workspace_elems, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention forward, reference ABI (OPS/src/ms_deform_attn.h:25-44):
 *   value [B][S][M][D], spatial_shapes int64 [L][2] = (H,W), level_start_index int64 [L],
 *   sampling_loc [B][Lq][M][L][P][2] (x,y in [0,1]), attn_weight [B][Lq][M][L][P],
 *   out [B][Lq][M*D].  Any D (D <= 64 takes the tuned kernels: 16-byte taps when D % 4 == 0; larger D the
 *   shape-generic kernel of csrc/msda_generic.hip).  The reference dispatches float and double
 *   (OPS/src/cuda/ms_deform_attn_cuda.cu:69,139); the _f64 entry points are the double instantiation its own
 *   test drives (OPS/test.py:33-43 exact forward check, :66-89 gradcheck).
 * ------------------------------------------------------------------------------------------- */
int msm_msdeform_attn_fwd(const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, float* out,
                          int B, int S, int M, int D, int L, int Lq, int P, void* stream);

/* Backward of the above, reference ABI ms_deform_attn_backward (OPS/src/ms_deform_attn.h:46-66,
 * OPS/src/cuda/ms_deform_attn_cuda.cu:88-158, kernels ms_deform_im2col_cuda.cuh:306-925):
 *   grad_output [B][Lq][M*D] -> grad_value [B][S][M][D], grad_sampling_loc [B][Lq][M][L][P][2],
 *   grad_attn_weight [B][Lq][M][L][P].
 *   Exception to "the library only launches kernels": grad_value is accumulated with hardware atomics as in the
 *   reference (cuh:128-160), so these entry points first zero-fill it (and, for the fp32 kernels when D/4 is not a
 *   power of two, the other two outputs) with hipMemsetAsync on `stream` -- the counterpart of the reference's
 *   at::zeros (cu:121-123).  Under stream capture that is a memset node of the graph. */
int msm_msdeform_attn_bwd(const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, const float* grad_output,
                          float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                          int B, int S, int M, int D, int L, int Lq, int P, void* stream);
int msm_msdeform_attn_fwd_f64(const double* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const double* sampling_loc,
                              const double* attn_weight, double* out,
                              int B, int S, int M, int D, int L, int Lq, int P, void* stream);
int msm_msdeform_attn_bwd_f64(const double* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const double* sampling_loc,
                              const double* attn_weight, const double* grad_output,
                              double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                              int B, int S, int M, int D, int L, int Lq, int P, void* stream);

/* Encoder self-attention form with the sampling arithmetic fused in
 * (OPS/modules/ms_deform_attn.py:101-109 + msdeformattn.py:141-153): query i is pixel i of the
 * concatenated levels, its reference point is that pixel's centre; `proj` [B][S][M*L*P*3] holds
 * the raw sampling_offsets (first M*L*P*2 columns, (M,L,P,2) order) and attention logits (last
 * M*L*P columns, (M,L*P) order) of the two linears; softmax over L*P is done here. */
int msm_msdeform_attn_enc_fwd(const float* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* proj, float* out,
                              int B, int S, int M, int D, int L, int P, void* stream);

/* The same with `value_hm` in HEAD-MAJOR order [B][M][S][D]: the two x-neighbours of a bilinear tap are adjacent in
 * memory, so a tap row is one contiguous 2*D*4-byte segment (the gather is bound by distinct cache lines per wave
 * instruction).  D % 4 == 0 and 2*D/4 dividing 256 (or D % 4 != 0 and 2*D dividing 256).
 * msm_value_to_head_major_f32 converts a token-major value [B][S][M][D]; msm_encoder_block_fwd can write the layout itself. */
int msm_value_to_head_major_f32(const float* value, float* value_hm, int B, int S, int M, int D, void* stream);
int msm_msdeform_attn_enc_hm_fwd(const float* value_hm, const int64_t* spatial_shapes,
                                 const int64_t* level_start_index, const float* proj, float* out,
                                 int B, int S, int M, int D, int L, int P, void* stream);

/* The encoder form with the sampling projection computed in the kernel (round 3): instead of reading `proj`, a workgroup
 * evaluates [sampling_offsets | attention_weights](src + pos) (OPS/modules/ms_deform_attn.py:99-101, query = src + pos
 * msdeformattn.py:124) for its 64 queries and one head on the matrix pipe -- the 58 MB `proj` tensor per layer (B = 8)
 * never exists.  value_hm as msm_msdeform_attn_enc_hm_fwd; src [B][S][64] the layer input, pos [S][64] its position /
 * level code; wpack / bpack from msm_msda_pack_proj (per head: 24 offset rows, 12 logit rows, zero-padded to 48, in MFMA
 * fragment order; (M*3*4*64*4) and (M*48) floats).  Only the pixel decoder's geometry: 8 heads x 8 channels, 3 levels x 4
 * points.  Projected values are bitwise those msm_encoder_block_fwd writes. */
int msm_msda_pack_proj(const float* w /* [M*L*P*3][64] = [offsets ; logits] */, const float* bias, float* wpack, float* bpack,
                       int M, int L, int P, void* stream);
int msm_msdeform_attn_enc_fused_fwd(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* src, const float* pos, const float* wpack, const float* bpack, float* out,
                                    int B, int S, int M, int D, int L, int P, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused token-wise block of one MSDeformAttn encoder layer (msdeformattn.py:122-131):
 *   src_out = LN2(x + linear2(relu(linear1(x)))),  x = LN1(src + output_proj(attn))
 * and, when value_out/proj_out are given, the NEXT layer's value_proj(src_out) and
 * [sampling_offsets | attention_weights](src_out + pos) (ops/modules/ms_deform_attn.py:95-104).
 *   attn, src, src_out, value_out: [M][64]; proj_out [M][proj_width]; pos [S][64], token t uses pos[t % S].
 *   wstream: the layer's weights packed by the host into 32 KiB chunks of 4 KiB blocks in consumption
 *   order (unseenobjectswithmeanshift_amd/modeling.py::pack_encoder_block documents the layout);
 *   msm_encoder_block_stream_floats() gives its length.  small: bo,g1,be1 (64 each), b1 (d_ffn), b2,
 *   g2,be2,bv (64 each), bp (proj_width).  d_model is fixed to 64.
 *   value_heads: 0 -> value_out is token-major [M][64]; h > 0 -> head-major [M/S][h][S][64/h], the layout
 *   msm_msdeform_attn_enc_hm_fwd reads.
 * ------------------------------------------------------------------------------------------- */
int64_t msm_encoder_block_stream_floats(int d_ffn, int proj_width);
int msm_encoder_block_fwd(const float* attn, const float* src, const float* wstream, const float* small,
                          const float* pos, float* src_out, float* value_out, float* proj_out,
                          int M, int S, int d_ffn, int proj_width, int value_heads, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Folded key/value projection of one feature level (DEC:575 input_proj + level_embed, DEC:251 "+ pos", AU:134-140
 * k/v in-projection -- everything affine in the level feature folded on the host):
 *   out [B][HW][N] = x^T w^T + cmat,  w [N][64], cmat [HW][N] shared by the batch,
 *   N in {256, 512} (512 = [K | V] of one cross-attention layer).  x: image b starts at x + b*x_batch_stride and is
 *   [C = 64][HW] (x_tokens = 0, NCHW) or [HW][64] (x_tokens = 1: token-major = torch channels_last, e.g. a slice of
 *   the pixel decoder's token buffer, so no transpose pass is needed).
 *   cmat_width = 0: cmat is the dense [HW][N] constant.  cmat_width = W > 0 (W divides HW): the constant is SEPARABLE and cmat
 *   holds [HW / W row vectors | W column vectors] x N -- token p = (y, x) gets row[y] + col[x] (the sine position embedding's
 *   first half depends on y only, its second on x only, position_encoding.py:44-51): a 307 200-key map then reads two tables
 *   of 1120 rows instead of 629 MB of constants.  Result: fl(fl(x^T w^T + row) + col).
 * ------------------------------------------------------------------------------------------- */
int msm_kv_project_f32(const float* x, const float* w, const float* cmat, float* out,
                       int B, int C, int HW, int N, int x_tokens, int64_t x_batch_stride, int cmat_width, void* stream);
/* n_jobs <= 16 such projections (one per cross-attention layer: its level's features, its folded weight and constant)
 * in ONE launch; x / w / cmat / out / HW / x_tokens / x_batch_stride / cmat_width are HOST arrays of n_jobs entries (cmat_width may
 * be NULL = all dense; all jobs separable or none), B, C = 64 and N are shared.  Each job gets a share of the chip's workgroups
 * proportional to its HW. */
int msm_kv_project_multi_f32(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                             float* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                             const int32_t* cmat_width, int B, int C, int N, void* stream);
/* The same with the result stored as bf16 (low-precision mode): half the bytes of this write-bound launch and of the K/V
 * reads of msm_hypersphere_attn_lp_fwd.  w is rounded to one bf16 and x enters as a hi + lo pair (bf16 MFMAs, fp32
 * accumulation; MSM_OPT_KV_PIPE = 0: exact fp32 MFMAs, only the stored value rounded).
 * half_format = 1 (precision "f16"; N = 512 = [K | V]): w and x enter v_mfma_f32_16x16x32_f16 as one IEEE-half term each (one MFMA
 * per product instead of two; 2^-12 roundings instead of w's 2^-9), the K columns [0, N/2) are stored as IEEE half, the V columns
 * as bf16: the layout msm_hypersphere_attn_lp_fwd reads with kv_format = 2. */
int msm_kv_project_multi_bf16(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                              uint16_t* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                              const int32_t* cmat_width, int B, int C, int N, int half_format, void* stream);
/* fp32 results on the bf16 matrix pipe (exact three-term splits of x and w, six K = 32 MFMAs per product; see
 * msm_encoder_block_split_fwd): same arguments and output as msm_kv_project_multi_f32. */
int msm_kv_project_multi_split(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                               float* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                               const int32_t* cmat_width, int B, int C, int N, void* stream);

/* mask_features (MSD:349-358): the last GroupNorm + ReLU of the FPN level fused into the 1x1 convolution after it.
 *   out [B][N][HW] (NCHW) = bias + w act(x),  x [B][HW][64] tokens, w [N][64], N in {256, 512}, HW % 4 == 0;
 *   act(x) = relu?((x - mean_g) * rstd_g * gamma + beta) with (mean, rstd) from gn_stats -- the per-(image, channel)
 *   double (sum, sum of squares) written by msm_groupnorm_stats_f32 -- or the identity when gn_stats is null. */
int msm_tokens_proj_nchw_f32(const float* x, const float* w, const float* bias, const double* gn_stats,
                             const float* gn_gamma, const float* gn_beta, int groups, float eps, int relu, float* out,
                             int B, int C, int HW, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused row-local tails of one decoder layer on the query matrix [rows = B*Q][E], E fixed to 256.  Row r uses
 * query_pos[r % Q].  Every weight MATRIX argument (wo, w_in, w1, w2, m0w..m2w, wq) is the PACKED form of torch's
 * (out_features N, in_features K) matrix produced by msm_dec_pack_weight -- MFMA B-fragment order, so that a
 * wavefront's 16-byte loads are 1 KiB contiguous:
 *     packed[((t*(K/64) + kc)*4 + u)*256 + (lq*16 + lj)*4 + c] = W[t*16 + lj][kc*64 + u*16 + lq*4 + c]
 * (t < N/16, kc < K/64, u,lq,c < 4, lj < 16).  Row blocks of 16 stay contiguous, so "rows [a, b) of W" is still the
 * pointer offset a*K.  Bias / LayerNorm vectors are plain.
 *
 * msm_dec_post_cross -- after the cross-attention core (forward_post DEC:245-260, then the self-attention
 *   in-projection AU:134-140 with q = k = tgt + query_pos, v = tgt, DEC:171-175):
 *     x_out = LN(res + attn_out wo^T + bo);  qk_out [rows][2E] = (x + query_pos) w_in[0:2E]^T + b_in[0:2E];
 *     v_out = x w_in[2E:3E]^T + b_in[2E:3E]
 * msm_dec_post_self -- after the self-attention core (DEC:171-181, then the FFN body DEC:296-299 split over the
 *   hidden dimension): x_out = LN(res + attn_out wo^T + bo);
 *     parts[p] [rows][E] = relu(x w1[S_p]^T + b1[S_p]) w2[:, S_p]^T,  p < n_parts, S_p = the p-th of n_parts equal
 *     slices of the hidden dimension (n_parts divides F/256; their sum is linear2's output WITHOUT its bias)
 * msm_dec_heads -- FFN residual/norm, block norm and prediction-head inputs (DEC:300, DEC:637-638, DEC:661-665)
 *   plus the next layer's cross-attention query projection:
 *     t = x + sum_c parts[c] + bias;  if ln_g: t = LN(t);  if l2norm: t = t / max(||t||, 1e-12);  out = t
 *     d_out = LN_dec(t);  e_out = m2(relu(m1(relu(m0(d)))));  q_out = (t + query_pos) wq^T + bq
 *   out, d_out and the wq/bq/query_pos/q_out group are optional (null).  row_any_zero (nullable) int32 [rows] is
 *   cleared: the row_any buffer of the mask step that consumes e_out (pass MSM_MASK_ROW_ANY_CLEARED there).
 * ------------------------------------------------------------------------------------------- */
int msm_dec_pack_weight(const float* w, float* packed, int N, int K, void* stream);
int msm_dec_post_cross(const float* attn_out, const float* res, const float* query_pos,
                       const float* wo, const float* bo, const float* ln_g, const float* ln_b,
                       const float* w_in, const float* b_in,
                       float* x_out, float* qk_out, float* v_out,
                       int rows, int Q, int E, float eps, void* stream);
int msm_dec_post_self(const float* attn_out, const float* res,
                      const float* wo, const float* bo, const float* ln_g, const float* ln_b,
                      const float* w1, const float* b1, const float* w2, int F,
                      float* x_out, float* parts, int n_parts, int rows, int E, float eps, void* stream);
int msm_dec_heads(const float* x, const float* parts, int n_parts, const float* bias,
                  const float* ln_g, const float* ln_b, int l2norm,
                  const float* dec_g, const float* dec_b,
                  const float* m0w, const float* m0b, const float* m1w, const float* m1b,
                  const float* m2w, const float* m2b,
                  const float* wq, const float* bq, const float* query_pos,
                  float* out, float* d_out, float* e_out, float* q_out, int32_t* row_any_zero,
                  int rows, int Q, int E, float eps, void* stream);

/* The same three tails with bf16 MFMA operands and fp32 accumulation (low-precision mode, BASELINE configs 3 / 5; the
 * reference's counterpart is torch.autocast over the whole model, MSMFormer/tabletop_train_net_pretrained.py:232).  Every
 * weight MATRIX argument is the bf16 packed form produced by msm_dec_pack_weight_bf16 (same tile order, 2 bytes per
 * weight):
 *     packed[(((t*(K/64) + kc)*2 + up)*64 + lq*16 + lj)*8 + h*4 + c] = bf16(W[t*16 + lj][kc*64 + (2*up + h)*16 + lq*4 + c])
 * so "rows [a, b) of W" is still the element offset a*K.  Activations enter the MFMAs as hi + lo bf16 pairs (exact to
 * 2^-17); biases, LayerNorms, residual streams and every tensor that leaves the kernels stay fp32. */
int msm_dec_pack_weight_bf16(const float* w, uint16_t* packed, int N, int K, void* stream);
int msm_dec_post_cross_bf16(const float* attn_out, const float* res, const float* query_pos,
                            const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                            const uint16_t* w_in, const float* b_in,
                            float* x_out, float* qk_out, float* v_out,
                            int rows, int Q, int E, float eps, void* stream);
int msm_dec_post_self_bf16(const float* attn_out, const float* res,
                           const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                           const uint16_t* w1, const float* b1, const uint16_t* w2, int F,
                           float* x_out, float* parts, int n_parts, int rows, int E, float eps, void* stream);
int msm_dec_heads_bf16(const float* x, const float* parts, int n_parts, const float* bias,
                       const float* ln_g, const float* ln_b, int l2norm,
                       const float* dec_g, const float* dec_b,
                       const uint16_t* m0w, const float* m0b, const uint16_t* m1w, const float* m1b,
                       const uint16_t* m2w, const float* m2b,
                       const uint16_t* wq, const float* bq, const float* query_pos,
                       float* out, float* d_out, float* e_out, float* q_out, int32_t* row_any_zero,
                       int rows, int Q, int E, float eps, void* stream);

/* The same three tails with IEEE-half MFMA operands (precision "f16": the 16-bit plan with the smaller rounding error).  Weight
 * matrices are the packed form of msm_dec_pack_weight_f16 -- the bf16 layout above with fp16(W) elements (Linear weights are
 * O(0.01 .. 1): three more significand bits, no range concern) -- and the activation fragment enters v_mfma_f32_16x16x32_f16 as
 * ONE fp16 term, clamped to +-65504 (its rounding is of the weight's order, so the hi + lo pair of the bf16 form buys nothing):
 * half the MFMAs of the bf16 tails and an eighth of their rounding error (0.2 % against 0.85 % of the final
 * mask bits for the tails alone; 28 us per layer against 32 at 800 rows).  Everything else as the bf16 form. */
int msm_dec_pack_weight_f16(const float* w, uint16_t* packed, int N, int K, void* stream);
int msm_dec_post_cross_f16(const float* attn_out, const float* res, const float* query_pos,
                           const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                           const uint16_t* w_in, const float* b_in,
                           float* x_out, float* qk_out, float* v_out,
                           int rows, int Q, int E, float eps, void* stream);
int msm_dec_post_self_f16(const float* attn_out, const float* res,
                          const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                          const uint16_t* w1, const float* b1, const uint16_t* w2, int F,
                          float* x_out, float* parts, int n_parts, int rows, int E, float eps, void* stream);
int msm_dec_heads_f16(const float* x, const float* parts, int n_parts, const float* bias,
                      const float* ln_g, const float* ln_b, int l2norm,
                      const float* dec_g, const float* dec_b,
                      const uint16_t* m0w, const float* m0b, const uint16_t* m1w, const float* m1b,
                      const uint16_t* m2w, const float* m2b,
                      const uint16_t* wq, const float* bq, const float* query_pos,
                      float* out, float* d_out, float* e_out, float* q_out, int32_t* row_any_zero,
                      int rows, int Q, int E, float eps, void* stream);

/* The bf16 tails with hi + lo WEIGHT fragments (round 6; what set_precision("bf16") runs): msm_dec_pack_weight_bf16x2 packs
 * [bf16(W) | bf16(W - bf16(W))] along K (packed holds N x 2 K elements, the bf16 fragment order per half), and a stage accumulates the hi
 * chunks and then the lo chunks -- with the activation's own hi + lo split every product keeps 2^-17 instead of the weight's 2^-9, which
 * was 0.85 % of the bf16 plan's 1.08 % of flipped final-mask bits.  Twice the weight stream and MFMAs of the _bf16 forms; same arguments. */
int msm_dec_pack_weight_bf16x2(const float* w, uint16_t* packed, int N, int K, void* stream);
int msm_dec_post_cross_bf16x2(const float* attn_out, const float* res, const float* query_pos,
                              const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                              const uint16_t* w_in, const float* b_in,
                              float* x_out, float* qk_out, float* v_out,
                              int rows, int Q, int E, float eps, void* stream);
int msm_dec_post_self_bf16x2(const float* attn_out, const float* res,
                             const uint16_t* wo, const float* bo, const float* ln_g, const float* ln_b,
                             const uint16_t* w1, const float* b1, const uint16_t* w2, int F,
                             float* x_out, float* parts, int n_parts, int rows, int E, float eps, void* stream);
int msm_dec_heads_bf16x2(const float* x, const float* parts, int n_parts, const float* bias,
                         const float* ln_g, const float* ln_b, int l2norm,
                         const float* dec_g, const float* dec_b,
                         const uint16_t* m0w, const float* m0b, const uint16_t* m1w, const float* m1b,
                         const uint16_t* m2w, const float* m2b,
                         const uint16_t* wq, const float* bq, const float* query_pos,
                         float* out, float* d_out, float* e_out, float* q_out, int32_t* row_any_zero,
                         int rows, int Q, int E, float eps, void* stream);

/* msm_dec_heads_bf16 / _f16 with the NEXT layer's attention mask at key resolution as the kernel's epilogue (round 6; replaces the
 * pair msm_dec_heads_* + msm_attn_mask_pooled(flags & 2) of the 16-bit plans: DEC:660-682 with interpolate(einsum(e, F)) =
 * einsum(e, interpolate(F))).  e_out must be the folded embedding [e Wm | e.bm | ...]: columns 0..63 contract with pooled
 * [rows / Q][T][64] (msm_pool_mask_taps), column qcol (>= 64) is the per-query bias.  attn: (B, Q, T) bytes, or with flags & 1 the
 * bit-packed blocked form of msm_attn_pack_mask_bits (T % 16 == 0); row_any (B, Q) must arrive ZEROED and receives 1 where a row
 * keeps an unmasked key.  flags & 2: fp16 weight fragments (msm_dec_pack_weight_f16), else bf16 fragments; flags & 4: the mask
 * contraction on IEEE-half operands (msm_attn_mask_pooled's flag 2), else its fp32 MFMA chain.  Same values as the two-launch form,
 * bit for bit.  rows must be a multiple of Q: tiles
 * are image-aligned (ceil(Q / 16) per image). */
int msm_dec_heads_mask(const float* x, const float* parts, int n_parts, const float* bias,
                       const float* ln_g, const float* ln_b, int l2norm,
                       const float* dec_g, const float* dec_b,
                       const uint16_t* m0w, const float* m0b, const uint16_t* m1w, const float* m1b,
                       const uint16_t* m2w, const float* m2b,
                       const uint16_t* wq, const float* bq, const float* query_pos,
                       float* out, float* d_out, float* e_out, float* q_out,
                       const float* pooled, int T, int qcol, uint8_t* attn, int32_t* row_any, int flags,
                       int rows, int Q, int E, float eps, void* stream);

/* msm_l2_prefetch -- touch up to 8 device byte ranges (16-byte aligned) so that every XCD's L2 holds them: launched on a side stream
 * beside a kernel that leaves the fabric idle, ahead of the kernel that streams the ranges (the decoder tails' packed weights; round 6,
 * csrc/prefetch.hip).  ptrs / bytes are HOST arrays of n entries.  Affects speed only. */
int msm_l2_prefetch(const void* const* ptrs, const int64_t* bytes, int n, void* stream);
/* msm_dec_set_prefetch -- the same job without a launch of its own: the NEXT msm_dec_post_cross* / msm_dec_post_self* / msm_dec_heads*
 * call of this thread carries one extra row of workgroups that touch the n <= 6 ranges (the packed weights of the launches behind it in
 * the decoder's chain) and exit; n = 0 clears a pending request.  (A forked side stream costs ~10 us per fork / join inside a HIP graph.) */
int msm_dec_set_prefetch(const void* const* ptrs, const int64_t* bytes, int n);

/* ---------------------------------------------------------------------------------------------
 * Backward of msm_hypersphere_attn_fwd (training step of the reference: hypersphere_attention under autograd, AU:64-82,
 * MSMFormer/tabletop_train_net_pretrained.py:209-246).  q/k/v, masked, row_any, strides and kappa as in the forward call;
 * grad_out [B][Lq][heads*32] -> grad_q [B][Lq][heads*32], grad_k / grad_v [B][S][heads*32] (all contiguous).  The
 * probabilities are recomputed (nothing of size Lq x S is stored); workspace floats >= msm_hypersphere_attn_bwd_workspace. */
int64_t msm_hypersphere_attn_bwd_workspace(int B, int Lq, int heads);
int msm_hypersphere_attn_bwd(const float* q, const float* k, const float* v, const uint8_t* masked, const int32_t* row_any,
                             const float* grad_out, float* grad_q, float* grad_k, float* grad_v, int B, int Lq, int S, int heads,
                             int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv, int64_t v_sb, float kappa,
                             float* workspace, int64_t workspace_elems, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Classic vMF mean shift over unit embeddings X [n][d] (d == 64), cosine metric.
 * ------------------------------------------------------------------------------------------- */
/* Farthest-point seeding (MS:155-187): indices[0] = first_index, then S-1 x { nearest =
 * min(nearest, 0.5*(1 - X.s)); next = first argmax }.  seeds_out [S][d], indices_out int64 [S].
 * workspace floats >= msm_ms_seed_workspace(n).
 * Maps of up to 393 216 rows take ONE persistent launch whose workgroups meet at a grid barrier after every step; that
 * needs all of them co-resident.  When other work holds CUs for too long the kernel gives up at a bounded wait and
 * writes -1 to every index (the seeds are then undefined): the caller re-issues the call with flags bit 0 set, which
 * takes the one-launch-per-step path (identical results).  flags bit 0: stepwise path. */
#define MSM_MS_SEED_STEPWISE 1
#define MSM_MS_SEED_TEST_GIVE_UP 2   /* tests: the persistent kernel starts with its give-up flag raised */
int64_t msm_ms_seed_workspace(int n);
int msm_ms_select_seeds(const float* X, int n, int d, int num_seeds, int64_t first_index,
                        float* seeds_out, int64_t* indices_out,
                        float* workspace, int64_t workspace_elems, int flags, void* stream);
/* iters x { Z = normalize( exp(kappa * Z X^T) X ) } (MS:90-107); Z [S][d] updated in place. */
int64_t msm_ms_hill_climb_workspace(int n, int S);
int msm_ms_hill_climb(const float* X, int n, int d, float* Z, int S, float kappa, int iters,
                      float* workspace, int64_t workspace_elems, void* stream);
/* The same iteration with every fp32 product carried out as six bf16 MFMAs on exact three-term splits of both operands (X,
 * Z and the exp() weights): fp32-accurate results -- the error against float64 stays within 1.5x of the fp32 MFMA kernel's --
 * at 0.375 of its matrix time.  X is split once per call into three bf16 planes kept in the workspace (6 bytes per element),
 * which is therefore larger: msm_ms_hill_climb_split_workspace floats, 16-byte aligned like X and Z.  Opt-in (the host passes
 * precision="f32_split"). */
int64_t msm_ms_hill_climb_split_workspace(int n, int S);
int msm_ms_hill_climb_split(const float* X, int n, int d, float* Z, int S, float kappa, int iters,
                            float* workspace, int64_t workspace_elems, void* stream);
/* ---- precision "bf16" of the classic clustering (BASELINE configs[4]: n = 1 228 800, 300 seeds, 20 iterations, "HBM-bound stress") ----
 * One bf16 copy of X (rows padded with zeros to msm_ms_bf16_rows(n), a multiple of 32) serves seeding and the hill climb: half the
 * bytes of the S seeding passes (lib/utils/mean_shift.py:128-189: at this size nothing but an HBM stream), one plane instead of
 * three in the hill climb (MS:79-109) with single bf16 products except the seeds (h + l terms).  Distances are those of the rounded
 * points: seeds and labels equal the fp32 results up to which member of a cluster is picked / a permutation of the labels
 * (SURVEY 8c); the fp32 and f32_split entry points stay exact.
 * msm_ms_select_seeds_bf16: workspace as msm_ms_select_seeds (msm_ms_seed_workspace(n) floats); seeds_out are rows of the fp32 X
 * (MS:186-189 returns X[selected]); n >= 16.  flags as msm_ms_select_seeds: maps of >= 65536 rows take ONE persistent launch that
 * keeps 917 504 rows of the copy on chip (VGPRs + LDS) and streams the rest per step; it gives up (indices -1) under the same
 * co-residency condition, and MSM_MS_SEED_STEPWISE selects the one-launch-per-step kernel, whose indices are bit-identical.  msm_ms_hill_climb_bf16: workspace msm_ms_hill_climb_workspace(n, S) floats. */
int64_t msm_ms_bf16_rows(int n);
int msm_ms_pack_bf16(const float* X, int n, int d, void* Xb, void* stream);
int msm_ms_select_seeds_bf16(const void* Xb, const float* X, int n, int d, int num_seeds, int64_t first_index, float* seeds_out,
                             int64_t* indices_out, float* workspace, int64_t workspace_elems, int flags, void* stream);
int msm_ms_hill_climb_bf16(const void* Xb, int n, int d, float* Z, int S, float kappa, int iters, float* workspace,
                           int64_t workspace_elems, void* stream);
/* closest = first argmin_s 0.5*(1 - X.Z_s); labels_out[i] = seed_labels[closest] (int64);
 * counts int64 [num_labels] histogram of labels_out (zeroed here) (MS:206-221). */
int msm_ms_assign(const float* X, int n, int d, const float* Z, int S, const int64_t* seed_labels,
                  int64_t* labels_out, int64_t* counts, int num_labels, void* stream);
/* connected_components of the converged seeds (lib/utils/mean_shift.py:41-76: sequential, order dependent) on the device:
 * Z [S][64] unit rows, S <= 304; seed_labels int64 [S] (labels in order of creation; a later step may overwrite every seed
 * of an earlier label, as in the reference); num_labels int32 [2] = {labels that survive = len(unique(seed_labels)), labels
 * created}; one wave, no host involvement. */
int msm_ms_connected_components(const float* Z, int S, int d, float epsilon, int64_t* seed_labels, int32_t* num_labels,
                                void* stream);
/* swap label 0 with the first-argmax label of counts[0 .. num) (MS:211-227); labels int64 [n] in place.  num = min(num_labels,
 * *num_alive) when num_alive (device, e.g. num_labels[0] of msm_ms_connected_components) is given: the reference counts the
 * labels 0 .. len(unique(seed_labels)) - 1 only, which differs from "every label" once a label has vanished. */
int msm_ms_relabel_largest_zero(int64_t* labels, int n, const int64_t* counts, int num_labels, const int32_t* num_alive,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Instance post-processing for one batch (pretrained_meanshiftformer_model.py:337-343,461-497):
 *   mask_logits [B][Q][h*w] (low-res), query_index int32 [B][T] (selected queries, top-k done by
 *   the caller on the Q*K class scores).  For each selected mask: bilinear upsample to H x W
 *   (align_corners=False), pred_masks [B][T][H*W] = (m > 0), mask_score [B][T] =
 *   sum(sigmoid(m)*[m>0]) / (sum([m>0]) + 1e-6), multiplied by class_scores [B][T] when that pointer
 *   is not NULL (result.scores, :495); boxes [B][T][4] = x0,y0,x1+1,y1+1 (zeros if empty).
 *   workspace floats >= msm_instance_postprocess_workspace(B, T, H, W) (one 32-byte partial per instance and workgroup; no
 *   atomics: the partials are added in strip order by the finishing kernel).
 * ------------------------------------------------------------------------------------------- */
/* Canonical top-k over the Q*K object-class scores of every image (pretrained_meanshiftformer_model.py:
 * 466-474): scores = softmax(pred_logits [B][Q][K+1])[:, :-1] flattened to Q*K entries (index =
 * q*K + class); the reference's topk(sorted=False) order is implementation-defined, here the T
 * winners are returned score-descending with ascending index on ties.
 * scores_out [B][T] float, classes_out [B][T] int64, query_index_out [B][T] int32.  Q*K <= 4096. */
int msm_topk_class_scores(const float* pred_logits, int B, int Q, int K1, int T,
                          float* scores_out, int64_t* classes_out, int32_t* query_index_out, void* stream);
/* The same selection, and in the same launch the selected rows of a per-query matrix: gather_out [B][T][gather_cols] =
 * gather_src [B][Q][gather_ld] rows query_index_out[b][t], leading gather_cols columns (the embeddings the final mask step keeps:
 * replaces an index conversion + torch.gather pair of launches). */
int msm_topk_class_scores_gather(const float* pred_logits, int B, int Q, int K1, int T,
                                 float* scores_out, int64_t* classes_out, int32_t* query_index_out,
                                 const float* gather_src, int64_t gather_ld, int gather_cols, float* gather_out, void* stream);

int64_t msm_instance_postprocess_workspace(int B, int T, int H, int W);
int msm_instance_postprocess(const float* mask_logits, const int32_t* query_index,
                             const float* class_scores, float* pred_masks, float* mask_score, float* boxes,
                             int B, int Q, int T, int h, int w, int H, int W, int Hs, int Ws,
                             float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input projections of the pixel decoder: 1x1 convolution of a backbone feature map to 64 channels, written
 * token-major, with the GroupNorm moments of the result as a by-product (msdeformattn.py:212-220,326-329 input_proj
 * on res3..res5; :225-238,343-347 lateral convolution on res2).
 *   x [B][Cin][HW] (NCHW), bias [64] or NULL
 *   w_packed: the (64, Cin) weight in MFMA fragment order, 64*Cin floats:
 *       w_packed[(((k/8)*4 + o/16)*64 + ((k%8)/2)*16 + o%16)*2 + k%2] = w[o][k]
 *   out: token (b, p) at out + b*out_batch_stride + p*64 (floats) -- a slice of a larger token buffer is allowed
 *   stats [B][64][2] double or NULL: += (sum over p, sum of squares over p) per channel; zeroed here unless
 *   stats_cleared != 0.  Cin a multiple of 128, HW a multiple of 4.  The K sum is split over 8 waves and reduced in a fixed order. */
int msm_conv1x1_in_f32(const float* x, const float* w_packed, const float* bias, float* out, int64_t out_batch_stride,
                       double* stats, int stats_cleared, int B, int Cin, int HW, void* stream);

/* The same for n_levels <= 4 levels in ONE launch (each coarse level alone fills a fraction of the chip): x / w_packed /
 * bias / Cin / HW are HOST arrays of n_levels entries (bias entries may be NULL); level l writes tokens
 * [sum_{i<l} HW[i], +HW[l]) of every image of out [B][out_batch_stride] and moments stats[l] of stats [n_levels][B][64][2].
 * Workgroups are numbered level by level in the order given: pass the deepest-K level first. */
int msm_conv1x1_in_multi_f32(int n_levels, const float* const* x, const float* const* w_packed, const float* const* bias,
                             const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                             int stats_cleared, int B, void* stream);

/* The same two entry points on the bf16 matrix pipe (the bf16 plan; csrc/conv_in.hip, conv_in_lp_tile): both operands as hi + lo
 * bf16 pairs, three K = 32 MFMAs per product (w_l x_h + w_h x_l + w_h x_h), fp32 accumulation and fp32 results -- the error against
 * float64 stays at the fp32 kernels' level (the dropped term is 2^-18 of a product); what changes is the bound: the stream of x
 * instead of the fp32 matrix pipe.  x is split in registers; the weight arrives pre-split:
 *   w_packed: 2 * 64 * Cin bf16,  w_packed[(((g*4 + o/16)*2 + plane)*64 + ((k%32)/8)*16 + o%16)*8 + k%8] = plane(w)[o][k],  g = k/32,
 *   plane 0 = bf16(w), plane 1 = bf16(w - plane 0).  Cin a multiple of 256, HW of 4; everything else as the fp32 entry points. */
int msm_conv1x1_in_lp(const float* x, const void* w_packed, const float* bias, float* out, int64_t out_batch_stride,
                      double* stats, int stats_cleared, int B, int Cin, int HW, void* stream);
int msm_conv1x1_in_multi_lp(int n_levels, const float* const* x, const void* const* w_packed, const float* const* bias,
                            const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                            int stats_cleared, int B, void* stream);
/* msm_conv1x1_in_multi_lp's arithmetic with the packed weight broadcast through LDS (round 6): an eight-wave workgroup covers 8 / KW adjacent
 * 64-pixel tiles x KW slices of K (KW = Cin / the shallowest level's Cin, <= 4) and the 8-KiB packed weight of each 32-deep K group arrives
 * once per workgroup by LDS-DMA -- the deep levels of the 16-bit plans at batch sizes that give every CU a workgroup (the per-tile weight
 * re-reads from L2 bound the _lp form).  Same arguments; K slices are summed in a fixed order. */
int msm_conv1x1_in_multi_wide(int n_levels, const float* const* x, const void* const* w_packed, const float* const* bias,
                              const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                              int stats_cleared, int B, void* stream);

/* The decoder's attention masks at the resolution they are used at (meanshiftformer_transformer_decoder.py:668-680; csrc/attn_mask.hip).
 * interpolate(einsum(e, F), size, bilinear, align_corners=False) = einsum(e, interpolate(F)): the two act on different axes.
 * msm_pool_mask_taps: act [B][64][H][W] (the 64-channel factored mask features, NCHW planes) -> for each of n_levels target sizes
 *   th[l] x tw[l] (HOST arrays; H / th = W / tw in {2, 4, 8}) out[l] [B][th*tw][64] token-major = the bilinear reduction of act
 *   (mean of the four centre taps of every p x p cell).  One launch.  zero_buf (nullable): zero_count int32 words cleared by
 *   the same launch (the row_any flags of the first msm_attn_mask_pooled call: no fill launch).
 * msm_attn_mask_pooled: attn [B][Q][T] bytes = (sum_c embed[b][q][c] pooled[b][t][c] + qbias[b][q]) < 0 for the 64-column embedding
 *   (row stride embed_ld floats, batch stride Q * embed_ld; qbias NULL or element stride qbias_ld) and row_any [B][Q] = 1 where a
 *   row keeps an unmasked key (zeroed here unless row_any_cleared != 0).  Q <= 112.
 *   flags & 1 (T % 16 == 0): attn receives the mask bit-packed and blocked instead -- msm_attn_mask_bits_bytes(B, Q, T) bytes in the layout
 *   of msm_attn_pack_mask_bits, what msm_hypersphere_attn_fused_kv_fwd reads (word 7 of a query's eight is never written nor read).
 *   flags & 2 (ABI 18; 16-bit plans): embedding and pooled activation enter two v_mfma_f32_16x16x32_f16 as IEEE halves (clamped), fp32
 *   accumulation, instead of sixteen dependent fp32 MFMAs per (query block, key block).
 *   flags & 4 (ABI 21; the 16-bit plans' default since round 6; not with flags & 2): both operands as hi + lo IEEE-half pairs (2^-22), three terms per
 *   product -- six K = 32 MFMAs per key block with fp32-class logits (the mask bits feed back into the attention). */
int msm_pool_mask_taps(const float* act, int B, int H, int W, int n_levels, const int32_t* th, const int32_t* tw,
                       float* const* out, int32_t* zero_buf, int64_t zero_count, void* stream);
int msm_attn_mask_pooled(const float* embed, int64_t embed_ld, const float* qbias, int64_t qbias_ld, const float* pooled,
                         uint8_t* attn, int32_t* row_any, int row_any_cleared, int flags, int B, int Q, int T, void* stream);

/* The FPN output convolution (msdeformattn.py:264-279, 349-351: Conv2d(64, 64, 3, padding=1) in front of a GroupNorm):
 *   in / out [B][H*W][64] token maps, w_tap_major [64][9*64] with k = (dy*3 + dx)*64 + c_in (zero padding), no bias (a
 *   norm follows); stats [B][64][2] double or NULL: += (sum, sum of squares) of out per (image, channel), zeroed here unless
 *   stats_cleared != 0.  The whole weight is held in LDS (one workgroup per CU). */
int msm_conv3x3_c64_f32(const float* in, const float* w_tap_major, float* out, double* stats,
                        int stats_cleared, int B, int H, int W, void* stream);
/* Low-precision mode of the same convolution (same arguments, fp32 in / out): the weight rounded to bf16 as it enters LDS, the
 * activations as hi + lo bf16 operands, v_mfma_f32_16x16x32_bf16 with fp32 accumulation; moments from the fp32 results. */
int msm_conv3x3_c64_bf16(const float* in, const float* w_tap_major, float* out, double* stats,
                         int stats_cleared, int B, int H, int W, void* stream);
/* ... with IEEE-half operands (precision "f16"): weight and activations one fp16 term each (activations clamped to the half range),
 * v_mfma_f32_16x16x32_f16: half the MFMAs of the bf16 form, 2^-12 roundings instead of the weight's 2^-9. */
int msm_conv3x3_c64_f16(const float* in, const float* w_tap_major, float* out, double* stats,
                        int stats_cleared, int B, int H, int W, void* stream);
/* msm_conv3x3_c64_f16 on an input that already is the clamped halves (in_f16 [B][H*W][64] IEEE halves, msm_groupnorm_apply_f16):
 * the same output bits; a wave takes up to three output rows of a 32-pixel strip with all of its loads in flight at once and the
 * dx = -1 / +1 operands as lane shifts of a row loaded once.  out / stats as msm_conv3x3_c64_f32 (the moments' fp32 partial sums are
 * grouped by unit, so they agree with msm_conv3x3_c64_f16's to rounding). */
int msm_conv3x3_c64_f16h(const void* in_f16, const float* w_tap_major, float* out, double* stats,
                         int stats_cleared, int B, int H, int W, void* stream);
/* The same convolution with fp32-accurate results on the bf16 matrix pipe (f32_split plan): the activation as the three bf16
 * planes of msm_groupnorm_apply_split, the weight (fp32, tap-major) split when a workgroup copies its 32 output channels
 * into LDS, six bf16 MFMAs per product.  out / stats as msm_conv3x3_c64_f32. */
int msm_conv3x3_c64_split(const uint16_t* planes, const float* w_tap_major, float* out, double* stats,
                          int stats_cleared, int B, int H, int W, void* stream);

/* The same kernel with a planar result: out [B][Cout][H*W] (NCHW) = bias + conv3x3(in), Cout a multiple of 64 (every slice of
 * 64 output channels has its own workgroups and its own 147 KB of the [Cout][9*64] weight in LDS), W % 4 == 0
 * (SimpleBasePixelDecoder.mask_features: Conv2d(64, 256, 3, padding=1), fpn.py:237-246; 90.6 GFLOP per 640x480 frame). */
int msm_conv3x3_c64_nchw_f32(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                             int Cout, void* stream);
/* ... in the low-precision mode of msm_conv3x3_c64_bf16 (weight rounded to one bf16 as it is copied into LDS, activations as hi + lo
 * operands, fp32 accumulation and fp32 planes out): the UCN path's mask_features convolution under set_precision("bf16"). */
int msm_conv3x3_c64_nchw_bf16(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                              int Cout, void* stream);
int msm_conv3x3_c64_nchw_f16(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                             int Cout, void* stream);      /* the IEEE-half operand form (msm_conv3x3_c64_f16) */

/* Encoder prologue: everything between the input projections and the first deformable-attention layer in one pass
 * over the token buffer (msdeformattn.py:326-329 GroupNorm of input_proj, :60-75 level concatenation;
 * ops/modules/ms_deform_attn.py:95-104 layer 0's value_proj / sampling_offsets / attention_weights):
 *   src   = GroupNorm_l(raw)        raw [B][S][64]: msm_conv1x1_in_f32 outputs of the n_levels levels, concatenated;
 *                                   level l of an image covers tokens [level_starts[l], level_starts[l+1]) (HOST array of
 *                                   n_levels+1 ints, 0 .. S); stats [n_levels][B][64][2] double moments (same call);
 *                                   gn_params [n_levels][2][64] = gamma, beta; src_out may alias raw
 *   value = value_proj(src)         value_out [B][S][64] or, value_heads > 0, head-major [B][heads][S][64/heads]
 *   proj  = [offsets|weights](src + pos)    proj_out [B][S][proj_width], pos [S][64]
 *   wstream: value_proj weight (64,64) then the (proj_width,64) weight, as consecutive 16-row blocks of 1024 floats,
 *   zero-padded to msm_encoder_prologue_stream_floats(proj_width); small = [value_proj bias (64) | proj bias].
 * n_levels <= 4, S >= 86, proj_width a multiple of 16 and <= 512 (the weights are held in LDS).
 * out_bf16_hm != 0 (8 heads, proj_width 288): value_out / proj_out are the bf16 plan's head-major tensors -- value [B][8][S][8] fp16,
 * proj [B][8][120 S bytes] (fp32 offsets + fp16 logits, plane-major; see msm_encoder_block_hm_fwd) -- instead of fp32. */
int64_t msm_encoder_prologue_stream_floats(int proj_width);
int msm_encoder_prologue_fwd(const float* raw, const double* stats, const float* gn_params, const int32_t* level_starts,
                             int n_levels, int groups, float gn_eps, const float* wstream, const float* small,
                             const float* pos, float* src_out, void* value_out, void* proj_out, int B, int S,
                             int proj_width, int value_heads, int out_bf16_hm, void* stream);
/* The same prologue for the bf16 plan (csrc/enc_lp.hip, enc_prologue_hm_kernel): value [B][8][S][8] in fp16 and the sampling
 * projection [B][8][120 S bytes] (fp32 offsets + fp16 logits, plane-major), the two projections on the bf16 matrix pipe with hi + lo operands (132 MFMAs of 16 cycles per 16-token tile
 * instead of 352 of 32).  wblocks: msm_encoder_prologue_hm_weight_bytes() bytes = the value_proj blocks (16 KiB) and the projection
 * blocks (72 KiB) exactly as msm_encoder_block_hm_fwd's stream holds them ([row block][k-group][hi, lo] 1-KiB fragments, rows in the
 * value order / the reference's [192 offsets | 96 logits] order); small: bv [64] and bp [288] in those row orders. */
int64_t msm_encoder_prologue_hm_weight_bytes(void);
int msm_encoder_prologue_hm_fwd(const float* raw, const double* stats, const float* gn_params, const int32_t* level_starts, int n_levels,
                                int groups, float gn_eps, const void* wblocks, const float* small, const float* pos, float* src_out,
                                void* value_out, void* proj_out, int B, int S, void* stream);

/* The fp32 encoder-layer tail on the bf16 matrix pipe (csrc/enc_block_split.hip): every fp32 operand is split exactly into three
 * bf16 terms and a product is the six bf16 MFMAs of weight >= 2^-18 with fp32 accumulation -- fp32-accurate results (the
 * dropped terms are below 2^-26 of a product) at 6/16 of the fp32 MFMA's cost.  Same arguments as msm_encoder_block_fwd except the weight stream;
 * wstream is the triple-split weight stream (blocks of [4 k-groups][64 lanes][4 bf16]; a logical block = its h, m, l blocks;
 * 12 blocks per stage: output_proj | two hidden blocks [W1 h,m,l, W2 h,m,l] per stage | value_proj | four proj row blocks per
 * stage, zero padded), msm_encoder_block_split_stream_bytes(d_ffn, proj_width) bytes. */
int64_t msm_encoder_block_split_stream_bytes(int d_ffn, int proj_width);
int msm_encoder_block_split_fwd(const float* attn, const float* src, const void* wstream, const float* small, const float* pos,
                                float* src_out, float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn,
                                int proj_width, int value_heads, float eps, void* stream);
/* The low-precision encoder block (BASELINE configs 3 / 5; the reference's low-precision mode is autocast over the whole model,
 * MSMFormer/tabletop_train_net_pretrained.py:232) on the same kernel structure (K = 32 bf16 MFMAs, two token tiles per wave):
 * bf16 MFMA operands, fp32 accumulation; the residual stream, the LayerNorms, the biases and every output stay fp32.  Operand
 * roundings: the three 64-wide projections w(h + m) x(h + m) without the m x m term (w = h + m, both bf16), linear1 w(h) x(h + m),
 * linear2 single bf16 operands -- and only the copies that are read in the stream: 12 blocks of [2 k-groups][64 lanes][8 bf16] per
 * stage = output_proj [h, m] x 4 row blocks (+ 4 zero blocks) | three hidden pairs [W1(q0) h, W1(q1) h, W2 h (4 KiB)] per stage,
 * the hidden dimension zero-padded to whole stages | value_proj like output_proj | six proj row blocks [h, m] per stage. */
int64_t msm_encoder_block_lp_stream_bytes(int d_ffn, int proj_width);
int msm_encoder_block_lp_fwd(const float* attn, const float* src, const void* wstream, const float* small, const float* pos,
                             float* src_out, float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn,
                             int proj_width, int value_heads, float eps, void* stream);

/* ---- the bf16 plan's encoder layers with head-major bf16 activations between the kernels (csrc/enc_lp.hip) ----------------
 * Replaces, per layer, MSDeformAttn.forward (ops/modules/ms_deform_attn.py:95-125) + the rest of
 * MSDeformAttnTransformerEncoderLayer.forward (pixel_decoder/msdeformattn.py:116-131) when the model runs in the low-precision
 * mode (BASELINE configs[2] / configs[4]).  value_hm / attn_hm [B][8 heads][S][8 dims] are IEEE half (fp16, not bf16: same bytes,
 * three more mantissa bits).  proj_hm [B][8][120 S bytes] holds, per (image, head), the head's 24 sampling offsets ((level, point,
 * xy) order) of every query as FLOAT32 and its 12 attention logits as fp16, plane-major: six planes [S][4 floats] (offsets 4 p ..
 * 4 p + 3), then three planes [S][4 halves] (logits 4 p .. 4 p + 3) -- offsets are pixel distances of several pixels, and an fp16
 * offset (2^-11 |o| ~ 2e-3 pixel) alone put 6e-3 of relative error into the encoder output (round 5; bf16 offsets: 5e-2); planes make
 * the producers' stores contiguous over the 16 consecutive tokens of a tile.
 * The matrix pipe multiplies bf16 operands (an fp16 value is a hi + lo bf16 pair exactly); the residual stream stays fp32.
 *
 * msm_encoder_block_hm_fwd: src_out = LN2(x + linear2(relu(linear1(x)))), x = LN1(src + output_proj(attn));
 *   value_out / proj_out (both null for the last layer) = the NEXT layer's value_proj(src_out) and
 *   [sampling_offsets | attention_weights](src_out + pos).
 *   wstream (msm_encoder_block_hm_stream_bytes(d_ffn, with_next) bytes of bf16 bit patterns): 1-KiB blocks in the A-operand
 *   order of v_mfma_f32_16x16x32_bf16 ([kq = 4][row = 16][8 bf16] = lane kq*16 + row): resident 32 KiB = output_proj
 *   [rb 4][G 2][h, l] (k natural: attn feature 32 G + 8 kq + j) | value_proj [rb 4][G 2][h, l] (row 16 rb + 4 lq + r = head
 *   4 (rb >> 1) + lq, dim 4 (rb & 1) + r; k order L: feature (2 G + (j >> 2)) 16 + 4 kq + (j & 3)); then per pair P of 16-wide
 *   hidden blocks 8 KiB = W1 [q 2][G 2] (rows hidden 16 (2 P + q) + i, k order L) | W2 [ob 4] (rows feature 16 ob + i, k = hidden
 *   (2 P + (j >> 2)) 16 + 4 kq + (j & 3)); four pairs per 32-KiB stage, the hidden dimension zero-padded to whole stages; then
 *   (with_next) three stages of eight projection row blocks [rb][G 2][h, l] (rows: the offsets of all heads, 24 head + c, then the
 *   logits, 192 + 12 head + c; k order L), zero padded.
 *   small (msm_encoder_block_hm_small_floats(d_ffn) floats) = output_proj bias | norm1 w | norm1 b | linear2 bias | norm2 w |
 *   norm2 b | value_proj bias (row order above) | projection bias (288, the row order above) | linear1 bias (zero padded).
 *   M = B * tokens_per_image tokens; pos [tokens_per_image][64].
 *   ffn_f16 != 0 (precision "f16"): the W1 / W2 blocks of the FFN stages hold IEEE-half bit patterns (same layout), linear1 takes x
 *   and linear2 the hidden activation as one fp16 term each (v_mfma_f32_16x16x32_f16; 8 instead of 12 MFMAs per pair of hidden
 *   blocks, roundings of 2^-12 where the bf16 form has 2^-9); the resident / projection blocks stay [h, l] bf16 pairs. */
int64_t msm_encoder_block_hm_stream_bytes(int d_ffn, int with_next);
int msm_encoder_block_hm_small_floats(int d_ffn);
int msm_encoder_block_hm_fwd(const void* attn_hm, const float* src, const void* wstream, const float* small, const float* pos,
                             float* src_out, void* value_out, void* proj_out, int M, int tokens_per_image, int d_ffn, float eps,
                             int ffn_f16, void* stream);
/* msm_msdeform_attn_enc_lp_fwd: out_hm = MSDeformAttn core (ms_deform_im2col_cuda.cuh:242-304) over the fp16 value_hm with the
 *   sampling offsets / attention logits of proj_hm (encoder self-attention: reference points = pixel centres,
 *   msdeformattn.py:141-153; softmax over the 12 logits, ms_deform_attn.py:102-109).  Shipped geometry only: M = 8, D = 8, L = 3, P = 4.
 * msm_msdeform_attn_enc_lp_fused_fwd: the same with the projection computed in the kernel from src + pos (measured slower than
 *   the stored projection; kept as the tested alternative).  wpack: per head 12 KiB = [rb 3][G 2][h, l] blocks (k order L)
 *   of the head's 24 offset rows, 12 logit rows and 12 zero rows; bpack [M][48] their biases (fp32). */
int msm_msdeform_attn_enc_lp_fwd(const void* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const void* proj_hm, void* out_hm, int B, int S, int M, int D, int L, int P, void* stream);
int msm_msdeform_attn_enc_lp_fused_fwd(const void* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const float* src, const float* pos, const void* wpack, const float* bpack, void* out_hm, int B,
                                       int S, int M, int D, int L, int P, void* stream);
/* fp32 -> fp16, round to nearest even, clamped to the half range; n a multiple of 8 */
/* Elementwise glue of the backbones (csrc/backbone_ops.hip; the convolutions stay MIOpen / hipBLASLt calls):
 * msm_bias_act_nhwc: x[p][c] = act(x[p][c] + bias[c] (+ residual[p][c])) in place on a channels_last map of `pixels` x C values; dtype 0: fp32
 *   (C % 4 == 0), 1: bf16, 2: IEEE half (C % 8 == 0; bias and residual in the same type; fp32 arithmetic, one rounding); relu != 0: max(., 0).  Replaces
 *   the bias kernel MIOpen appends to a convolution, F.relu, the residual add and its ReLU of detectron2's BottleneckBlock forward.
 * msm_nhwc_to_nchw_f32: in [B][HW][C] (dtype as above) -> out [B][C][HW] fp32. */
int msm_bias_act_nhwc(void* x, const void* bias, const void* residual, int relu, int64_t pixels, int C, int dtype, void* stream);
int msm_nhwc_to_nchw_f32(const void* in, float* out, int B, int C, int HW, int dtype, void* stream);
/* The tail of the UCN RGB-D backbone in one pass (lib/networks/SEG.py:97-117 + pretrained_meanshiftformer_model.py:298-300):
 * out [B][64][H][W] = N(..N(up(a) + up(b2))), up = bilinear with align_corners=True from the towers' [B][h][w][64] fp32 (channels_last) maps
 * (b2 NULL: one tower), N(v) = v / max(|v|_2, eps) over the 64 channels applied `norms` (0..2) times. */
int msm_ucn_embedding_tail(const float* a, const float* b2, float* out, int B, int h, int w, int H, int W, int norms, float eps, void* stream);
int msm_f32_to_f16(const float* in, void* out, int64_t n, void* stream);
/* the same for B rows of n floats spaced in_batch_stride floats apart (n % 8 == 0, stride % 4 == 0): out is contiguous [B][n] */
int msm_f32_to_f16_rows(const float* in, void* out, int B, int64_t n, int64_t in_batch_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Label-image statistics of the two-stage harness: one pass instead of the reference's per-label
 * unique()/masked-reduction/.item() loops (lib/fcn/test_dataset.py:62-112 crop_rois +
 * lib/utils/mask.py:179-186 tight boxes, test_dataset.py:121-126 overlap test, :183-198 depth filter).
 *   labels [B][H][W] float, integer valued in [0, k); weight [B][H][W] float or NULL.
 *   stats  [B][k][5] int32 = area, x_min, y_min, x_max, y_max (W, H, -1, -1 for an absent label)
 *   wsum   [B][k] float    = sum of weight over the label's pixels (0 when weight is NULL); fp32 sum in
 *                            unspecified order: exact for 0/1 weights (counts < 2^24)
 *   overflow [B] int32     = pixels whose value lies outside [0, k) (they are counted in the clamped bin;
 *                            callers treat a non-zero count as an error)
 * k <= 2048 (per-workgroup LDS table). */
int msm_label_stats(const float* labels, const float* weight, int32_t* stats, float* wsum, int32_t* overflow,
                    int B, int H, int W, int k, void* stream);

/* Batched two-stage harness (lib/fcn/test_utils.py:375-406 walks frames and crops one at a time):
 *   msm_label_image   combine_masks(get_confident_instances(...)) (test_utils.py:35-52, 93-112) for B images: masks [B][K][H][W]
 *                     (non-zero = inside), inst_labels [B][K] = the label an instance carries (2 + kept instances before it; 0 =
 *                     dropped) -> out [B][H][W] = per-pixel maximum ("later instances overwrite earlier ones").
 *   msm_crop_resize   crop_rois (lib/fcn/test_dataset.py:62-112) for N ROIs of any frames in one launch: table [N][8] int32 =
 *                     frame, label, x0, y0, x1, y1 (inclusive), 2 unused; rgb / depth [F][3][H][W] bilinear with
 *                     align_corners=True (F.upsample_bilinear, :104,109), mask = (labels[frame] == label) nearest (:106)
 *                     -> rgb_out / depth_out [N][3][S][S], mask_out [N][S][S].  depth / depth_out may both be NULL.
 *   msm_paste_labels  paste-back of match_label_crop (test_dataset.py:160-177): renum [N][S][S] renumbered crop labels, order
 *                     [N] the crops grouped by frame in paste order, frame_start [F+1]; refined [F][H][W] takes, per pixel, the
 *                     last crop in order that covers it with a non-zero (nearest-resized) value, else 0. */
int msm_label_image(const float* masks, const float* inst_labels, float* out, int B, int K, int H, int W, void* stream);
int msm_crop_resize(const float* rgb, const float* depth, const float* labels, const int32_t* table, float* rgb_out,
                    float* depth_out, float* mask_out, int N, int H, int W, int S, void* stream);
int msm_paste_labels(const float* renum, const int32_t* table, const int32_t* order, const int32_t* frame_start,
                     float* refined, int F, int H, int W, int S, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSM_HIP_H */
