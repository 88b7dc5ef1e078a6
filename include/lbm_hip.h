/* liblbm_hip.so — C ABI of the MI355X (gfx950) kernels for the LBM online point-tracking step.
 *
 * This is the drop-in boundary of the hot path: plain pointers and sizes, no torch types.  Every
 * pointer is DEVICE memory owned by the caller (in this repo: PyTorch-allocated tensors, passed as
 * tensor.data_ptr()); the library allocates nothing, keeps no tracking state (the ring-buffer head is a
 * caller-held int) and enqueues all work on the caller's HIP stream without synchronising.  Process-wide state it does
 * keep: the debug knobs (lbm_hip_debug_set / _ptr), the profiling records between lbm_hip_profile_begin / _end, and four
 * timing-disabled HIP events created on first use by lbm_hip_track_step when n_aux_streams > 0 (fork / join of the
 * auxiliary streams; concurrent calls with auxiliary streams from several host threads are therefore not supported).
 * All functions return 0 on success and a negative value on error; the message is available from
 * lbm_hip_last_error() (thread-local).  Nothing throws across the ABI.
 *
 * Reference citations are file:line in George-Zhuang/lbm (`/root/reference`).
 * Conventions: C = 256 channels, 4 heads x 64, 4 pyramid levels; "R" = B*N tracked points.
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lbm_stream_t; /* hipStream_t */

#define LBM_HIP_ABI_VERSION 5
#define LBM_HIP_MAX_LAYERS 8

int lbm_hip_version(void);
const char* lbm_hip_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense block.  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) + res_scale * RES[(m / res_div) % res_mod, :]
 * then, if ln_gamma != NULL (N must be 256): Y = LayerNorm(Y) * gamma + beta, rows with keep[m] == 0
 * replaced by fallback[m, :], and Yn (optional) = Y / max(|Y|_2, 1e-12).
 * Replaces: nn.Linear / 1x1 nn.Conv2d / nn.LayerNorm / MLP call sites of transformer.py:114-149,
 * 268-338, 343-355; msdeformattn.py:146-153,179; backbone.py:53-63.  act: 0 none, 1 exact GELU.
 * `batch` > 1 repeats the product with element strides sX/sW/sB/sY/sR (gridDim.z).
 * Requirements: K % 32 == 0, ldx/ldw % 4 == 0, X/W 16-byte aligned.                            */
typedef struct {
  const float* X; const float* W; const float* bias; float* Y; const float* res;
  int32_t M, N, K, ldx, ldw, ldy, ldres, res_div, res_mod;
  float res_scale;
  int32_t act, batch;
  int64_t sX, sW, sB, sY, sR;
  const float* ln_gamma; const float* ln_beta; const uint8_t* keep; const float* fallback; float* Yn;
  /* split-precision path (see lbm_hip_split_planes): when both are non-NULL, W is ignored and the product runs on
   * the f16 matrix cores with fp32-grade accuracy (three MFMAs per block, fp32 accumulate); ldw % 8 == 0.       */
  const void* W_hi; const void* W_lo;
} lbm_hip_linear_desc;
int lbm_hip_linear(const lbm_hip_linear_desc* d, lbm_stream_t stream);

/* fp32 x[n] -> two f16 planes with x = hi + lo * 2^-11 (hi = f16(x), lo = f16((x - hi) * 2^11); |x| clamped to
 * 65000).  This is the operand format of the split-precision GEMM / convolution: weights are split once per
 * model, per-frame operands (correlation maps) by their producing kernel.  n % 4 == 0.                         */
int lbm_hip_split_planes(const float* x, void* hi, void* lo, int64_t n, lbm_stream_t stream);

/* Dense block with BOTH operands pre-split (f16 hi plane at the pointer, lo plane `*_plane` elements further):
 * Y = act(A . B^T + bias) + res_scale * RES[(m / res_div) % res_mod], written as fp32 (Y) and/or as split planes
 * (Ys) for the next product.  Operands are staged by LDS-DMA, three f16 MFMAs per 32x32x16 block, fp32 accumulate.
 * K % 32 == 0; lda, ldb, planes and batch strides multiples of 8; act / residual as in lbm_hip_linear.        */
typedef struct {
  const void* A; const void* B; int64_t a_plane, b_plane;
  const float* bias; float* Y; void* Ys; int64_t ys_plane; const float* res;
  int32_t M, N, K, lda, ldb, ldy, ldys, ldres, res_div, res_mod;
  float res_scale;
  int32_t act, batch;
  int64_t sA, sB, sBias, sY, sYs, sR;
  int32_t planes;   /* 0 or 2: hi and lo planes multiplied (fp32-grade); 1: hi planes only (plain f16 operands) */
} lbm_hip_linear_ss_desc;
int lbm_hip_linear_ss(const lbm_hip_linear_ss_desc* d, lbm_stream_t stream);

/* Row-wise LayerNorm over C = 256 (+ optional keep/fallback select and L2-normalised copy).
 * Replaces nn.LayerNorm (transformer.py:31,163,...) and the valid-row scatter of transformer.py:279,301. */
int lbm_hip_layernorm(const float* X, const float* gamma, const float* beta, float* Y, int32_t rows,
                      const uint8_t* keep, const float* fallback, float* Yn, lbm_stream_t stream);

/* Y = X / max(|X|_2, 1e-12) per 256-channel row.  Replaces F.normalize (transformer.py:207-212). */
int lbm_hip_l2norm(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t rows, lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Single-query 4-head attention read-out over a per-point sequence of S rows (memory ring buffer or
 * lattice tokens).  U[R,4,256] are the per-head queries already mapped to key space
 * (u_h = Wk_h^T (Wq_h x + bq_h) / 8); out[R,4,256] = sum_j softmax_j(u_h . (seq_j + temb_j)) seq_j.
 * Logical position j lives in physical slot (ring_head + j) % S; masks are in physical order,
 * non-zero = ignore; valid[r] = 0 when every position of row r is ignored (then out[r] = 0).
 * Replaces the nn.MultiheadAttention calls of transformer.py:272, 294, 333 (together with two
 * lbm_hip_linear calls using weights folded by lbm_amd/engine.py: fold_attention / TrackerWeights).                              */
int lbm_hip_attn_readout(const float* U, const float* seq, int64_t point_stride, int64_t slot_stride,
                         int32_t S, int32_t ring_head, const float* temb, const uint8_t* mask_a,
                         const uint8_t* mask_b, float* out, uint8_t* valid, int32_t R, lbm_stream_t stream);
/* The same read-out over an f16 sequence (strides in elements): the memories of the f16-storage mode (step precision 3).  The f16
 * values are widened exactly, so the result equals lbm_hip_attn_readout on the widened sequence bit for bit.  temb is required. */
int lbm_hip_attn_readout_h16(const float* U, const void* seq_f16, int64_t point_stride, int64_t slot_stride,
                             int32_t S, int32_t ring_head, const float* temb, const uint8_t* mask_a,
                             const uint8_t* mask_b, float* out, uint8_t* valid, int32_t R, lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * tokens[B, L, 256], level 0 in rows [0, h*w): fills levels 1..3 (avg-pool 2/4/8) in place.
 * Replaces LBMTransformer.get_multiscale_feats (transformer.py:51-71).                          */
int lbm_hip_pyramid(float* tokens, int32_t B, int32_t h, int32_t w, lbm_stream_t stream);

/* Correlation maps F[n_layers, B, h*w, 256] such that corr = q_hat . F^T + bias reproduces
 * LBMTransformerLayer.correlation (transformer.py:193-229).  corr_mix_host: n_layers x 4 corr_conv
 * weights in HOST memory; inv_norm[B, L] is scratch.                                            */
int lbm_hip_corr_maps(const float* tokens, float* inv_norm, float* F, const float* corr_mix_host,
                      int32_t n_layers, int32_t B, int32_t h, int32_t w, lbm_stream_t stream);

/* Correlation product and top-k in one pass: for every point the k cells with the largest q_hat . F_cell (ties -> lowest cell) and
 * their centres, without the [N, P0] score matrix ever reaching memory.  Replaces the same lines as lbm_hip_corr_maps' consumer +
 * lbm_hip_topk together (transformer.py:306-308: torch.einsum / conv score + torch.topk; coord_utils.py:4-29).
 * q: the L2-normalised point tokens as split f16 planes [B, N, 256] (hi plane; lo plane q_plane ELEMENTS further), F: one layer of
 * lbm_hip_corr_maps' output as split planes [B, P0, 256] (f_plane likewise).  k in 1..9.  workspace: device scratch of
 * lbm_hip_corr_topk_workspace_bytes(B, N, P0, k) bytes (per-point candidate lists).  val_out (optional): the k scores.            */
int64_t lbm_hip_corr_topk_workspace_bytes(int32_t B, int32_t N, int32_t P0, int32_t k);
int lbm_hip_corr_topk(const void* q_hi, int64_t q_plane, const void* f_hi, int64_t f_plane, int32_t B, int32_t N, int32_t P0,
                      int32_t k, int32_t grid_w, int32_t stride_px, void* workspace, int64_t workspace_bytes, int32_t* idx_out,
                      float* ref_out, float* val_out, lbm_stream_t stream);

/* Top-k per row (ties -> lowest index) + cell centres (x, y) in network pixels.
 * Replaces torch.topk + indices_to_coords (transformer.py:307-308, coord_utils.py:4-29).        */
int lbm_hip_topk(const float* corr, int64_t ld, int32_t P, int32_t k, int32_t rows, int32_t grid_w,
                 int32_t stride_px, int32_t* idx_out, float* ref_out, float* val_out, lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention sampling core = multi_scale_deformable_attn_pytorch
 * (msdeformattn.py:188-235) fused with the offset normalisation / softmax of
 * MultiScaleDeformableAttention.forward (msdeformattn.py:150-170).  See msda.hpp for the two modes. */
typedef struct {
  const float* value; int64_t v_batch_stride; int32_t ldv;
  int32_t lvl_h[4], lvl_w[4], lvl_start[4];
  int32_t B, N, k, P;
  const float* offlog; int32_t ld_offlog, logit_col;
  const float* ref; float net_w, net_h;
  const float* loc; const float* attw;
  float* out;
  int32_t value_f16;      /* 1: `value` points at f16 elements (same element strides), implicit mode only: the f16-storage mode's value maps */
} lbm_hip_msda_desc;
int lbm_hip_msda_sample(const lbm_hip_msda_desc* d, lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder side.  preprocess: backbone.py:181-183; upcat: backbone.py:57-62 (pixel-major 64-channel
 * maps -> [B*h*w, 192]); sample_queries: lbm_online.py:72-81 / backbone.py:88-92.               */
int lbm_hip_preprocess(const float* frames, int64_t frame_bstride, float* out, int32_t B, int32_t H, int32_t W,
                       int32_t oh, int32_t ow, int32_t out_nhwc /* 0: [B,3,oh,ow]; 3 or 4: [B,oh,ow,c], channel 3 zero */,
                       lbm_stream_t stream);
/* Video ingest: the same pre-processing straight from the decoder's bytes, so frames cross PCIe and HBM as uint8 (a quarter
 * of the bytes of demo_utils.py:127-131's float tensor) and the BGR->RGB / HWC->CHW / float conversions of the host loader
 * (demo_utils.py:127-131) happen in the kernel.  in_layout 0: planar (B,3,H,W) RGB; 1: interleaved (B,H,W,3) RGB;
 * 2: interleaved (B,H,W,3) BGR (OpenCV's order).  frame_bstride in bytes (= elements).  Bit-identical to lbm_hip_preprocess on
 * the same pixel values.                                                                                                   */
int lbm_hip_preprocess_u8(const uint8_t* frames, int64_t frame_bstride, int32_t in_layout, float* out, int32_t B, int32_t H, int32_t W,
                          int32_t oh, int32_t ow, int32_t out_nhwc, lbm_stream_t stream);
/* The same pre-processing into f16 hi / lo planes [2][B, oh, ow, 4] (value = hi + lo / 2048, channel 3 zero): what lbm_hip_stem_pool
 * reads.  is_u8 = 0: planar fp32 frames as lbm_hip_preprocess (in_layout 0); 1: bytes as lbm_hip_preprocess_u8.  out_plane in elements. */
int lbm_hip_preprocess_split(const void* frames, int32_t is_u8, int64_t frame_bstride, int32_t in_layout, void* out_hi, int64_t out_plane,
                             int32_t B, int32_t H, int32_t W, int32_t oh, int32_t ow, lbm_stream_t stream);
/* The ResNet stem in one launch (backbone.py:48-50 = torchvision conv1 / bn1 / relu / maxpool with the BatchNorm folded into W and
 * bias): 7 x 7 stride-2 pad-3 convolution 4 -> 64 channels over split planes X [2][B, H, W, 4], ReLU, 3 x 3 stride-2 pad-1
 * max-pooling; Y fp32 and / or Ys split planes [B, H / 4, W / 4, 64].  W_hi / W_lo: [64, ldw] with columns (ky, kx, c), c < 4, as
 * lbm_hip_conv2d_nhwc takes them.  H and W must be multiples of 4.  Replaces lbm_hip_conv2d_nhwc + lbm_hip_maxpool3x3s2_nhwc_s
 * for the stem: same values up to the summation order inside the matrix cores (about 1e-7 relative).                          */
int lbm_hip_stem_pool(const void* X, int64_t x_plane, const void* W_hi, const void* W_lo, int32_t ldw, const float* bias, float* Y, void* Ys,
                      int64_t ys_plane, int32_t B, int32_t H, int32_t W, lbm_stream_t stream);
int lbm_hip_upcat(const float* p1, const float* p2, const float* p3, float* out, int32_t B, int32_t h, int32_t w,
                  lbm_stream_t stream);
int lbm_hip_sample_queries(const float* tokens, int64_t tok_bstride, int32_t ld, const float* pts, float* out,
                           int32_t B, int32_t N, int32_t h, int32_t w, int32_t net_h, int32_t net_w,
                           lbm_stream_t stream);

/* ResNet-18 body (backbone.py:20-26, 49-51; torchvision BasicBlock) on pixel-major activations:
 * Y[b,oy,ox,:] = relu?(conv(X)[...] + bias (+ res)), BatchNorm folded into W/bias by the caller.
 * X [B,H,W,Cin] with Cin a power of two >= 4 (the RGB stem is padded to 4 channels); W [Cout, ldw] with
 * columns ordered (ky, kx, ci) and zero-padded to a multiple of 32 (ldw = 0 selects exactly that).
 * W_lo == NULL: W is fp32 (exact fp32 MFMA).  W_lo != NULL: W is the f16 hi plane and W_lo the f16 lo plane of
 * the same packed matrix (split-precision path).  maxpool: 3x3 / stride 2 / pad 1 (backbone.py:23).        */
int lbm_hip_conv2d_nhwc(const float* X, const void* W, const float* bias, const float* res, float* Y,
                        const void* W_lo, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int32_t Cout,
                        int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t relu, int32_t ldw,
                        lbm_stream_t stream);
int lbm_hip_maxpool3x3s2_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C,
                              lbm_stream_t stream);

/* Split-precision encoder (same reference lines as above).  A "split tensor" is two f16 planes: hi at the pointer,
 * lo `*_plane` elements further (lbm_hip_split_planes).
 * conv2d_nhwc_ss: activation X [B,H,W,Cin] (Cin a power of two >= 32) and packed weights both split; im2col rows are
 *   gathered straight into LDS by DMA, out-of-image taps read `zeros` (>= 16 zero bytes of device memory).  Output as
 *   fp32 (Y) and/or split (Ys).  `scratch` (optional, fp32) lets small problems split K over more workgroups.
 *   W_lo == NULL: only the hi planes of X and W are multiplied (reduced precision, as step precision 2).
 * maxpool / upcat / pyramid *_s variants additionally (or only) emit the split twin of their output.             */
int lbm_hip_conv2d_nhwc_ss(const void* X, int64_t x_plane, const void* W_hi, const void* W_lo, const float* bias,
                           const float* res, float* Y, void* Ys, int64_t ys_plane, const void* zeros, void* scratch,
                           uint64_t scratch_bytes, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int32_t Cout, int32_t kh,
                           int32_t kw, int32_t stride, int32_t pad, int32_t relu, int32_t ldw, lbm_stream_t stream);
int lbm_hip_maxpool3x3s2_nhwc_s(const float* x, float* y, void* ys, int64_t ys_plane, int32_t B, int32_t H, int32_t W,
                                int32_t C, lbm_stream_t stream);
int lbm_hip_upcat_s(const float* p1, const float* p2, const float* p3, float* out, void* outs, int64_t outs_plane, int32_t B,
                    int32_t h, int32_t w, lbm_stream_t stream);
int lbm_hip_pyramid_s(float* tokens, void* tokens_s, int64_t tokens_plane, int32_t B, int32_t h, int32_t w,
                      lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole tracking step = LBMTransformer.forward (transformer.py:74-151) + the state roll and output
 * scaling of LBM_online.online_forward (lbm_online.py:143-156), enqueued as one call.
 * Weight table layout: see the LBM_W_* indices below (built by lbm_amd/engine.py: TrackerWeights).             */
#define LBM_W_ATTN_G 0      /* [1024,256]  per-head Wk_h^T Wq_h / 8, heads stacked           */
#define LBM_W_ATTN_GB 1     /* [1024]                                                         */
#define LBM_W_ATTN_TEMB 2   /* [M,256] time embedding of the past slots (NULL for lattice update) */
#define LBM_W_ATTN_HM 3     /* [256,1024] per-head Wo[:,h] Wv_h, heads concatenated along K   */
#define LBM_W_ATTN_HB 4     /* [256]                                                          */
#define LBM_W_ATTN_LN1G 5
#define LBM_W_ATTN_LN1B 6
#define LBM_W_ATTN_F1W 7    /* [1024,256] */
#define LBM_W_ATTN_F1B 8
#define LBM_W_ATTN_F2W 9    /* [256,1024] */
#define LBM_W_ATTN_F2B 10
#define LBM_W_ATTN_LN2G 11
#define LBM_W_ATTN_LN2B 12
#define LBM_W_ATTN_HMSUM 13  /* [256,256] sum over the heads of the HM blocks: the whole read-out when the key sequence has
                              * ONE element (soft-max over one key is exactly 1, transformer.py:333 at top-k = 1)        */
#define LBM_W_ATTN_F1FM 14   /* f16 fragment-major copies of F1W / F2W (lbm_hip_pack_fragment_major) for the fused feed-forward; */
#define LBM_W_ATTN_F2FM 15   /* optional (NULL: the feed-forward runs as two products)                                        */
#define LBM_W_ATTN_COUNT 16

#define LBM_W_MSDA_OLW 0    /* [48P,256] sampling_offsets (32P rows) then attention_weights (16P rows) */
#define LBM_W_MSDA_OLB 1
#define LBM_W_MSDA_OUTW 2   /* [256,256] output_proj */
#define LBM_W_MSDA_OUTB 3
#define LBM_W_MSDA_LN1G 4
#define LBM_W_MSDA_LN1B 5
#define LBM_W_MSDA_F1W 6
#define LBM_W_MSDA_F1B 7
#define LBM_W_MSDA_F2W 8
#define LBM_W_MSDA_F2B 9
#define LBM_W_MSDA_LN2G 10
#define LBM_W_MSDA_LN2B 11
#define LBM_W_MSDA_F1FM 12   /* as LBM_W_ATTN_F1FM / F2FM */
#define LBM_W_MSDA_F2FM 13
#define LBM_W_MSDA_COUNT 14

/* per layer i, base = i * LBM_W_LAYER_COUNT */
#define LBM_W_LAYER_COLL 0
#define LBM_W_LAYER_STREAM 16
#define LBM_W_LAYER_LATTICE 32
#define LBM_W_LAYER_UPDATE 46
#define LBM_W_LAYER_RHO_W 62  /* [1,256] reference_rho */
#define LBM_W_LAYER_RHO_B 63
#define LBM_W_LAYER_OLG_W 64  /* [48*9 + 1024, 256] LATTICE.OLW stacked on UPDATE.G: both read the stream-attention output, one
                               * product serves the deformable offsets/weights and the update attention's queries      */
#define LBM_W_LAYER_OLG_B 65
#define LBM_W_LAYER_COUNT 66
/* after the layers, base = n_layers * LBM_W_LAYER_COUNT */
#define LBM_W_HEAD_OFFSET 0
#define LBM_W_HEAD_CMEM 14
#define LBM_W_HEAD_VIS 28
#define LBM_W_HEAD_OFFH_W0 42 /* offset_head.net.0 [256,256] */
#define LBM_W_HEAD_OFFH_B0 43
#define LBM_W_HEAD_OFFH_W3 44 /* offset_head.net.3 [2,256] */
#define LBM_W_HEAD_OFFH_B3 45
#define LBM_W_HEAD_VISH_W0 46
#define LBM_W_HEAD_VISH_B0 47
#define LBM_W_HEAD_VISH_W3 48
#define LBM_W_HEAD_VISH_B3 49
#define LBM_W_HEAD_VPROJ_W 50 /* [(n_layers+3)*256, 256] value_proj of lattice_0..n-1, offset, collision_mem, vis */
#define LBM_W_HEAD_VPROJ_B 51
#define LBM_W_HEAD_OL3_W 52   /* [48*9 + 48*4 + 48*9, 256] OFFSET.OLW, CMEM.OLW, VIS.OLW stacked: the three heads' deformable
                               * offsets/weights are functions of the same final token                                 */
#define LBM_W_HEAD_OL3_B 53
#define LBM_W_HEAD_COUNT 54

typedef struct {
  /* geometry */
  int32_t B, N, M, n_layers;
  int32_t topk[LBM_HIP_MAX_LAYERS];
  int32_t net_h, net_w, stride;         /* 384, 512, 4 */
  int32_t n_points, n_points_cmem;      /* 9, 4 */
  int32_t ring_head;                    /* physical slot holding the oldest memory entry */
  int32_t update_state;                 /* 1: overwrite slot ring_head of memories and masks (state roll) */
  float vis_threshold;                  /* 0.8 (lbm.py:17) */
  float img_h, img_w;                   /* caller frame size for the coords output */
  float corr_mix[LBM_HIP_MAX_LAYERS * 4]; /* corr_conv.weight per layer */
  float corr_bias[LBM_HIP_MAX_LAYERS];
  /* weights */
  const float* const* weights;          /* HOST array of DEVICE pointers, LBM_W_* layout */
  int32_t n_weights;
  /* precision 0: every contraction on the exact-fp32 MFMA (weights[] are fp32 matrices).
   * precision 1: contractions on the f16 matrix cores with split operands (fp32-grade, see lbm_hip_split_planes):
   *   for every MATRIX entry of the table, weights[i] is its f16 hi plane and weights_lo[i] its f16 lo plane;
   *   vectors (biases, LayerNorm, time embeddings) and LBM_W_LAYER_RHO_W stay fp32 in weights[].
   * precision 2: the reduced-precision mode ("bf16 storage with fp32 accumulate", bar 1e-2 on logits): the
   *   same kernels multiply the f16 hi planes only (plain f16 operands, 11 significant bits -- 3 more than bf16 --
   *   one MFMA per block, half the operand staging); weights_lo may be NULL.  Everything that is not a dense
   *   contraction stays fp32.
   * precision 3: the f16-STORAGE mode, the reduced-precision mode to prefer (lbm_online.py:133-146 state; main.py:69 is the
   *   reference's own fp16 practice): stream_mem / coll_mem point at F16 elements (same element layout, half the bytes), the six
   *   projected value maps are f16 in the workspace, and every product of the step multiplies the hi planes only -- EXCEPT the
   *   correlation product + top-k (transformer.py:306-313), which stay on both planes, so the arg-max is taken on fp32-grade
   *   scores of the fp32-grade tokens (pass tokens / tokens_split from the fp32-grade encoder).  Bars: 1e-2 logits, 0.05 px.  */
  int32_t precision;
  const void* const* weights_lo;
  /* inputs */
  const float* tokens;                  /* [B, L, 256] pyramid (levels filled) */
  const void* tokens_split;             /* optional (precision 1): split twin of tokens (lbm_hip_pyramid_s), lo plane
                                           B*L*256 elements after hi; NULL = the step derives it itself */
  const float* query;                   /* [B, N, 256] initial point tokens */
  float* stream_mem;                    /* [B, N, M, 256] ring buffers (precision 3: f16 elements behind these pointers) */
  float* coll_mem;
  uint8_t* vis_mask;                    /* [B, N, M] physical slot order */
  uint8_t* mem_mask;
  const uint8_t* queried;               /* [B, N] or NULL (all queried) */
  /* outputs (optional unless noted) */
  float* coords;                        /* [B, N, 2] caller pixels */
  float* vis;                           /* [B, N] logits (required) */
  float* rho;                           /* [B, N] logits (required) */
  float* offset;                        /* [B, N, 2] network pixels (required) */
  float* stream_out;                    /* [B, N, 256] memory['stream'] */
  float* coll_out;                      /* [B, N, 256] memory['collision'] */
  float* ref_points[LBM_HIP_MAX_LAYERS];/* per layer [B, N, k, 2] network pixels */
  int32_t* topk_idx[LBM_HIP_MAX_LAYERS];/* per layer [B, N, k] */
  float* ref_rho[LBM_HIP_MAX_LAYERS];   /* per layer [B, N, k] */
  float* corr[LBM_HIP_MAX_LAYERS];      /* per layer [B, N, h*w] (training-style output; slow path) */
  /* workspace */
  void* workspace;
  uint64_t workspace_bytes;
  /* Optional point-parallel execution (B == 1 only): the per-point chain of up to 1 + n_aux_streams row ranges is
   * enqueued concurrently, range 0 on `stream`, the others on these caller-owned HIP streams; the call forks them
   * after the frame-level kernels and joins them back into `stream` before returning (event record / wait only, no
   * host synchronisation; stream-capture safe).  The library keeps 4 timing-disabled events for this.          */
  lbm_stream_t aux_streams[3];
  int32_t n_aux_streams;
  /* Phased execution, for callers that orchestrate the concurrency themselves (e.g. one HIP graph per phase):
   * phase 0 = whole step; 1 = frame-level kernels only (split tokens, value maps, correlation maps);
   * 2 = the per-point chain of rows [row_lo, row_lo + row_count) only (B == 1 unless the range is everything),
   *     `sub_index` < 4 naming this range's private scratch region; phase 2 calls of disjoint ranges may run
   *     concurrently on different streams once phase 1 has completed.                                          */
  int32_t phase, row_lo, row_count, sub_index;
  /* optional conveniences, each saving the caller (and the GPU) a tiny launch per frame:
   * query_split: split twin of `query` (lbm_hip_split_planes; lo plane B*N*256 elements further) for precision >= 1 --
   *   the query never changes between point-set updates, so it need not be re-split every frame;
   * vis_flag [B,N] bytes = sigmoid(vis) > vis_threshold; rho_prob [B,N] = sigmoid(rho)   (lbm_online.py:148-149)       */
  const void* query_split;
  uint8_t* vis_flag;
  float* rho_prob;
  /* optional: query_u0 [B, N, 1024] fp32, the first layer's collision-attention queries G . query + gb.  The query only
   * changes at point-set updates, so this product need not be repeated every frame: with query_u0_valid == 0 the step
   * computes it INTO query_u0, with query_u0_valid != 0 it reads it from there (the caller keeps the flag; same bits
   * either way).  NULL: computed every frame into the workspace.                                                  */
  float* query_u0;
  int32_t query_u0_valid;
  /* optional: inv_norm [B, L] fp32 = 1 / max(|tokens row|, 1e-12) (F.normalize's factor, transformer.py:207) as written by
   * lbm_hip_encoder_tail_fused for exactly these tokens: the step then skips its own norm pass.  NULL: computed here.        */
  const float* inv_norm;
} lbm_hip_step_desc;

/* Tail of the encoder (backbone.py:53-65, 181-193 and transformer.py:58-69 in one pass): the 1x1 projections, the
 * bilinear up-sampling and the fusion 1x1 + BatchNorm are all linear, so the caller multiplies each feature level by
 * (fusion o projection) at its own resolution (q1 [B,h*w,256] with the folded bias, q2 [B,h*w/4,256], q3 [B,h*w/16,256])
 * and this kernel forms tokens = GELU(q1 + up2(q2) + up4(q3)) + pos for level 0, the three pooled levels, and (tokens_s
 * != NULL) the split twin of the whole buffer.                                                                  */
int lbm_hip_encoder_tail(const float* q1, const float* q2, const float* q3, const float* pos, float* tokens, void* tokens_s,
                         int64_t tokens_plane, int32_t B, int32_t h, int32_t w, lbm_stream_t stream);
/* The same with the finest level's product inside: q1 = f1 . W1^T + b1 is formed per 8 x 8 block of cells on the matrix cores
 * (f1: split planes [B, h*w, 64] of the layer-1 features, W1_hi / W1_lo: [256, ldw1] split weights of (fusion o proj1), b1 [256] or
 * NULL) and never reaches memory.  Bit-identical tokens to lbm_hip_linear_ss + lbm_hip_encoder_tail.  inv_norm (optional) [B, L]:
 * receives 1 / max(|row|, 1e-12) of every token row written, for lbm_hip_step_desc.inv_norm.                                   */
int lbm_hip_encoder_tail_fused(const void* f1, int64_t f1_plane, const void* W1_hi, const void* W1_lo, int32_t ldw1, const float* b1,
                               const float* q2, const float* q3, const float* pos, float* tokens, void* tokens_s, int64_t tokens_plane,
                               float* inv_norm, int32_t B, int32_t h, int32_t w, lbm_stream_t stream);

/* Fragment-major copy of a split weight matrix [N, K] (N % 32 == 0, K % 16 == 0): out[N/32][K/16][2][64][8] f16 -- for every
 * (32-row block, 16-deep k step) the 64 lanes' MFMA operand fragments of the hi plane, then of the lo plane, contiguous, so
 * a wave fetches a fragment with one coalesced 16-byte-per-lane load straight into registers.  Packing-time helper
 * (the reference has no counterpart: its weights stay in nn.Linear layout).                                       */
int lbm_hip_pack_fragment_major(const void* hi, const void* lo, int32_t N, int32_t K, void* out, lbm_stream_t stream);

/* Feed-forward 256 -> 1024 -> 256 of transformer.py:343-355 (Linear, GELU(erf), Linear) in one launch for few rows:
 * slabs[c] = GELU(A . F1[128 c : 128 c + 128]^T + b1[...]) . F2[:, 128 c : 128 c + 128]^T, c = 0..7 -- eight partial [M, 256]
 * outputs (slab_stride floats apart) whose sum, plus the second layer's bias, is the MLP output; the LayerNorm that
 * follows in every block of the step reduces them (lbm_hip_track_step does exactly that).  A: split planes [M, lda];
 * F1_fm / F2_fm: lbm_hip_pack_fragment_major of the two weight matrices; planes 2 (fp32-grade) or 1 (hi planes only). */
int lbm_hip_ffn_slabs(const void* A, int64_t a_plane, int32_t lda, const void* F1_fm, const float* b1, const void* F2_fm, float* slabs,
                      int64_t slab_stride, int32_t M, int32_t planes, lbm_stream_t stream);

/* Add / remove tracked points without a host round trip.  Replaces the state edit of LBM_online.update
 * (lbm_online.py:84-110): the old rows with remove_mask == 0 move, in order, to the front of the new state
 * (query, both memories in their physical ring order -- the caller's ring head is unchanged --, both masks); rows with
 * update_mask != 0 take new_query[row] -- a surviving row re-planted this way keeps its memories, as in the reference
 * (lbm_online.py:93-98), a row past the survivors starts with an empty memory (every slot ignored); rows that are
 * neither are zero / ignored.
 * The kept-row count is computed and consumed on the device (the reference's .item() / nonzero syncs are gone).
 * scratch: N_old + 1 int32.  The new state must not alias the old one.                                          */
int lbm_hip_update_points(const uint8_t* remove_mask, int32_t N_old, const uint8_t* update_mask, int32_t N_new, int32_t M,
                          const float* old_query, const float* old_stream, const float* old_coll, const uint8_t* old_vis,
                          const uint8_t* old_mem, const float* new_query, float* query, float* stream_mem, float* coll_mem,
                          uint8_t* vis_mask, uint8_t* mem_mask, int32_t* scratch, lbm_stream_t stream);

/* lbm_hip_update_points for callers that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference.  Same arguments and results, minus `scratch`: the scan's N_old + 1
 * ints live in a static device array of the library (LBM_UPDATE_STATIC_MAX_POINTS + 1 ints = 256 KiB, held by every GPU the
 * library runs on until the process ends; its address is looked up once per device, on the first call there, with
 * hipGetSymbolAddress).  The call allocates nothing and frees nothing.  N_old above the maximum is refused (use
 * lbm_hip_update_points).  Nothing in the array needs clearing between calls (the scan writes every entry the gather reads),
 * but every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two host threads) race.  */
#define LBM_UPDATE_STATIC_MAX_POINTS 65536
int lbm_hip_update_points_static(const uint8_t* remove_mask, int32_t N_old, const uint8_t* update_mask, int32_t N_new, int32_t M,
                                 const float* old_query, const float* old_stream, const float* old_coll, const uint8_t* old_vis,
                                 const uint8_t* old_mem, const float* new_query, float* query, float* stream_mem, float* coll_mem,
                                 uint8_t* vis_mask, uint8_t* mem_mask, lbm_stream_t stream);

/* bytes of workspace lbm_hip_track_step needs for this geometry (only B, N, M, n_layers, topk, net_* are read) */
uint64_t lbm_hip_step_workspace_bytes(const lbm_hip_step_desc* d);
int lbm_hip_track_step(const lbm_hip_step_desc* d, lbm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference, which only has wall-clock timing, benchmark_torch.py:71-76):
 * between _begin and _end every kernel launch of this library is bracketed by HIP events on its launch stream;
 * _end synchronises on them and returns one aggregated entry per kernel (launch count, summed event time,
 * summed algorithmic flops / bytes as defined in DESIGN.md).  Returns the number of distinct kernels (entries
 * beyond `cap` are dropped) or a negative value.  Not thread-safe; adds ~2 events per launch while enabled.  */
typedef struct {
  char name[96];
  int64_t launches;
  double total_ms, flops, bytes;
} lbm_hip_profile_entry;
int lbm_hip_profile_begin(void);
int lbm_hip_profile_end(lbm_hip_profile_entry* out, int32_t cap);

/* tuning / debugging knobs (process-wide; 0 restores the default unless noted).  None changes results beyond the last bit except
 * the reproducer variants of key 9 and the timing-only ablation switches (keys 14 / 16 / 18 bits 0-7 / 20 / 23, bits 4.. of key 24, bits 0-7 of key 26 and bits 12.. of key 2:
 * results are WRONG when set).  The ablation code exists only in the tuning build (`make -C lbm_amd/csrc tuning`, -DLBM_TUNING ->
 * liblbm_hip_tuning.so, see lbm_hip_tuning_build): the product library refuses those values with an error.
 *   1  force the split-GEMM tiling (1..23, gemm16.hip)          2  split-GEMM flags (256 = never / 512 = always the row-streaming
 *   3  attention read-out mode                                      kernel, 32 = no XCD remap, 1024 = the rows-stationary form of it (slower; kept for
 *                                                                   the A/B), bits 12.. = its ablation switches)
 *   4  force the convolution tiling (1..14)                     5  dense top-k mode
 *   6  largest k fused into the tiled correlation product below 4096 rows (default 1)
 *   7  most rows for the few-rows fused feed-forward (default 2048)   8  feed-forward sites that may fuse (bit mask, default 63)
 *   9  correlation-map kernel variant (0 product, 6 cell-by-cell, 1..4 packed-fp32 reproducers)
 *  10  corr_maps uses the caller's inv_norm as it is             11  serial-load LayerNorm      12  fused feed-forward claims the register file
 *  13  correlation + top-k by the row-streaming kernel (default 1)   14  its ablation bits
 *  15  many-rows fused feed-forward: 0 auto (above key 7's row count), 1 never, 2 always        16  its ablation bits
 *  17  offset and visibility heads as one batch of two (default 1)
 *  18  conv64_rows timing-only ablation bits (results are wrong when set): 1 no DMA after the prologue, 2 no epilogue,
 *      4 no first-half MFMAs, 8 no second-half MFMAs (default 0); convolution force mode (key 4) 15 = never, 16 = always
 *  20  stem_pool timing-only ablation bits (results are wrong when set): 1 no DMA after the prologue, 2 no pooling / stores,
 *      4 no MFMAs, 8 no hand-over writes (default 0)
 *  21  msda_sample with a batch of streams: 1 (default) workgroups dealt so that each XCD works through one stream's value map
 *      at a time, 0 dispatch order
 *  22  deformable-sampling kernels (bit mask, default 3): 1 = LDS-window kernel for the k = 9 / 4 lattice launches, 2 = tap-table
 *      kernel for the k = 1 launches; 0 = the direct gather everywhere (bit-identical rows, the A/B of DESIGN.md section 5)
 *  23  msda_window timing-only ablation bits (tuning build): 1 no tap table, 2 no staging, 4 no accumulation, 8 no prologue reductions
 *  24  encoder tail (bit mask, default 1): bit 0 = workgroup order, the same 8 x 8 block of every stream back to back (0: stream by
 *      stream); bit 1 = cell-by-cell bilinear taps instead of one patch per 2 x 2 group (bit-identical tokens, the A/B of DESIGN.md
 *      section 8); bits 4.. timing-only ablation (tuning build): 16 no level-0 product, 32 no tap / pos loads, 64 no stores, 128 no GELU
 *  25  gemm_rs LDS ring: 3 (default) or 4 stages (steps fetched two / three ahead); bit-identical, measured equal
 *  26  conv_patch (3 x 3 convolutions 128 -> 128 / 256 -> 256 of many frames; key 4: 21 = never, 22 = whenever the shape admits it):
 *      bits 8.. = pixels per tile (192 / 256, 0 = chosen by the launcher), bits 0-7 timing-only ablation (tuning build): 1 no patch DMA
 *      after the first, 2 no weight DMA after the prologue, 4 no MFMAs, 8 no epilogue, 16 no barrier in the main loop, 32 no DMA waits                                                 */
int lbm_hip_debug_set(int32_t key, int32_t value);
/* The knobs that currently differ from their defaults (what a measurement must state beside its figure; bench.py prints them
 * in `config.knobs` and refuses a headline when any is set): writes up to `cap` (key, value) pairs, returns how many there are. */
int32_t lbm_hip_debug_active(int32_t* keys, int32_t* values, int32_t cap);
/* 1 when the library was compiled with -DLBM_TUNING (timing-only ablation code present), 0 for the product build */
int32_t lbm_hip_tuning_build(void);
/* the same for pointer-valued knobs: key 1 = device buffer (768 floats per 2x2 cell block) that correlation-map variant 2
 * (debug key 9 = 2) fills with what each wave loaded and computed -- reproduction aid; NULL = off */
int lbm_hip_debug_ptr(int32_t key, void* p);

/* debugging aid for parity tests: byte offset of a named intermediate inside the workspace, or -1 */
int64_t lbm_hip_step_workspace_offset(const lbm_hip_step_desc* d, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* LBM_HIP_H */
