// Dense building block: Y = act(X · Wᵀ + bias) + res_scale · RES, exact fp32 on the matrix cores.
#pragma once
#include "common.hpp"

namespace lbm {

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU_POST = 2 };  // RELU_POST: relu(acc + bias + res)

struct GemmArgs {
  const float* X = nullptr;   // [M, K] row-major, leading dim ldx
  const float* W = nullptr;   // [N, K] row-major (nn.Linear layout), leading dim ldw
  // split-precision path (gemm16): W pre-split into two f16 planes [N, ldw] (hi, lo * 2^11); W above is ignored
  const void* Wh = nullptr;
  const void* Wl = nullptr;
  const float* bias = nullptr;  // [N] or null
  float* Y = nullptr;         // [M, N], leading dim ldy
  const float* res = nullptr;   // optional residual, row index (m / res_div) % res_mod, leading dim ldres
  int M = 0, N = 0, K = 0;
  int ldx = 0, ldw = 0, ldy = 0, ldres = 0;
  int res_div = 1, res_mod = 0x7fffffff;
  float res_scale = 1.0f;
  int act = ACT_NONE;
  int batch = 1;              // gridDim.z; element strides below
  long long sX = 0, sW = 0, sB = 0, sY = 0, sR = 0;
  // optional LayerNorm fused behind the residual add (requires N == 256):
  //   Y = LN(act(XWᵀ+b) + res) · gamma + beta ; rows with keep[m]==0 are replaced by fallback[m,:]
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  const uint8_t* keep = nullptr;     // [M] or null
  const float* fallback = nullptr;   // [M, 256], leading dim 256
  float* Yn = nullptr;               // optional second output: L2-normalised rows of Y (ld 256)
};

int gemm(const GemmArgs& a, hipStream_t stream);

// NHWC convolution as an implicit GEMM: Y[b,oy,ox,:] = relu?( sum W[:, (ky,kx,ci)] * X[b, oy*s-p+ky, ox*s-p+kx, ci] + bias (+ res) )
struct ConvArgs {
  const float* X = nullptr;     // [B, H, Wd, Cin], Cin a power of two >= 4
  const float* W = nullptr;     // [Cout, ldw], row = (ky, kx, ci) tap-major, zero-padded to a multiple of 32 columns
  const void* Wh = nullptr;     // split-precision path: the same matrix as two f16 planes
  const void* Wl = nullptr;
  const float* bias = nullptr;  // [Cout] (folded BatchNorm shift) or null
  const float* res = nullptr;   // optional [B, Ho, Wo, Cout] residual added before the ReLU
  float* Y = nullptr;           // [B, Ho, Wo, Cout]
  int B = 0, H = 0, Wd = 0, Cin = 0, Cout = 0, kh = 0, kw = 0, stride = 1, pad = 0, relu = 0, ldw = 0;
};
int conv2d_nhwc(const ConvArgs& c, hipStream_t stream);
// split-precision variants (gemm16.hip): fp32-grade results from three f16 MFMAs per product block
int gemm16(const GemmArgs& a, hipStream_t stream);
int conv2d_nhwc16(const ConvArgs& c, hipStream_t stream);
int split_planes(const float* x, void* hi, void* lo, long long n, hipStream_t stream);
void gemm16_force_tile(int t);
void gemm16_debug(int flags);
void conv16_force(int mode);

// Both operands pre-split into f16 planes: Y = act(A · Bᵀ + bias) + res_scale · RES, fp32 and/or split output.
struct GemmSSArgs {
  const void* A = nullptr;      // f16 hi plane [M, lda]; lo plane a_plane elements further
  const void* B = nullptr;      // f16 hi plane [N, ldb]; lo plane b_plane elements further
  long long a_plane = 0, b_plane = 0;
  const float* bias = nullptr;
  float* Y = nullptr;           // optional fp32 output [M, ldy]
  void* Ys = nullptr;           // optional split output (hi plane [M, ldys], lo plane ys_plane elements further)
  long long ys_plane = 0;
  const float* res = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldy = 0, ldys = 0, ldres = 0;
  int res_div = 1, res_mod = 0x7fffffff;
  float res_scale = 1.0f;
  int act = ACT_NONE;
  int batch = 1;
  long long sA = 0, sB = 0, sBias = 0, sY = 0, sYs = 0, sR = 0;
  // Optional fused row-wise top-k pre-selection (the correlation product): instead of Y, every 128-column tile leaves the
  // topk_k best (value, column) pairs of each of its rows in cand_v / cand_i [batch * M, ceil(N / 128) * topk_k], ties to
  // the lower column; the union holds the exact top-k of the row.  Needs the 128 x 128 tiling (gemm_ss_topk_ok).
  float* cand_v = nullptr;
  int* cand_i = nullptr;
  int topk_k = 0;
  int planes = 2;               // 2: hi and lo planes (fp32-grade); 1: hi planes only (plain f16 operands, reduced precision)
};
int gemm_ss(const GemmSSArgs& a, hipStream_t stream);
bool gemm_ss_topk_ok(int M, int N, int K, int batch, int k);   // would gemm_ss take the fused top-k path for this shape?
// Row-streaming variant for K = 256 (gemm_rs.hip): B stationary in registers, A streamed through LDS; same arithmetic, bit-identical.
bool gemm_rs_ok(const GemmSSArgs& a);
int gemm_rs(const GemmSSArgs& a, int dbg, hipStream_t stream);
// the same product with the ROWS stationary (activations in registers, weights streamed): many rows x many columns, fp32 output
bool gemm_rs_rows_ok(const GemmSSArgs& a);
int gemm_rs_rows(const GemmSSArgs& a, int dbg, hipStream_t stream);
// ... and with the roles swapped for the correlation product + per-point top-k pre-selection: candidates per point
// (gemm_rs_topk_cands tells how many, a multiple of 4) in cand_v / cand_i [batch * M, n_cand]
int gemm_rs_topk_cands(int P0, int N, int batch, int k);
int gemm_rs_topk(const GemmSSArgs& a, int dbg, hipStream_t stream);

// NHWC convolution with split operands staged by LDS-DMA gather (Cin a power of two >= 32).
struct ConvSSArgs {
  const void* X = nullptr;        // activation, f16 hi plane [B, H, Wd, Cin]; lo plane x_plane elements further
  long long x_plane = 0;
  const void* Wh = nullptr;       // packed weights [Cout, ldw], columns (ky, kx, ci): hi and lo planes
  const void* Wl = nullptr;
  const float* bias = nullptr;
  const float* res = nullptr;     // optional fp32 residual [B, Ho, Wo, Cout] added before the ReLU
  float* Y = nullptr;             // optional fp32 output
  void* Ys = nullptr;             // optional split output (lo plane ys_plane elements further)
  long long ys_plane = 0;
  const void* zeros = nullptr;    // >= 16 zero bytes in device memory (what out-of-image taps read)
  void* scratch = nullptr;        // optional fp32 scratch for K-split slabs
  unsigned long long scratch_bytes = 0;
  int B = 0, H = 0, Wd = 0, Cin = 0, Cout = 0, kh = 0, kw = 0, stride = 1, pad = 0, relu = 0, ldw = 0;
  int planes = 2;                 // as GemmSSArgs::planes
};
int conv2d_nhwc_ss(const ConvSSArgs& c, hipStream_t stream);
// 3 x 3 / stride 1 / 64 -> 64 channels over many pixels: weights stationary, image patches streamed (conv_rs.hip); bit-identical
bool conv64_rows_ok(const ConvSSArgs& c);
int conv64_rows(const ConvSSArgs& c, hipStream_t stream);
// 3 x 3 / stride 1 / 128 -> 128 or 256 -> 256 channels on 64- / 32-pixel-wide maps, many pixels: the input patch of a 256-pixel tile stays
// in LDS for all nine taps, only the weights stream (conv_patch.hip); same products, summed channel chunk by channel chunk
bool conv_patch_ok(const ConvSSArgs& c);
int conv_patch(const ConvSSArgs& c, hipStream_t stream);
// stem_rs.hip: 7 x 7 stride-2 convolution (3 + 1 zero channels -> 64) + ReLU + 3 x 3 stride-2 max-pooling in one launch
bool stem_pool_ok(int H, int W, int ldw);
int stem_pool(const void* X, long long x_plane, const void* Wh, const void* Wl, int ldw, const float* bias, float* Y, void* Ys,
              long long ys_plane, int B, int H, int W, hipStream_t stream);
int maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, hipStream_t stream, void* ys = nullptr,
                      long long ys_plane = 0);

// Fused feed-forward 256 -> 1024 -> 256 over fragment-major weights (ffn.hip): partial outputs, one slab per hidden chunk.
struct FfnArgs {
  const void* A = nullptr;        // [M, lda] split planes of the input rows (hi; lo a_plane elements further)
  long long a_plane = 0;
  int lda = 0;
  const void* F1 = nullptr;       // pack_fragment_major of the [1024, 256] first layer
  const float* b1 = nullptr;      // [1024]
  const void* F2 = nullptr;       // pack_fragment_major of the [256, 1024] second layer
  float* slabs = nullptr;         // [1024 / chunk][M, 256] partial outputs (no bias)
  long long slab_stride = 0;
  int M = 0, chunk = 128, planes = 2;
  // two weight sets in one launch (few-rows kernel): rows [0, set_rows) use F1 / b1 / F2, rows [set_rows, M) the second set
  int set_rows = 0;               // 0: one set
  const void* F1b = nullptr;
  const float* b1b = nullptr;
  const void* F2b = nullptr;
};
int ffn_fused(const FfnArgs& f, hipStream_t stream);
int ffn_rows(const FfnArgs& f, hipStream_t stream);     // the same slabs for many rows: weights stationary, rows streamed (ffn.hip)
int pack_fragment_major(const void* hi, const void* lo, int N, int K, void* out, hipStream_t stream);

}  // namespace lbm
