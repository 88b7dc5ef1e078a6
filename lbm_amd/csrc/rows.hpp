// Row-wise (per-token, C = 256) kernels: LayerNorm, L2 normalisation, select.
#pragma once
#include "common.hpp"

namespace lbm {

// Y[r] = keep[r] ? LN(X[r]) * gamma + beta : fallback[r];  Yn[r] = Y[r] / max(|Y[r]|, 1e-12) (optional).
// X and Y may alias.  eps = 1e-5 (nn.LayerNorm default, transformer.py:31).
int layernorm_rows(const float* X, const float* gamma, const float* beta, float* Y, int rows,
                   const uint8_t* keep, const float* fallback, float* Yn, hipStream_t stream);

// Same with any subset of outputs: fp32 Y / Yn and their split-plane twins (operands of the split-precision GEMM).
struct RowsOut {
  float* Y = nullptr;
  void* Ys = nullptr;      // split planes of Y  (hi at Ys, lo ys_plane elements further)
  long long ys_plane = 0;
  float* Yn = nullptr;     // L2-normalised rows
  void* Yns = nullptr;
  long long yns_plane = 0;
};
int layernorm_rows_ex(const float* X, const float* gamma, const float* beta, int rows, const uint8_t* keep,
                      const float* fallback, const RowsOut& out, hipStream_t stream);

// LayerNorm whose input is assembled on the fly: x[r] = sum_{s < nslab} X[s * slab_stride + r] + bias + res_scale * res[r / res_div]
// (the K-split partial products of a GEMM, its bias and its residual: the reduction the GEMM epilogue would do).
struct RowsIn {
  const float* X = nullptr;
  int nslab = 1;
  long long slab_stride = 0;     // elements
  const float* bias = nullptr;   // [256] or null
  const float* res = nullptr;    // [rows / res_div, 256] or null
  int res_div = 1;
  float res_scale = 1.f;
  // two parameter sets in one launch: rows [set_rows, ...) take bias2 / gamma2 / beta2 / res2
  int set_rows = 0;              // 0: one set
  const float* bias2 = nullptr;
  const float* res2 = nullptr;   // residual of the second set, indexed by (row - set_rows) / res_div
  const float* gamma2 = nullptr;
  const float* beta2 = nullptr;
};
int layernorm_rows_in(const RowsIn& in, const float* gamma, const float* beta, int rows, const uint8_t* keep,
                      const float* fallback, const RowsOut& out, hipStream_t stream);

// Y[r] = X[r] / max(|X[r]|_2, 1e-12)   (F.normalize, transformer.py:207)
int l2norm_rows(const float* X, long long ldx, float* Y, long long ldy, long long rows, hipStream_t stream);

}  // namespace lbm
