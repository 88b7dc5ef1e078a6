// Row-wise (per-token, C = 256) kernels: LayerNorm, L2 normalisation, select.
#pragma once
#include "common.hpp"

namespace lbm {

// Y[r] = keep[r] ? LN(X[r]) * gamma + beta : fallback[r];  Yn[r] = Y[r] / max(|Y[r]|, 1e-12) (optional).
// X and Y may alias.  eps = 1e-5 (nn.LayerNorm default, transformer.py:31).
int layernorm_rows(const float* X, const float* gamma, const float* beta, float* Y, int rows,
                   const uint8_t* keep, const float* fallback, float* Yn, hipStream_t stream);

// Y[r] = X[r] / max(|X[r]|_2, 1e-12)   (F.normalize, transformer.py:207)
int l2norm_rows(const float* X, long long ldx, float* Y, long long ldy, long long rows, hipStream_t stream);

}  // namespace lbm
