// Dense building block: Y = act(X · Wᵀ + bias) + res_scale · RES, exact fp32 on the matrix cores.
#pragma once
#include "common.hpp"

namespace lbm {

enum Act { ACT_NONE = 0, ACT_GELU = 1 };

struct GemmArgs {
  const float* X = nullptr;   // [M, K] row-major, leading dim ldx
  const float* W = nullptr;   // [N, K] row-major (nn.Linear layout), leading dim ldw
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

}  // namespace lbm
