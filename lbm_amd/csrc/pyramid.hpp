#pragma once
#include "common.hpp"

namespace lbm {

// tokens [B, L, 256] with level 0 in rows [0, h*w); fills rows [h*w, L) (levels 1..3) in place.
// toks (optional): split-plane twin of the whole buffer (all four levels), lo plane toks_plane elements after hi.
int pyramid_pool(float* tokens, int B, int h, int w, hipStream_t stream, void* toks = nullptr, long long toks_plane = 0);

// inv_norm [B, L] scratch; F [n_layers, B, h*w, 256]; wmix_host = n_layers x 4 corr_conv weights (host memory).
// split: F is written as the B operand of the split-precision GEMM (per layer: hi planes of all streams, then lo planes)
int corr_maps(const float* tokens, float* inv_norm, float* F, const float* wmix_host, int n_layers, int B, int h, int w,
              bool split, hipStream_t stream, bool have_norm = false);

// per row: indices of the k largest values (ties -> lowest index), optional cell centres (x, y) and values.
int topk_rows(const float* corr, long long ld, int P, int k, int rows, int grid_w, int stride_px, int* idx_out,
              float* ref_out, float* val_out, hipStream_t stream);

int encoder_tail(const float* q1, const float* q2, const float* q3, const float* pos, float* tokens, int B, int h, int w,
                 hipStream_t stream, void* toks = nullptr, long long toks_plane = 0);
int encoder_tail_fused(const void* f1, long long f1_plane, const void* w1h, const void* w1l, int ldw1, const float* b1, const float* q2,
                       const float* q3, const float* pos, float* tokens, int B, int h, int w, hipStream_t stream, void* toks = nullptr,
                       long long toks_plane = 0, float* inv_norm = nullptr);
int topk_candidates(const float* cand_v, const int* cand_i, int n_cand, int P, int k, int rows, int grid_w, int stride_px, int* idx_out,
                    float* ref_out, float* val_out, hipStream_t stream);
void topk_force_mode(int m);   // tuning: 1 = the per-lane-list kernels for every shape

}  // namespace lbm
