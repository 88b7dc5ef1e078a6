// Single-query attention read-out over a short per-point sequence (memory slots or lattice tokens).
#pragma once
#include "common.hpp"

namespace lbm {

struct AttnArgs {
  const float* U = nullptr;        // [R, 4, 256]  per-head key-space queries u_h = Wk_hᵀ(Wq_h x + bq_h) / 8
  long long ldu = 4 * LBM_C;       // row pitch of U in floats (wider when U is a column block of a merged product)
  const float* seq = nullptr;      // per-point sequence, element (r, j, c) at seq + r*point_stride + j*slot_stride + c
  long long point_stride = 0;
  long long slot_stride = LBM_C;
  int S = 0;                       // sequence length (memory size M, or k lattice tokens)
  int ring_head = 0;               // physical slot of logical position 0 (ring buffer); 0 for plain sequences
  const float* temb = nullptr;     // [S, 256] time embedding added to the *keys* only, logical order; may be null
  const uint8_t* mask_a = nullptr; // [R, S] physical order, non-zero = ignore (vis_mask); may be null
  const uint8_t* mask_b = nullptr; // [R, S] (mem_mask); may be null
  float* out = nullptr;            // [R, 4, 256]  softmax-weighted sums of the raw sequence rows, per head
  void* out_s = nullptr;           // the same as split f16 planes (hi at out_s, lo out_plane elements further); either may be null
  long long out_plane = 0;
  uint8_t* valid = nullptr;        // [R] 1 if at least one position was attended; may be null
  int R = 0;
  int seq_f16 = 0;                 // 1: `seq` points at f16 elements (same element strides): the f16-storage mode's memories
};

int attn_readout(const AttnArgs& a, hipStream_t stream);
void attn_force_mode(int m);

}  // namespace lbm
