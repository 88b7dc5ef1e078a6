#pragma once
#include "common.hpp"

namespace lbm {

int offset_final(const float* hidden, const float* w3, const float* b3, const float* ref, float stride_px, float* offset,
                 float* pred, int R, hipStream_t stream);

struct StateFinalArgs {
  const float* vis_hidden = nullptr;  // [R,256] GELU(vis_head.net.0(token))
  const float* w3 = nullptr;          // [2,256] vis_head.net.3.weight
  const float* b3 = nullptr;          // [2]
  const float* stream_tok = nullptr;  // [R,256] token after the layers  (memory['stream'])
  const float* coll_tok = nullptr;    // [R,256] collision-memory head output
  const float* query_init = nullptr;  // [R,256]
  const uint8_t* queried = nullptr;   // [R] or null (= all true)
  float* stream_out = nullptr;        // optional [R,256] copies (stateless API)
  float* coll_out = nullptr;
  float* stream_mem = nullptr;        // optional ring buffers [R, M, 256]; slot `ring_head` is overwritten
  float* coll_mem = nullptr;
  uint8_t* vis_mask = nullptr;        // optional [R, M]
  uint8_t* mem_mask = nullptr;
  int M = 0, ring_head = 0;
  float vis_threshold = 0.8f;
  float* vis = nullptr;               // [R] logits
  float* rho = nullptr;               // [R]
  const float* ref = nullptr;         // [R,2] last-layer cell centre (network px)
  const float* offset = nullptr;      // [R,2]
  float* coords = nullptr;            // optional [R,2] in caller pixels
  float net_w = 0, net_h = 0, img_w = 0, img_h = 0;
  int R = 0;
};

int state_final(const StateFinalArgs& a, hipStream_t stream);

}  // namespace lbm
