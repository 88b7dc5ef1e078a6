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
  uint8_t* vis_flag = nullptr;        // optional [R]: sigmoid(vis) > vis_threshold
  float* rho_prob = nullptr;          // optional [R]: sigmoid(rho)
  float net_w = 0, net_h = 0, img_w = 0, img_h = 0;
  int R = 0;
  int mem_f16 = 0;                    // 1: the ring buffers hold f16 elements (f16-storage mode)
};

int state_final(const StateFinalArgs& a, hipStream_t stream);

// Add / remove tracked points on the device (lbm_online.py:48-112): rows of the old state that are kept move, in order,
// to the front of the new state; rows flagged in update_mask take the freshly sampled query and an empty (all-ignored)
// memory; any other row is zero.  No host round trip: the number of kept rows is computed and consumed on the device.
struct UpdatePointsArgs {
  const uint8_t* remove_mask = nullptr;   // [N_old] non-zero = drop the point
  const uint8_t* update_mask = nullptr;   // [N_new] non-zero = (re)initialise from new_query
  int N_old = 0, N_new = 0, M = 0;
  const float* old_query = nullptr;       // [N_old, 256]
  const float* old_stream = nullptr;      // [N_old, M, 256] (physical ring order; the ring head does not change)
  const float* old_coll = nullptr;
  const uint8_t* old_vis = nullptr;       // [N_old, M]
  const uint8_t* old_mem = nullptr;
  const float* new_query = nullptr;       // [N_new, 256] samples at the new query points (rows outside update_mask unused)
  float* query = nullptr;                 // outputs, [N_new, ...]
  float* stream = nullptr;
  float* coll = nullptr;
  uint8_t* vis = nullptr;
  uint8_t* mem = nullptr;
  int* scratch = nullptr;                 // N_old + 1 ints
};
int update_points(const UpdatePointsArgs& a, hipStream_t stream);
// the same with the scan's scratch (N_old + 1 ints) in a static device array of the library: allocates nothing
constexpr int LBM_MAX_DEVICES = 64;
int update_points_static(UpdatePointsArgs a, hipStream_t stream);

}  // namespace lbm
