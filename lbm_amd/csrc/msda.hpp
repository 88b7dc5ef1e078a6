// Multi-scale deformable attention sampling core (msdeformattn.py:150-177, 188-235).
#pragma once
#include "common.hpp"

namespace lbm {

struct MsdaArgs {
  // projected value maps: element (b, token, head, c) at value + b*v_batch_stride + token*ldv + head*64 + c
  const float* value = nullptr;
  long long v_batch_stride = 0;
  int ldv = LBM_C;
  int lvl_h[LBM_LEVELS] = {0, 0, 0, 0};
  int lvl_w[LBM_LEVELS] = {0, 0, 0, 0};
  int lvl_start[LBM_LEVELS] = {0, 0, 0, 0};
  int B = 0, N = 0;      // streams, points per stream (rows R = B*N)
  int k = 1;             // reference points per point (queries Q = R*k share offsets and weights)
  int P = 0;             // sampling points per head and level (<= 16)
  // --- implicit mode (the LBM path): offsets + logits straight from the two Linears ---
  const float* offlog = nullptr;  // [R, ld_offlog]: cols [0, 32P) sampling offsets (head, level, point, xy),
  int ld_offlog = 0;              //                 cols [logit_col, logit_col + 16P) attention logits (head, level*P+point)
  int logit_col = 0;
  const float* ref = nullptr;     // [R, k, 2] reference points in network pixels (x, y)
  float net_w = 0.f, net_h = 0.f; // network input size (512, 384): ref is divided by it and clamped to [0,1]
  // --- explicit mode (generic operator): final sampling locations and soft-maxed weights ---
  const float* loc = nullptr;     // [R, 4, 4, P, 2] in [0,1]
  const float* attw = nullptr;    // [R, 4, 4, P]
  float* out = nullptr;           // [R*k, 256]
  void* out_s = nullptr;          // the same as split f16 planes (lo plane out_plane elements after hi); either may be null
  long long out_plane = 0;
  // --- two modules in one launch (implicit mode): rows [0, set_rows) use the pointers above, rows [set_rows, 2 set_rows) the SAME
  // points and reference cells with another value map and another column block of offlog (element offsets below) ---
  int set_rows = 0;               // 0: one module
  long long set_value_off = 0, set_offlog_off = 0;
  int value_f16 = 0;              // 1: `value` points at f16 elements (same element strides): the f16-storage mode's value maps
  int dbg = 0;                    // tuning build only (debug key 23): 1 no tap table, 2 no staging, 4 no accumulation, 8 no prologue reductions
  int xcd_streams = 0;            // set by msda_sample: workgroups dealt so that each XCD works through one stream's map at a time
};

int msda_sample(const MsdaArgs& a, hipStream_t stream);

}  // namespace lbm
