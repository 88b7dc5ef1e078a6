#pragma once
#include "common.hpp"

namespace lbm {

int preprocess(const float* frames, long long frame_bstride, float* out, int B, int H, int W, int oh, int ow, int out_nhwc,
               hipStream_t stream);
int preprocess_u8(const uint8_t* frames, long long frame_bstride, int in_layout, float* out, int B, int H, int W, int oh, int ow,
                  int out_nhwc, hipStream_t stream);
int preprocess_split(const void* frames, int is_u8, long long frame_bstride, int in_layout, void* outs, long long outs_plane, int B, int H,
                     int W, int oh, int ow, hipStream_t stream);
int upcat(const float* p1, const float* p2, const float* p3, float* out, int B, int h, int w, hipStream_t stream,
          void* outs = nullptr, long long outs_plane = 0);
int sample_queries(const float* tokens, long long tok_bstride, int ld, const float* pts, float* out, int B, int N, int h, int w,
                   int net_h, int net_w, hipStream_t stream);

}  // namespace lbm
