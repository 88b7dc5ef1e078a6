// Feature-encoder side kernels: frame pre-processing, multi-scale concat for the fusion conv, query sampling.
#include "encoder.hpp"

namespace lbm {

// F.interpolate(..., mode='bilinear', align_corners=False) source coefficients for arbitrary in/out sizes.
__device__ __forceinline__ void resize_coeff(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// backbone.py:181-183: x/255 -> bilinear resize (no antialias) -> (x - mean)/std.   NCHW or NHWC out.
// T = float: frames as the reference hands them over, (B,3,H,W) fp32 in [0,255] (tools/demo_grid.py:40).
// T = uint8_t: the decoder's own bytes (demo_utils.py:127-131 converts them to float on the host, 4x the bytes over PCIe and
//   out of HBM): LAYOUT 0 planar (B,3,H,W) RGB; 1 interleaved (B,H,W,3) RGB; 2 interleaved (B,H,W,3) BGR (cv2's order).
//   uint8 -> float is exact, so the result is bit-identical to the fp32 path on the same values.
// Bilinear blend + normalisation of one value with every rounding spelled out: the two preprocess kernels below must give the same
// bits, and left to itself the compiler contracts the blend into fused multiply-adds differently in each.
__device__ __forceinline__ float resize_norm(float v00, float v01, float v10, float v11, float lx0, float lx1, float ly0, float ly1,
                                             float mean, float sd) {
  const float top = __fmaf_rn(lx0, v00, __fmul_rn(lx1, v01));
  const float bot = __fmaf_rn(lx0, v10, __fmul_rn(lx1, v11));
  const float v = __fmaf_rn(ly0, top, __fmul_rn(ly1, bot));
  return __fdiv_rn(__fsub_rn(v, mean), sd);
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int H,
                                                         int W, int oh, int ow, long long in_bstride, int out_nhwc) {
  const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
  const long long total = (long long)B * 3 * oh * ow;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % ow);
    const int y = (int)((i / ow) % oh);
    const int c = (int)((i / ((long long)ow * oh)) % 3);
    const int b = (int)(i / ((long long)ow * oh * 3));
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    resize_coeff(y, sy, H, y0, y1, ly0, ly1);
    resize_coeff(x, sx, W, x0, x1, lx0, lx1);
    const T* p = in + (long long)b * in_bstride;
    auto px = [&](int yy, int xx) {
      if constexpr (LAYOUT == 0) return (float)p[(long long)c * H * W + (long long)yy * W + xx];
      else return (float)p[((long long)yy * W + xx) * 3 + (LAYOUT == 2 ? 2 - c : c)];
    };
    const float v00 = px(y0, x0) / 255.0f, v01 = px(y0, x1) / 255.0f;
    const float v10 = px(y1, x0) / 255.0f, v11 = px(y1, x1) / 255.0f;
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    // out_nhwc: 0 = [B,3,oh,ow]; 3 or 4 = pixel-major with that many channels (channel 3 zero-filled)
    const long long o = out_nhwc ? (((long long)b * oh + y) * ow + x) * out_nhwc + c : i;
    out[o] = resize_norm(v00, v01, v10, v11, lx0, lx1, ly0, ly1, mean, sd);
    if (out_nhwc == 4 && c == 0) out[o + 3] = 0.f;
  }
}

// The 4-channel pixel-major form the encoder takes: one thread per output pixel (the per-element form above writes the three
// channels of a pixel from three different waves, 4 bytes at a 16-byte stride: 1.9 TB/s at 64 frames).  Same arithmetic per value.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void preprocess_px_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int H, int W, int oh,
                                                            int ow, long long in_bstride, void* outs = nullptr, long long outs_plane = 0) {
  const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= ow || y >= oh) return;
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  resize_coeff(y, sy, H, y0, y1, ly0, ly1);
  resize_coeff(x, sx, W, x0, x1, lx0, lx1);
  const T* p = in + (long long)b * in_bstride;
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto px = [&](int yy, int xx) {
      if constexpr (LAYOUT == 0) return (float)p[(long long)c * H * W + (long long)yy * W + xx];
      else return (float)p[((long long)yy * W + xx) * 3 + (LAYOUT == 2 ? 2 - c : c)];
    };
    const float v00 = px(y0, x0) / 255.0f, v01 = px(y0, x1) / 255.0f;
    const float v10 = px(y1, x0) / 255.0f, v11 = px(y1, x1) / 255.0f;
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    r[c] = resize_norm(v00, v01, v10, v11, lx0, lx1, ly0, ly1, mean, sd);
  }
  const long long o = (((long long)b * oh + y) * ow + x) * 4;
  if (out) st4(out + o, make_float4(r[0], r[1], r[2], 0.f));
  if (outs) st4_split(outs, outs_plane, o, make_float4(r[0], r[1], r[2], 0.f));      // what stem_pool reads (stem_rs.hip)
}

// frames -> split planes [2][B, oh, ow, 4] directly (is_u8: in_layout as preprocess_u8, else planar fp32)
int preprocess_split(const void* frames, int is_u8, long long frame_bstride, int in_layout, void* outs, long long outs_plane, int B, int H,
                     int W, int oh, int ow, hipStream_t stream) {
  LBM_REQUIRE(frames && outs && B > 0 && B <= 65535 && H > 0 && W > 0 && oh > 0 && ow > 0, "preprocess_split: bad arguments");
  LBM_REQUIRE(((uintptr_t)outs & 7) == 0 && outs_plane % 4 == 0, "preprocess_split: output planes must be 8-byte aligned");
  LBM_REQUIRE(!is_u8 ? in_layout == 0 : (in_layout >= 0 && in_layout <= 2), "preprocess_split: in_layout must be 0 (planar RGB), 1 (interleaved RGB) or 2 (interleaved BGR)");
  ProfScope prof("preprocess", stream, 0.0, (is_u8 ? 1.0 : 4.0) * B * 3 * (double)H * W + 16.0 * B * (double)oh * ow);
  const dim3 gp((ow + 63) / 64, (oh + 3) / 4, B), blk(256);
  float* none = nullptr;
  if (!is_u8) hipLaunchKernelGGL((preprocess_px_kernel<float, 0>), gp, blk, 0, stream, (const float*)frames, none, B, H, W, oh, ow, frame_bstride, outs, outs_plane);
  else if (in_layout == 0) hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 0>), gp, blk, 0, stream, (const uint8_t*)frames, none, B, H, W, oh, ow, frame_bstride, outs, outs_plane);
  else if (in_layout == 1) hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 1>), gp, blk, 0, stream, (const uint8_t*)frames, none, B, H, W, oh, ow, frame_bstride, outs, outs_plane);
  else hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 2>), gp, blk, 0, stream, (const uint8_t*)frames, none, B, H, W, oh, ow, frame_bstride, outs, outs_plane);
  return check_launch("preprocess_split");
}

static int preprocess_check(const void* frames, float* out, int B, int H, int W, int oh, int ow, int out_nhwc, long long& blocks) {
  LBM_REQUIRE(frames && out && B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "preprocess: bad arguments");
  LBM_REQUIRE(out_nhwc == 0 || out_nhwc == 3 || out_nhwc == 4, "preprocess: out_nhwc must be 0, 3 or 4");
  blocks = ((long long)B * 3 * oh * ow + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  return 0;
}

int preprocess(const float* frames, long long frame_bstride, float* out, int B, int H, int W, int oh, int ow, int out_nhwc,
               hipStream_t stream) {
  long long blocks;
  if (preprocess_check(frames, out, B, H, W, oh, ow, out_nhwc, blocks)) return -1;
  ProfScope prof("preprocess", stream, 0.0, 4.0 * B * 3 * ((double)H * W + (double)oh * ow));
  if (out_nhwc == 4 && B <= 65535) {
    hipLaunchKernelGGL((preprocess_px_kernel<float, 0>), dim3((ow + 63) / 64, (oh + 3) / 4, B), dim3(256), 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, (void*)nullptr, 0LL);
    return check_launch("preprocess");
  }
  hipLaunchKernelGGL((preprocess_kernel<float, 0>), dim3((unsigned)blocks), dim3(256), 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, out_nhwc);
  return check_launch("preprocess");
}

int preprocess_u8(const uint8_t* frames, long long frame_bstride, int in_layout, float* out, int B, int H, int W, int oh, int ow,
                  int out_nhwc, hipStream_t stream) {
  long long blocks;
  if (preprocess_check(frames, out, B, H, W, oh, ow, out_nhwc, blocks)) return -1;
  LBM_REQUIRE(in_layout >= 0 && in_layout <= 2, "preprocess_u8: in_layout must be 0 (planar RGB), 1 (interleaved RGB) or 2 (interleaved BGR)");
  ProfScope prof("preprocess_u8", stream, 0.0, (double)B * 3 * ((double)H * W + 4.0 * oh * ow));
  const dim3 grid((unsigned)blocks), blk(256);
  if (out_nhwc == 4 && B <= 65535) {
    const dim3 gp((ow + 63) / 64, (oh + 3) / 4, B);
    if (in_layout == 0) hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 0>), gp, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, (void*)nullptr, 0LL);
    else if (in_layout == 1) hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 1>), gp, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, (void*)nullptr, 0LL);
    else hipLaunchKernelGGL((preprocess_px_kernel<uint8_t, 2>), gp, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, (void*)nullptr, 0LL);
    return check_launch("preprocess_u8");
  }
  if (in_layout == 0) hipLaunchKernelGGL((preprocess_kernel<uint8_t, 0>), grid, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, out_nhwc);
  else if (in_layout == 1) hipLaunchKernelGGL((preprocess_kernel<uint8_t, 1>), grid, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, out_nhwc);
  else hipLaunchKernelGGL((preprocess_kernel<uint8_t, 2>), grid, blk, 0, stream, frames, out, B, H, W, oh, ow, frame_bstride, out_nhwc);
  return check_launch("preprocess_u8");
}

// backbone.py:57-62: cat([p1, up2(p2), up4(p3)]) with pixel-major (NHWC) 64-channel maps -> [B*h*w, 192].
__global__ __launch_bounds__(256) void upcat_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                    const float* __restrict__ p3, float* __restrict__ out, int B, int h, int w,
                                                    void* outs, long long outs_plane) {
  const long long total = (long long)B * h * w * 48;   // float4 pieces
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % 48);
    const long long pix = i / 48;
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const int b = (int)(pix / ((long long)w * h));
    const int src = f >> 4, c4 = (f & 15) * 4;
    float4 r;
    if (src == 0) {
      r = ld4(p1 + pix * 64 + c4);
    } else {
      const int lvl = src;                 // 1: half resolution, 2: quarter resolution
      const int hl = h >> lvl, wl = w >> lvl;
      const float scale = 1.0f / (float)(1 << lvl);
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      resize_coeff(y, scale, hl, y0, y1, ly0, ly1);
      resize_coeff(x, scale, wl, x0, x1, lx0, lx1);
      const float* p = (src == 1 ? p2 : p3) + (long long)b * hl * wl * 64 + c4;
      const float4 v00 = ld4(p + (long long)(y0 * wl + x0) * 64), v01 = ld4(p + (long long)(y0 * wl + x1) * 64);
      const float4 v10 = ld4(p + (long long)(y1 * wl + x0) * 64), v11 = ld4(p + (long long)(y1 * wl + x1) * 64);
      r.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
      r.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
      r.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
      r.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    }
    if (out) st4(out + pix * 192 + f * 4, r);
    if (outs) st4_split(outs, outs_plane, pix * 192 + f * 4, r);
  }
}

int upcat(const float* p1, const float* p2, const float* p3, float* out, int B, int h, int w, hipStream_t stream, void* outs,
          long long outs_plane) {
  LBM_REQUIRE(p1 && p2 && p3 && (out || outs) && B > 0 && h % 4 == 0 && w % 4 == 0, "upcat: bad arguments");
  long long blocks = ((long long)B * h * w * 48 + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  ProfScope prof("upcat", stream, 0.0, 4.0 * B * h * w * (64.0 + 16.0 + 4.0 + 192.0));
  hipLaunchKernelGGL(upcat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p1, p2, p3, out, B, h, w, outs, outs_plane);
  return check_launch("upcat");
}

// lbm_online.py:72-81 / backbone.py:88-92: grid_sample(bilinear, padding_mode='border', align_corners=False)
// of the level-0 token map at one point per query; one wave per query, lane = 4 channels.
__global__ __launch_bounds__(256) void sample_queries_kernel(const float* __restrict__ tok, long long tok_bstride, int ld,
                                                             const float* __restrict__ pts, float* __restrict__ out, int B, int N,
                                                             int h, int w, float net_h, float net_w) {
  const int lane = threadIdx.x & 63;
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= (long long)B * N) return;
  const int b = (int)(q / N);
  const float gx = pts[q * 2] / net_w * 2.0f - 1.0f;
  const float gy = pts[q * 2 + 1] / net_h * 2.0f - 1.0f;
  float px = ((gx + 1.0f) * (float)w - 1.0f) * 0.5f;
  float py = ((gy + 1.0f) * (float)h - 1.0f) * 0.5f;
  px = fminf(fmaxf(px, 0.f), (float)(w - 1));        // border: clip coordinates
  py = fminf(fmaxf(py, 0.f), (float)(h - 1));
  const float x0 = floorf(px), y0 = floorf(py);
  const int xi = (int)x0, yi = (int)y0;
  const float* base = tok + (long long)b * tok_bstride + lane * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tx = t & 1, ty = t >> 1;
    const int xx = xi + tx, yy = yi + ty;
    const float wgt = (tx ? (px - x0) : (x0 + 1.0f - px)) * (ty ? (py - y0) : (y0 + 1.0f - py));
    if (xx < w && yy < h) {
      const float4 v = ld4(base + (long long)(yy * w + xx) * ld);
      acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
    }
  }
  st4(out + q * LBM_C + lane * 4, acc);
}

int sample_queries(const float* tokens, long long tok_bstride, int ld, const float* pts, float* out, int B, int N, int h, int w,
                   int net_h, int net_w, hipStream_t stream) {
  if ((long long)B * N == 0) return 0;
  LBM_REQUIRE(tokens && pts && out && ld % 4 == 0 && tok_bstride % 4 == 0, "sample_queries: bad arguments");
  ProfScope prof("sample_queries", stream, 0.0, 4.0 * LBM_C * 5.0 * B * N);
  hipLaunchKernelGGL(sample_queries_kernel, dim3((unsigned)(((long long)B * N + 3) / 4)), dim3(256), 0, stream, tokens,
                     tok_bstride, ld, pts, out, B, N, h, w, (float)net_h, (float)net_w);
  return check_launch("sample_queries");
}

}  // namespace lbm
