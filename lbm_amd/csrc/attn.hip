// Single-query multi-head attention read-out, one wave per point.
//
// Restates nn.MultiheadAttention for a length-1 query (transformer.py:272, 294, 333) without ever
// projecting the keys/values:  score_jh = u_h · (seq_j + temb_j)   with u_h = Wk_hᵀ (Wq_h x + bq_h)/8
// (the key bias adds the same constant to every j and cancels in the softmax), and
//   out_h = Σ_j softmax_j(score_jh) · seq_j      (the value projection Wv_h is applied afterwards by a GEMM,
// which is exact because the softmax weights sum to one).  The per-point sequence (M memory slots of a
// ring buffer, or the k lattice tokens) is therefore read exactly once from HBM: 2·M·C·4 bytes per point
// and layer for the two memories — the algorithmic minimum of SURVEY §8(d).
//
// Lane mapping: 16 lanes per head (head = lane >> 4); lane c of a group owns channels {64 i + 4 c .. +3}.
// Each load instruction therefore covers 256 contiguous bytes per group; score partials are reduced
// inside the 16-lane DPP row (no LDS).  Softmax is online (running max / sum), so any S works.
#include "attn.hpp"

namespace lbm {

__global__ __launch_bounds__(256) void attn_readout_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.R) return;
  const int g = lane >> 4;
  const int c = lane & 15;

  float4 u[4];
  const float* up = a.U + ((long long)row * LBM_HEADS + g) * LBM_C + c * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = ld4(up + i * 64);

  const float* sp = a.seq + (long long)row * a.point_stride + c * 4;
  const uint8_t* ma = a.mask_a ? a.mask_a + (long long)row * a.S : nullptr;
  const uint8_t* mb = a.mask_b ? a.mask_b + (long long)row * a.S : nullptr;

  float m = -INFINITY, l = 0.f;
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll 2
  for (int j = 0; j < a.S; ++j) {
    int phys = a.ring_head + j;
    if (phys >= a.S) phys -= a.S;
    bool ignore = false;
    if (ma) ignore |= ma[phys] != 0;
    if (mb) ignore |= mb[phys] != 0;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ld4(sp + (long long)phys * a.slot_stride + i * 64);
    float part = 0.f;
    if (a.temb) {
      const float* tp = a.temb + (long long)j * LBM_C + c * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = ld4(tp + i * 64);
        part += u[i].x * (v[i].x + t.x) + u[i].y * (v[i].y + t.y) + u[i].z * (v[i].z + t.z) + u[i].w * (v[i].w + t.w);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) part += u[i].x * v[i].x + u[i].y * v[i].y + u[i].z * v[i].z + u[i].w * v[i].w;
    }
    const float s = row16_sum(part);
    if (!ignore) {   // wave-uniform: the masks are per (row, slot)
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn);
      const float p = expf(s - mn);
      l = l * sc + p;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i].x = acc[i].x * sc + p * v[i].x;
        acc[i].y = acc[i].y * sc + p * v[i].y;
        acc[i].z = acc[i].z * sc + p * v[i].z;
        acc[i].w = acc[i].w * sc + p * v[i].w;
      }
      m = mn;
    }
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const long long oo = ((long long)row * LBM_HEADS + g) * LBM_C + c * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 v = make_float4(acc[i].x * inv, acc[i].y * inv, acc[i].z * inv, acc[i].w * inv);
    if (a.out) st4(a.out + oo + i * 64, v);
    if (a.out_s) st4_split(a.out_s, a.out_plane, oo + i * 64, v);
  }
  if (a.valid && lane == 0) a.valid[row] = l > 0.f ? 1 : 0;
}

int attn_readout(const AttnArgs& a, hipStream_t stream) {
  if (a.R == 0) return 0;
  LBM_REQUIRE(a.U && a.seq && (a.out || a.out_s), "attn_readout: null argument");
  LBM_REQUIRE(a.S >= 1 && a.S <= 4096, "attn_readout: sequence length %d out of range", a.S);
  LBM_REQUIRE(a.ring_head >= 0 && a.ring_head < a.S, "attn_readout: ring head %d outside [0,%d)", a.ring_head, a.S);
  LBM_REQUIRE(a.slot_stride % 4 == 0 && a.point_stride % 4 == 0, "attn_readout: strides must be multiples of 4 floats");
  // algorithmic traffic: the S sequence rows once, 4 per-head queries in, 4 per-head read-outs out, the masks
  ProfScope prof("attn_readout", stream, (double)a.R * a.S * 4.0 * LBM_C * 4,
                 (double)a.R * (4.0 * LBM_C * (a.S + 8) + (a.mask_a ? a.S : 0) + (a.mask_b ? a.S : 0)));
  hipLaunchKernelGGL(attn_readout_kernel, dim3((a.R + 3) / 4), dim3(256), 0, stream, a);
  return check_launch("attn_readout");
}

}  // namespace lbm
