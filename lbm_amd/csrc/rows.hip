// One wave per 256-channel row; each lane owns 4 consecutive channels (16-byte accesses).
// This file is compiled WITHOUT packed-fp32 instructions (Makefile, NOPK): its kernels consume registers straight behind
// the partial waits of loads in flight, the pattern that misbehaved beside a second HIP queue when the compiler paired
// components of two fresh loads with v_pk_mov_b32.
#include "rows.hpp"

namespace lbm {

// NS: compile-time number of K-split slabs (0 = in.nslab at run time).  Every load of a row -- the NS
// partial slabs, bias, residual, gamma, beta -- is issued before the first one is used: loads retire in order, and the
// first version (flag, branch, then a run-time loop adding one slab per trip, then the optional terms behind their own
// branches) was a chain of up to twelve dependent round trips to the Infinity Cache per row.  Absent operands are read
// from `gamma` (always valid) and dropped by a select, so the instruction stream has no branch before the sums.
template <int NS>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(RowsIn in, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int rows,
                                                             const uint8_t* __restrict__ keep,
                                                             const float* __restrict__ fallback, RowsOut o) {
  const float* X = in.X;
  float* Y = o.Y;
  float* Yn = o.Yn;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= rows) return;
  const long long off = (long long)row * LBM_C + lane * 4;
  uint8_t kb = 1;
  if (keep != nullptr) kb = keep[row];   // (the few launches with a keep mask pay one extra round trip for it)
  float4 x = ld4(X + off);
  float4 ps[NS > 1 ? NS - 1 : 1];
#pragma unroll
  for (int sl = 1; sl < NS; ++sl) ps[sl - 1] = ld4(X + sl * in.slab_stride + off);
  const bool has_bias = in.bias != nullptr, has_res = in.res != nullptr;
  const bool set2 = in.set_rows != 0 && row >= in.set_rows;            // wave-uniform: a wave owns one row
  const float* gam = set2 ? in.gamma2 : gamma;
  const float* bet = set2 ? in.beta2 : beta;
  const int rrow = set2 ? row - in.set_rows : row;
  const float4 pb = ld4((has_bias ? (set2 ? in.bias2 : in.bias) : gam) + lane * 4);
  const float4 pr = ld4(has_res ? (set2 ? in.res2 : in.res) + (long long)(rrow / in.res_div) * LBM_C + lane * 4 : gam + lane * 4);
  const float4 g = ld4(gam + lane * 4), b = ld4(bet + lane * 4);
  float4 y;
  if (kb == 0) {
    y = ld4(fallback + off);
  } else {
#pragma unroll
    for (int sl = 1; sl < NS; ++sl) {
      x.x += ps[sl - 1].x; x.y += ps[sl - 1].y; x.z += ps[sl - 1].z; x.w += ps[sl - 1].w;
    }
    if (NS == 0) {
      for (int sl = 1; sl < in.nslab; ++sl) {
        const float4 p = ld4(X + sl * in.slab_stride + off);
        x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
      }
    }
    if (has_bias) {
      x.x += pb.x; x.y += pb.y; x.z += pb.z; x.w += pb.w;
    }
    if (has_res) {
      x.x += in.res_scale * pr.x; x.y += in.res_scale * pr.y; x.z += in.res_scale * pr.z; x.w += in.res_scale * pr.w;
    }
    const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / LBM_C);
    const float dx = x.x - mean, dy = x.y - mean, dz = x.z - mean, dw = x.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / LBM_C);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    y = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  }
  if (Y != nullptr) st4(Y + off, y);
  if (o.Ys != nullptr) st4_split(o.Ys, o.ys_plane, off, y);
  if (Yn != nullptr || o.Yns != nullptr) {
    const float n = sqrtf(wave_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w));
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    const float4 v = make_float4(y.x * inv, y.y * inv, y.z * inv, y.w * inv);
    if (Yn != nullptr) st4(Yn + off, v);
    if (o.Yns != nullptr) st4_split(o.Yns, o.yns_plane, off, v);
  }
}

// The serial-load form (one dependent round trip per slab, bias, residual): 7.5 vs 6.7 us per launch at 1024 rows.  It was
// the default while the cause of the frame-pipeline differences was open; kept for A/B runs (lbm_hip_debug_set key 11 = 1).
__global__ __launch_bounds__(256) void layernorm_rows_serial_kernel(RowsIn in, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int rows,
                                                             const uint8_t* __restrict__ keep,
                                                             const float* __restrict__ fallback, RowsOut o) {
  const float* X = in.X;
  float* Y = o.Y;
  float* Yn = o.Yn;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long off = (long long)row * LBM_C + lane * 4;
  float4 y;
  if (keep != nullptr && keep[row] == 0) {
    y = ld4(fallback + off);
  } else {
    float4 x = ld4(X + off);
    for (int sl = 1; sl < in.nslab; ++sl) {
      const float4 p = ld4(X + sl * in.slab_stride + off);
      x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    if (in.bias != nullptr) {
      const float4 p = ld4(in.bias + lane * 4);
      x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    if (in.res != nullptr) {
      const float4 p = ld4(in.res + (long long)(row / in.res_div) * LBM_C + lane * 4);
      x.x += in.res_scale * p.x; x.y += in.res_scale * p.y; x.z += in.res_scale * p.z; x.w += in.res_scale * p.w;
    }
    const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / LBM_C);
    const float dx = x.x - mean, dy = x.y - mean, dz = x.z - mean, dw = x.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / LBM_C);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 g = ld4(gamma + lane * 4), b = ld4(beta + lane * 4);
    y = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  }
  if (Y != nullptr) st4(Y + off, y);
  if (o.Ys != nullptr) st4_split(o.Ys, o.ys_plane, off, y);
  if (Yn != nullptr || o.Yns != nullptr) {
    const float n = sqrtf(wave_sum(y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w));
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    const float4 v = make_float4(y.x * inv, y.y * inv, y.z * inv, y.w * inv);
    if (Yn != nullptr) st4(Yn + off, v);
    if (o.Yns != nullptr) st4_split(o.Yns, o.yns_plane, off, v);
  }
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y,
                                                          long long ldy, long long rows) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float4 x = ld4(X + row * ldx + lane * 4);
    const float n = sqrtf(wave_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w));
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    st4(Y + row * ldy + lane * 4, make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv));
  }
}

int g_ln_serial = 0;   // debug key 11

int layernorm_rows_in(const RowsIn& in, const float* gamma, const float* beta, int rows, const uint8_t* keep,
                      const float* fallback, const RowsOut& out, hipStream_t stream) {
  if (rows == 0) return 0;
  LBM_REQUIRE(in.X && gamma && beta && (out.Y || out.Ys) && rows > 0, "layernorm_rows: null argument");
  LBM_REQUIRE(in.nslab >= 1 && in.res_div >= 1, "layernorm_rows: bad input description");
  LBM_REQUIRE(keep == nullptr || fallback != nullptr, "layernorm_rows: keep mask needs a fallback");
  LBM_REQUIRE(in.set_rows == 0 || (in.gamma2 && in.beta2 && (in.bias == nullptr || in.bias2) && (in.res == nullptr || in.res2) && !g_ln_serial && keep == nullptr),
              "layernorm_rows: a two-set launch needs the second gamma / beta / bias and no keep mask");
  const int nout = (out.Y ? 1 : 0) + (out.Ys ? 1 : 0) + (out.Yn ? 1 : 0) + (out.Yns ? 1 : 0);
  ProfScope prof("layernorm_rows", stream, 0.0, 4.0 * LBM_C * rows * (in.nslab + (in.res ? 1.0 / in.res_div : 0.0) + nout));
  const dim3 grid((rows + 3) / 4);
  if (g_ln_serial) {
    hipLaunchKernelGGL(layernorm_rows_serial_kernel, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out);
    return check_launch("layernorm_rows");
  }
  switch (in.nslab) {
    case 1: hipLaunchKernelGGL(layernorm_rows_kernel<1>, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out); break;
    case 2: hipLaunchKernelGGL(layernorm_rows_kernel<2>, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out); break;
    case 4: hipLaunchKernelGGL(layernorm_rows_kernel<4>, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out); break;
    case 8: hipLaunchKernelGGL(layernorm_rows_kernel<8>, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out); break;
    default: hipLaunchKernelGGL(layernorm_rows_kernel<0>, grid, dim3(256), 0, stream, in, gamma, beta, rows, keep, fallback, out); break;
  }
  return check_launch("layernorm_rows");
}

int layernorm_rows_ex(const float* X, const float* gamma, const float* beta, int rows, const uint8_t* keep,
                      const float* fallback, const RowsOut& out, hipStream_t stream) {
  RowsIn in;
  in.X = X;
  return layernorm_rows_in(in, gamma, beta, rows, keep, fallback, out, stream);
}

int layernorm_rows(const float* X, const float* gamma, const float* beta, float* Y, int rows, const uint8_t* keep,
                   const float* fallback, float* Yn, hipStream_t stream) {
  RowsOut o;
  o.Y = Y;
  o.Yn = Yn;
  return layernorm_rows_ex(X, gamma, beta, rows, keep, fallback, o, stream);
}

int l2norm_rows(const float* X, long long ldx, float* Y, long long ldy, long long rows, hipStream_t stream) {
  if (rows == 0) return 0;
  LBM_REQUIRE(X && Y && ldx % 4 == 0 && ldy % 4 == 0, "l2norm_rows: bad arguments");
  long long blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  ProfScope prof("l2norm_rows", stream, 0.0, 8.0 * LBM_C * rows);
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, ldx, Y, ldy, rows);
  return check_launch("l2norm_rows");
}

}  // namespace lbm
