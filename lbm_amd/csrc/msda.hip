// Deformable-attention sampling: bilinear, zero-padded gather over 4 pyramid levels x P points x 4 heads.
//
// Work split: one workgroup (4 waves) per tracked point, one wave per attention head.  In the LBM step the
// k lattice queries of a point are identical copies of its token (transformer.py:317-318), so their sampling
// offsets and soft-maxed weights are equal: they are computed once per (point, head) — lanes 0..4P-1 hold one
// sample each, soft-max by wave reduction — parked in LDS, and reused for all k reference points.
//
// Gather: a head's 64 channels of one value token are 256 contiguous bytes.  The 64 lanes are arranged as
// 4 bilinear taps x 16 lanes x float4, so ONE load instruction fetches all four taps of a sample as four
// 256-byte segments (coalesced, 16 B per lane); per-lane accumulators are combined across the tap groups
// with two cross-lane adds at the end.  Out-of-map taps use a clamped address and a zero weight
// (padding_mode='zeros', align_corners=False: msdeformattn.py:223-228), so the loop has no divergence.
#include "msda.hpp"

namespace lbm {

__device__ __forceinline__ int sel4(const int (&v)[4], int l) {
  return l == 0 ? v[0] : (l == 1 ? v[1] : (l == 2 ? v[2] : v[3]));
}

// PT: compile-time number of points per level (0 = a.P at run time).  blockIdx.y selects the reference point: the k
// lattice cells of a point are independent gathers, and one workgroup walking them in turn was bound by the latency of
// its ~36 dependent-address loads per cell (106 us at 1024 points x 9 cells); the cheap stage 1 is simply repeated.
template <bool EXPLICIT, int PT>
__global__ __launch_bounds__(256) void msda_sample_kernel(MsdaArgs a) {
  __shared__ float sm[LBM_HEADS][3][64];
  const int lane = threadIdx.x & 63;
  const int hd = threadIdx.x >> 6;
  // Row and reference cell of this workgroup.  Workgroups go to the 8 XCDs round-robin in dispatch order; with many streams
  // (xcd_streams) the order is re-dealt so that XCD c works through streams c, c + 8, ... one at a time, the k cells of a point
  // back to back: an L2 then holds the value map of ONE stream instead of a slice of eight (a bijection of the grid; B % 8 == 0).
  long long rq = blockIdx.x;                                         // output row
  int kk_ = blockIdx.y;
  if (a.xcd_streams) {
    const long long lin = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    const int c = (int)(lin & 7);
    const long long t = lin >> 3;
    const int per = a.N * a.k;                                       // workgroups of one stream
    const long long j = t / per;
    const int u = (int)(t - j * per);
    rq = (j * 8 + c) * a.N + u / a.k;          // (two-module launches: unit j * 8 + c runs over (module, stream) -- rows [R, 2R) follow [0, R))
    kk_ = u % a.k;
  }
  const int set = (a.set_rows != 0 && rq >= a.set_rows) ? 1 : 0;     // second module of a two-module launch (workgroup-uniform)
  const long long r = rq - (long long)set * a.set_rows;               // the point this row belongs to
  const int b = (int)(r / a.N);
  const int P = PT ? PT : a.P;
  const int S = LBM_LEVELS * P;

  // ---- stage 1: per-sample normalised offset (or location) and attention weight ----
  float dx = 0.f, dy = 0.f, wt = 0.f;
  if (EXPLICIT) {
    if (lane < S) {
      const long long idx = (rq * LBM_HEADS + hd) * S + lane;
      dx = a.loc[idx * 2];
      dy = a.loc[idx * 2 + 1];
      wt = a.attw[idx];
    }
  } else {
    float logit = -INFINITY;
    if (lane < S) {
      const int l = lane / P;
      const float* o = a.offlog + r * a.ld_offlog + set * a.set_offlog_off;
      const int oi = (hd * S + lane) * 2;
      dx = o[oi] / (float)sel4(a.lvl_w, l);        // msdeformattn.py:161-165 (offset / (w_l, h_l))
      dy = o[oi + 1] / (float)sel4(a.lvl_h, l);
      logit = o[a.logit_col + hd * S + lane];
    }
    const float mx = wave_max(logit);
    const float e = lane < S ? expf(logit - mx) : 0.f;
    wt = e / wave_sum(e);                            // softmax over levels*points (msdeformattn.py:154)
  }
  sm[hd][0][lane] = dx;
  sm[hd][1][lane] = dy;
  sm[hd][2][lane] = wt;
  __syncthreads();

  // ---- stage 2: gather ----
  const int tap = lane >> 4, tx = tap & 1, ty = tap >> 1, cq = lane & 15;
  const float* vb = a.value + (long long)b * a.v_batch_stride + set * a.set_value_off + hd * LBM_HD + cq * 4;
  {
    const int kk = kk_;
    float rx = 0.f, ry = 0.f;
    if (!EXPLICIT) {
      const float* rp = a.ref + (r * a.k + kk) * 2;
      rx = fminf(fmaxf(rp[0] / a.net_w, 0.f), 1.f);  // transformer.py:309-310
      ry = fminf(fmaxf(rp[1] / a.net_h, 0.f), 1.f);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < LBM_LEVELS; ++l) {
      const int hl = a.lvl_h[l], wl = a.lvl_w[l], st = a.lvl_start[l];
      const float fw = (float)wl, fh = (float)hl;
#pragma unroll
      for (int p = 0; p < (PT ? PT : P); ++p) {
        const int i = l * P + p;
        const float lx = rx + sm[hd][0][i];
        const float ly = ry + sm[hd][1][i];
        const float w = sm[hd][2][i];
        // grid = 2*loc - 1 ; pixel = ((grid + 1) * size - 1) / 2     (grid_sample, align_corners=False)
        const float px = (((2.0f * lx - 1.0f) + 1.0f) * fw - 1.0f) * 0.5f;
        const float py = (((2.0f * ly - 1.0f) + 1.0f) * fh - 1.0f) * 0.5f;
        const float x0 = floorf(px), y0 = floorf(py);
        const float wx = tx ? (px - x0) : (x0 + 1.0f - px);
        const float wy = ty ? (py - y0) : (y0 + 1.0f - py);
        const int xi = (int)x0 + tx, yi = (int)y0 + ty;
        const bool inb = (xi >= 0) & (xi < wl) & (yi >= 0) & (yi < hl);
        const int xc = min(max(xi, 0), wl - 1), yc = min(max(yi, 0), hl - 1);
        const float tw = inb ? wx * wy * w : 0.f;
        const float4 v = ld4(vb + (long long)(st + yc * wl + xc) * a.ldv);
        acc.x += tw * v.x;
        acc.y += tw * v.y;
        acc.z += tw * v.z;
        acc.w += tw * v.w;
      }
    }
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (lane < 16) {
      const long long oo = (rq * a.k + kk) * LBM_C + hd * LBM_HD + cq * 4;
      if (a.out) st4(a.out + oo, acc);
      if (a.out_s) st4_split(a.out_s, a.out_plane, oo, acc);
    }
  }
}

int g_msda_xcd = 1;   // debug key 21: 0 = dispatch order

int msda_sample(const MsdaArgs& a, hipStream_t stream) {
  const long long R = (long long)a.B * a.N;
  if (R == 0) return 0;
  LBM_REQUIRE(a.value && (a.out || a.out_s), "msda_sample: null value/out");
  LBM_REQUIRE(a.P >= 1 && a.P <= 16, "msda_sample: P=%d must be in [1,16]", a.P);
  LBM_REQUIRE(a.k >= 1, "msda_sample: k=%d", a.k);
  LBM_REQUIRE(a.ldv % 4 == 0 && a.v_batch_stride % 4 == 0 && ((uintptr_t)a.value & 15) == 0, "msda_sample: value must be 16-byte aligned");
  LBM_REQUIRE(R <= 0x7fffffffLL, "msda_sample: too many rows");
  long long tokens = 0;
  for (int l = 0; l < LBM_LEVELS; ++l) {
    LBM_REQUIRE(a.lvl_h[l] > 0 && a.lvl_w[l] > 0 && a.lvl_start[l] == tokens, "msda_sample: level table is not contiguous");
    tokens += (long long)a.lvl_h[l] * a.lvl_w[l];
  }
  LBM_REQUIRE(a.B == 1 || a.v_batch_stride >= tokens * a.ldv - (a.ldv - LBM_C), "msda_sample: batch stride smaller than one value map");
  const bool explicit_mode = a.loc != nullptr;
  // algorithmic traffic per query: 4 heads x 4 levels x P points x 4 taps x 64 channels read (L2/MALL-resident maps),
  // one 256-channel row written
  const double nq = (double)R * a.k * (a.set_rows ? 2 : 1);
  ProfScope prof("msda_sample", stream, nq * 16.0 * a.P * 4 * 64 * 2, nq * (16.0 * a.P * 4 * 64 * 4 + 4.0 * LBM_C));
  if (explicit_mode) {
    LBM_REQUIRE(a.attw != nullptr && a.k == 1, "msda_sample: explicit mode needs attw and k == 1");
    hipLaunchKernelGGL((msda_sample_kernel<true, 0>), dim3((unsigned)R, 1), dim3(256), 0, stream, a);
  } else {
    LBM_REQUIRE(a.offlog && a.ref && a.net_w > 0.f && a.net_h > 0.f, "msda_sample: implicit mode needs offlog/ref/net size");
    LBM_REQUIRE(a.ld_offlog >= a.logit_col + 16 * a.P && a.logit_col >= 32 * a.P, "msda_sample: offlog layout too small");
    LBM_REQUIRE(a.k <= 65535, "msda_sample: k=%d too large", a.k);
    LBM_REQUIRE(a.set_rows == 0 || a.set_rows == R, "msda_sample: a two-module launch covers every point twice (set_rows == B * N)");
    const dim3 grid((unsigned)(a.set_rows ? 2 * R : R), (unsigned)a.k);
    MsdaArgs b = a;
    b.xcd_streams = (g_msda_xcd && a.B >= 8 && a.B % 8 == 0) ? 1 : 0;
    if (a.P == 9) hipLaunchKernelGGL((msda_sample_kernel<false, 9>), grid, dim3(256), 0, stream, b);
    else if (a.P == 4) hipLaunchKernelGGL((msda_sample_kernel<false, 4>), grid, dim3(256), 0, stream, b);
    else hipLaunchKernelGGL((msda_sample_kernel<false, 0>), grid, dim3(256), 0, stream, b);
  }
  return check_launch("msda_sample");
}

}  // namespace lbm
