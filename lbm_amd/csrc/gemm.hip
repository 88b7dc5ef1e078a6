// Exact-fp32 GEMM / implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   Y[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] ) + res_scale * RES[(m / res_div) % res_mod, n]
//
// This one kernel family carries every dense contraction of the tracking step: the attention
// projections (transformer.py:272,294,333), the FFNs (transformer.py:343-355), value/output
// projections of the deformable attention (msdeformattn.py:146,179), the sampling-offset /
// attention-weight linears (msdeformattn.py:150-153), the correlation product (transformer.py:209-212),
// the encoder's 1x1 convolutions (backbone.py:53-63) and — with the CONV loader, which gathers the
// im2col row of an output pixel straight from the NHWC activation — the ResNet-18 body
// (backbone.py:20-26, 49-51; BatchNorm folded into W/bias on the host).
//
// Layout: both operands are K-contiguous ("NT"), staged global -> registers -> LDS in 16-byte pieces,
// LDS rows padded by 4 floats so the per-lane ds_read_b128 fragment reads are conflict-free.  A lane
// (r = lane & 31, h = lane >> 5) of the 32x32x2 MFMA supplies A[r][k = h] and B[k = h][r]; we feed it
// k = 8*s + 4*h + j for j = 0..3 from one float4, which keeps A's and B's k assignment identical, so
// the sum over k is complete after the four MFMAs of step s.
//
// Three tilings are instantiated and chosen by problem size (256 CUs need >= ~256 workgroups):
//   BIG  128x128, 4 waves as 2x2, 2x2 MFMA tiles per wave      (large M*N: value_proj, correlation)
//   MID   64x64,  4 waves as 2x2, one MFMA tile per wave
//   SPK   32x32,  4 waves splitting each K chunk four ways, summed through LDS (M ~ 1e3 rows)
#include "gemm.hpp"
#include "rows.hpp"

#include <string>

namespace lbm {

struct GemmDev {
  const float* X;
  const float* W;
  const float* bias;
  float* Y;
  const float* res;
  int M, N, K;
  int ldx, ldw, ldy, ldres;
  int res_div, res_mod;
  float res_scale;
  int act;
  long long sX, sW, sB, sY, sR;
  // implicit-GEMM convolution (CONV loader): X is [B, H, W, Cin] (Cin a power of two >= 4),
  // row m = (b, oy, ox), column k = (ky, kx, ci); K is kh*kw*Cin rounded up to the chunk size.
  int cH, cW, cin_log2, ckh, ckw, cstride, cpad, cHo, cWo;
};

template <int BM, int BN, int WM, int WN, int WK, int BK, bool CONV>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm_f32_kernel(GemmDev a) {
  constexpr int NT = WM * WN * WK * 64;
  constexpr int LS = BK + 4;                 // padded LDS row stride (floats)
  constexpr int TM = BM / WM / 32;           // 32x32 MFMA tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int KW = BK / WK;                // k extent one wave consumes per chunk
  static_assert(KW % 8 == 0, "each wave consumes k in groups of 8");
  constexpr int A_F4 = BM * BK / 4;          // float4 pieces per A chunk
  constexpr int B_F4 = BN * BK / 4;
  static_assert(A_F4 % NT == 0 && B_F4 % NT == 0, "staging must divide evenly over the threads");
  constexpr int A_PER_T = A_F4 / NT;
  constexpr int B_PER_T = B_F4 / NT;
  constexpr int STAGE = (BM + BN) * LS;      // floats per staging buffer
  constexpr int RED = (WK > 1) ? (WK - 1) * BM * BN : 0;
  constexpr int LDS_FLOATS = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WM * WN);
  const int wm = (wave / WN) % WM;
  const int wn = wave % WN;
  const int r = lane & 31;
  const int h = lane >> 5;

  const int z = blockIdx.z;
  const float* __restrict__ X = a.X + z * a.sX;
  const float* __restrict__ W = a.W + z * a.sW;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;

  float4 ra[A_PER_T], rb[B_PER_T];

  // per-thread constants of the A loader
  long long a_base[A_PER_T];   // GEMM: row offset;  CONV: pixel index of (b, 0, 0)
  int a_iy0[A_PER_T], a_ix0[A_PER_T];
  bool a_ok[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int f = tid + i * NT;
    const int m = m0 + f / (BK / 4);
    a_ok[i] = m < a.M;
    if (CONV) {
      const int hw = a.cHo * a.cWo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / a.cWo, ox = rem - oy * a.cWo;
      a_base[i] = (long long)b * a.cH * a.cW;
      a_iy0[i] = oy * a.cstride - a.cpad;
      a_ix0[i] = ox * a.cstride - a.cpad;
    } else {
      a_base[i] = (long long)m * a.ldx;
      a_iy0[i] = a_ix0[i] = 0;
    }
  }

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int f = tid + i * NT;
      const int k = k0 + (f % (BK / 4)) * 4;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (CONV) {
        const int tap = k >> a.cin_log2, ci = k & ((1 << a.cin_log2) - 1);
        const int ky = tap / a.ckw, kx = tap - ky * a.ckw;
        const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        if (a_ok[i] && ky < a.ckh && iy >= 0 && iy < a.cH && ix >= 0 && ix < a.cW)
          ra[i] = ld4(X + ((a_base[i] + (long long)iy * a.cW + ix) << a.cin_log2) + ci);
      } else {
        if (a_ok[i]) ra[i] = ld4(X + a_base[i] + k);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int f = tid + i * NT;
      const int row = f / (BK / 4), c4 = f % (BK / 4);
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + row < a.N) rb[i] = ld4(W + (long long)(n0 + row) * a.ldw + k0 + c4 * 4);
    }
  };
  auto lstore = [&](int buf) {
    float* As = lds + buf * STAGE;
    float* Bs = As + BM * LS;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int f = tid + i * NT;
      st4(As + (f / (BK / 4)) * LS + (f % (BK / 4)) * 4, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int f = tid + i * NT;
      st4(Bs + (f / (BK / 4)) * LS + (f % (BK / 4)) * 4, rb[i]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = a.K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const float* As = lds + cur * STAGE;
    const float* Bs = As + BM * LS;
#pragma unroll
    for (int s = 0; s < KW / 8; ++s) {
      const int kk = wk * KW + s * 8 + 4 * h;
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ld4(As + ((wm * TM + i) * 32 + r) * LS + kk);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = ld4(Bs + ((wn * TN + j) * 32 + r) * LS + kk);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (WK > 1) {
    // sum the K-split partials: waves wk > 0 park their tiles in LDS, wave group 0 adds them.
    float* red = lds;
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            red[(((wk - 1) * WM * WN + wm * WN + wn) * TM * TN + i * TN + j) * 1024 + e * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int q = 0; q < WK - 1; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            acc[i][j][e] += red[((q * WM * WN + wm * WN + wn) * TM * TN + i * TN + j) * 1024 + e * 64 + lane];
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  float* Y = a.Y + z * a.sY;
  const float* __restrict__ bias = a.bias ? a.bias + z * a.sB : nullptr;
  const float* res = a.res ? a.res + z * a.sR : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + r;
    if (col >= a.N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = m0 + (wm * TM + i) * 32 + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        if (row < a.M) {
          float v = acc[i][j][e] + bv;
          if (a.act == ACT_GELU) v = gelu_erf(v);
          if (res) {
            int rr = row;
            if (a.res_div != 1) rr /= a.res_div;
            if (a.res_mod != 0x7fffffff) rr %= a.res_mod;
            v += a.res_scale * res[(long long)rr * a.ldres + col];
          }
          if (a.act == ACT_RELU_POST) v = fmaxf(v, 0.f);
          Y[(long long)row * a.ldy + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int WK, int BK, bool CONV>
static int launch(const GemmDev& d, int batch, hipStream_t stream) {
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, batch);
  static const std::string name = std::string(CONV ? "conv_f32<" : "gemm_f32<") + std::to_string(BM) + "," + std::to_string(BN) + "," +
                                  std::to_string(WM) + "," + std::to_string(WN) + "," + std::to_string(WK) + "," + std::to_string(BK) + ">";
  const double mnk = (double)d.M * d.N * d.K * batch;
  const double in_elems = CONV ? (double)d.M / (d.cHo * d.cWo) * d.cH * d.cW * (1 << d.cin_log2) : (double)d.M * d.K * batch;
  ProfScope prof(name.c_str(), stream, 2.0 * mnk,
                 4.0 * (in_elems + (double)d.N * d.K * (d.sW || batch == 1 ? batch : 1) + (double)d.M * d.N * batch * (d.res ? 2 : 1)));
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, WK, BK, CONV>), grid, dim3(WM * WN * WK * 64), 0, stream, d);
  return check_launch(CONV ? "conv_f32" : "gemm_f32");
}

template <bool CONV>
static int dispatch(const GemmDev& d, int batch, hipStream_t stream) {
  const long long big = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * batch;
  const long long mid = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) * batch;
  if (big >= 192 && d.N > 64) return launch<128, 128, 2, 2, 1, 16, CONV>(d, batch, stream);
  if (mid >= 192) return launch<64, 64, 2, 2, 1, 32, CONV>(d, batch, stream);
  return launch<32, 32, 1, 1, 4, 32, CONV>(d, batch, stream);
}

int gemm(const GemmArgs& a, hipStream_t stream) {
  LBM_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.M == 0) return 0;
  LBM_REQUIRE(a.K % 32 == 0, "gemm: K=%d must be a multiple of 32", a.K);
  LBM_REQUIRE(a.ldx % 4 == 0 && a.ldw % 4 == 0, "gemm: ldx=%d / ldw=%d must be multiples of 4", a.ldx, a.ldw);
  LBM_REQUIRE(a.ldx >= a.K && a.ldw >= a.K && a.ldy >= a.N, "gemm: leading dims too small");
  LBM_REQUIRE(((uintptr_t)a.X & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: X/W must be 16-byte aligned");
  LBM_REQUIRE(a.sX % 4 == 0 && a.sW % 4 == 0, "gemm: batch strides must keep 16-byte alignment");
  LBM_REQUIRE(a.res == nullptr || (a.res_div >= 1 && a.res_mod >= 1 && a.ldres >= a.N), "gemm: bad residual spec");
  LBM_REQUIRE(a.batch >= 1 && a.batch <= 65535, "gemm: batch=%d out of range", a.batch);
  const bool ln = a.ln_gamma != nullptr;
  if (ln) LBM_REQUIRE(a.N == LBM_C && a.batch == 1 && a.ldy == LBM_C, "gemm: fused LayerNorm needs N == 256, ldy == 256");

  GemmDev d{};
  d.X = a.X; d.W = a.W; d.bias = a.bias; d.Y = a.Y; d.res = a.res;
  d.M = a.M; d.N = a.N; d.K = a.K; d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = a.ldy; d.ldres = a.ldres;
  d.res_div = a.res_div; d.res_mod = a.res_mod; d.res_scale = a.res_scale; d.act = a.act;
  d.sX = a.sX; d.sW = a.sW; d.sB = a.sB; d.sY = a.sY; d.sR = a.sR;
  if (dispatch<false>(d, a.batch, stream)) return -1;
  if (ln) return layernorm_rows(a.Y, a.ln_gamma, a.ln_beta, a.Y, a.M, a.keep, a.fallback, a.Yn, stream);
  return 0;
}

int conv2d_nhwc(const ConvArgs& c, hipStream_t stream) {
  LBM_REQUIRE(c.X && c.W && c.Y, "conv2d: null argument");
  LBM_REQUIRE(c.B >= 1 && c.H >= 1 && c.Wd >= 1 && c.Cout >= 1, "conv2d: bad shape");
  int lg = 0;
  while ((1 << lg) < c.Cin) ++lg;
  LBM_REQUIRE((1 << lg) == c.Cin && c.Cin >= 4, "conv2d: Cin=%d must be a power of two >= 4", c.Cin);
  LBM_REQUIRE(c.kh >= 1 && c.kw >= 1 && c.stride >= 1 && c.pad >= 0, "conv2d: bad kernel spec");
  const int Ho = (c.H + 2 * c.pad - c.kh) / c.stride + 1, Wo = (c.Wd + 2 * c.pad - c.kw) / c.stride + 1;
  LBM_REQUIRE(Ho >= 1 && Wo >= 1, "conv2d: empty output");
  const int Kt = c.kh * c.kw * c.Cin;
  const int Kp = (Kt + 31) / 32 * 32;
  LBM_REQUIRE(c.ldw >= Kp && c.ldw % 4 == 0, "conv2d: packed weight rows need %d (zero-padded) columns, have %d", Kp, c.ldw);
  LBM_REQUIRE(((uintptr_t)c.X & 15) == 0 && ((uintptr_t)c.W & 15) == 0, "conv2d: X/W must be 16-byte aligned");
  LBM_REQUIRE((long long)c.B * Ho * Wo < 0x7fffffffLL, "conv2d: too many output pixels");
  GemmDev d{};
  d.X = c.X; d.W = c.W; d.bias = c.bias; d.Y = c.Y; d.res = c.res;
  d.M = c.B * Ho * Wo; d.N = c.Cout; d.K = Kp; d.ldx = 0; d.ldw = c.ldw; d.ldy = c.Cout; d.ldres = c.Cout;
  d.res_div = 1; d.res_mod = 0x7fffffff; d.res_scale = 1.f; d.act = c.relu ? ACT_RELU_POST : ACT_NONE;
  d.cH = c.H; d.cW = c.Wd; d.cin_log2 = lg; d.ckh = c.kh; d.ckw = c.kw; d.cstride = c.stride; d.cpad = c.pad; d.cHo = Ho; d.cWo = Wo;
  return dispatch<true>(d, 1, stream);
}

// 3x3 stride-2 pad-1 max pooling on NHWC (torchvision resnet stem, backbone.py:23)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                           int Cq, int Ho, int Wo, void* ys, long long ys_plane) {
  const long long total = (long long)B * Ho * Wo * Cq;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % Cq);
    const long long pix = i / Cq;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int iy = oy * 2 - 1 + dy, ix = ox * 2 - 1 + dx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          const float4 v = ld4(x + (((long long)b * H + iy) * W + ix) * Cq * 4 + c4 * 4);
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    if (y) st4(y + pix * Cq * 4 + c4 * 4, m);
    if (ys) st4_split(ys, ys_plane, pix * Cq * 4 + c4 * 4, m);
  }
}

int maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, hipStream_t stream, void* ys, long long ys_plane) {
  LBM_REQUIRE(x && (y || ys) && B >= 1 && H >= 1 && W >= 1 && C % 4 == 0, "maxpool: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long long blocks = ((long long)B * Ho * Wo * (C / 4) + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  ProfScope prof("maxpool3x3s2", stream, 0.0, 4.0 * B * C * ((double)H * W + (double)Ho * Wo));
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, B, H, W, C / 4, Ho, Wo, ys, ys_plane);
  return check_launch("maxpool3x3s2");
}

}  // namespace lbm
