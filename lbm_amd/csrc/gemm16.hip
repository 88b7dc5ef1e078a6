// fp32-grade GEMM / implicit-GEMM convolution on the gfx950 *f16* matrix cores ("split" precision).
//
//   Y[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] ) + res_scale * RES[(m / res_div) % res_mod, n]
//
// gfx950 has no TF32-like fast path for fp32 operands (the fp32 MFMA runs at 1/16 of the f16 rate), so every
// fp32 operand is represented as an unevaluated sum of two halves,
//     x  =  hi + lo * 2^-11,      hi = f16(x),   lo = f16((x - hi) * 2^11)           (22-23 significant bits),
// and the product is assembled from three f16 MFMAs with fp32 accumulation:
//     X·Wᵀ  ≈  Xh·Whᵀ  +  2^-11 (Xh·Wlᵀ + Xl·Whᵀ)                  (the dropped Xl·Wl term is 2^-22 relative).
// Products of two f16 values are exact in fp32, so the only error beyond plain fp32 accumulation is the 2^-23
// representation error of the operands.  Rate: 3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block = 5.3x the
// algorithmic FLOP/s of v_mfma_f32_32x32x2_f32.  The lo halves are kept scaled by 2^11 so they stay in the f16
// normal range; magnitudes are clamped to +-65000 (f16 max 65504) before splitting.
//
// Operands: W (weights, or the per-frame correlation maps) arrives PRE-SPLIT as two f16 planes [N, ldw] (built
// once per model by lbm_amd/engine.py, or by the producing kernel); X stays fp32 in HBM and is split by the
// loader on its way into LDS — that VALU work rides in the issue slots the MFMAs leave free.  Same loader
// variants, tilings and epilogue as gemm.hip.
#include "gemm.hpp"
#include "rows.hpp"

// 1: pin all fragment reads of a chunk in front of its MFMAs (sched_barrier).  Measured and left off: with four waves per SIMD the
// compiler's own interleaving is better here (128 x 128 tile: 48.9 -> 60.6 us at 16384 x 1024 x 256); the row-streaming kernel
// (gemm_rs.hip, two waves per SIMD) is where the pinning pays.
#ifndef LBM_PIN
#define LBM_PIN 0
#endif

#include <cstdlib>
#include <set>
#include <string>

namespace lbm {


struct Gemm16Dev {
  const float* X;
  const _Float16* Wh;
  const _Float16* Wl;
  const float* bias;
  float* Y;
  const float* res;
  int M, N, K;
  int ldx, ldw, ldy, ldres;
  int res_div, res_mod;
  float res_scale;
  int act;
  long long sX, sW, sB, sY, sR;
  int cH, cW, cin_log2, ckh, ckw, cstride, cpad, cHo, cWo;
};

template <int BM, int BN, int WM, int WN, int WK, int BK, bool CONV>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm_s16_kernel(Gemm16Dev a) {
  constexpr int NT = WM * WN * WK * 64;
  constexpr int LSB = BK * 2 + 16;           // LDS row stride in bytes (an odd number of 16-byte slots)
  static_assert((LSB / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int KW = BK / WK;                // k extent one wave consumes per chunk
  static_assert(KW % 16 == 0, "each wave consumes k in groups of 16");
  constexpr int A_P = BM * BK / 4;           // float4 pieces of the fp32 A chunk
  constexpr int B_P = BN * BK / 8;           // 16-byte pieces of ONE f16 plane of the B chunk
  static_assert(A_P % NT == 0 && B_P % NT == 0, "staging must divide evenly over the threads");
  constexpr int A_PER_T = A_P / NT;
  constexpr int B_PER_T = B_P / NT;
  constexpr int STAGE = 2 * (BM + BN) * LSB;  // bytes: A hi, A lo, B hi, B lo
  constexpr int RED = (WK > 1) ? (WK - 1) * BM * BN * 4 : 0;
  constexpr int LDS_BYTES = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

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
  const _Float16* __restrict__ Wh = a.Wh + z * a.sW;
  const _Float16* __restrict__ Wl = a.Wl + z * a.sW;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;

  float4 ra[A_PER_T];
  uint4 rbh[B_PER_T], rbl[B_PER_T];

  long long a_base[A_PER_T];
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
      const int row = f / (BK / 8), c8 = f % (BK / 8);
      rbh[i] = make_uint4(0u, 0u, 0u, 0u);
      rbl[i] = make_uint4(0u, 0u, 0u, 0u);
      if (n0 + row < a.N) {
        const long long o = (long long)(n0 + row) * a.ldw + k0 + c8 * 8;
        rbh[i] = *reinterpret_cast<const uint4*>(Wh + o);
        rbl[i] = *reinterpret_cast<const uint4*>(Wl + o);
      }
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* Ah = lds + buf * STAGE;
    unsigned char* Al = Ah + BM * LSB;
    unsigned char* Bh = Al + BM * LSB;
    unsigned char* Bl = Bh + BN * LSB;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int f = tid + i * NT;
      const int o = (f / (BK / 4)) * LSB + (f % (BK / 4)) * 8;
      half4 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<half4*>(Ah + o) = hi;
      *reinterpret_cast<half4*>(Al + o) = lo;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int f = tid + i * NT;
      const int o = (f / (BK / 8)) * LSB + (f % (BK / 8)) * 16;
      *reinterpret_cast<uint4*>(Bh + o) = rbh[i];
      *reinterpret_cast<uint4*>(Bl + o) = rbl[i];
    }
  };

  f32x16 acc[TM][TN], acx[TM][TN];   // hi·hi and the 2^11-scaled cross terms
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        acx[i][j][e] = 0.f;
      }

  const int nk = a.K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const unsigned char* Ah = lds + cur * STAGE;
    const unsigned char* Al = Ah + BM * LSB;
    const unsigned char* Bh = Al + BM * LSB;
    const unsigned char* Bl = Bh + BN * LSB;
#pragma unroll
    for (int s = 0; s < KW / 16; ++s) {
      // lane (r, h) supplies A[r][k = 8h + j], B[k = 8h + j][r], j = 0..7, of this 16-deep step
      const int kb = (wk * KW + s * 16 + 8 * h) * 2;
      half8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = ((wm * TM + i) * 32 + r) * LSB + kb;
        fah[i] = *reinterpret_cast<const half8*>(Ah + o);
        fal[i] = *reinterpret_cast<const half8*>(Al + o);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = ((wn * TN + j) * 32 + r) * LSB + kb;
        fbh[j] = *reinterpret_cast<const half8*>(Bh + o);
        fbl[j] = *reinterpret_cast<const half8*>(Bl + o);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[j], acx[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh[j], acx[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] += acx[i][j][e] * (1.0f / 2048.0f);

  if constexpr (WK > 1) {
    float* red = reinterpret_cast<float*>(lds);
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
static int launch16(const Gemm16Dev& d, int batch, hipStream_t stream) {
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, batch);
  static const std::string name = std::string(CONV ? "conv_s16<" : "gemm_s16<") + std::to_string(BM) + "," + std::to_string(BN) + "," +
                                  std::to_string(WM) + "," + std::to_string(WN) + "," + std::to_string(WK) + "," + std::to_string(BK) + ">";
  const double mnk = (double)d.M * d.N * d.K * batch;
  const double in_elems = CONV ? (double)d.M / (d.cHo * d.cWo) * d.cH * d.cW * (1 << d.cin_log2) : (double)d.M * d.K * batch;
  ProfScope prof(name.c_str(), stream, 2.0 * mnk,
                 4.0 * (in_elems + (double)d.N * d.K * (d.sW || batch == 1 ? batch : 1) + (double)d.M * d.N * batch * (d.res ? 2 : 1)));
  hipLaunchKernelGGL((gemm_s16_kernel<BM, BN, WM, WN, WK, BK, CONV>), grid, dim3(WM * WN * WK * 64), 0, stream, d);
  return check_launch(CONV ? "conv_s16" : "gemm_s16");
}

static int g_force_tile16 = 0;
static int g_dbg16 = 0;
static int g_force_conv = 0;
static long long g_rs_min_rows = 32768;   // measured (scripts/gemm_rs_check.py): below, the 2-step cost of loading the stationary fragments eats the gain
void conv16_force(int mode) { g_force_conv = mode; }
void gemm16_force_tile(int t) { g_force_tile16 = t; }
void gemm16_debug(int flags) { g_dbg16 = flags; }

template <bool CONV>
static int dispatch16(const Gemm16Dev& d, int batch, hipStream_t stream) {
  const long long big = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * batch;
  const long long mid = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) * batch;
  int t = g_force_tile16;
  if (t == 0) t = (big >= 192 && d.N > 64) ? 1 : (mid >= 192 ? 2 : 3);
  if (t == 3 && d.K % 64 != 0) t = 2;
  if (t == 1) return launch16<128, 128, 2, 2, 1, 32, CONV>(d, batch, stream);
  if (t == 2) return launch16<64, 64, 2, 2, 1, 32, CONV>(d, batch, stream);
  return launch16<32, 32, 1, 1, 4, 64, CONV>(d, batch, stream);
}

int gemm16(const GemmArgs& a, hipStream_t stream) {
  LBM_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "gemm16: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.M == 0) return 0;
  LBM_REQUIRE(a.Wh && a.Wl, "gemm16: the W operand must be pre-split into f16 planes");
  LBM_REQUIRE(a.K % 32 == 0, "gemm16: K=%d must be a multiple of 32", a.K);
  LBM_REQUIRE(a.ldx % 4 == 0 && a.ldw % 8 == 0, "gemm16: ldx=%d must be a multiple of 4, ldw=%d of 8", a.ldx, a.ldw);
  LBM_REQUIRE(a.ldx >= a.K && a.ldw >= a.K && a.ldy >= a.N, "gemm16: leading dims too small");
  LBM_REQUIRE(((uintptr_t)a.X & 15) == 0 && ((uintptr_t)a.Wh & 15) == 0 && ((uintptr_t)a.Wl & 15) == 0,
              "gemm16: X/Wh/Wl must be 16-byte aligned");
  LBM_REQUIRE(a.sX % 4 == 0 && a.sW % 8 == 0, "gemm16: batch strides must keep 16-byte alignment");
  LBM_REQUIRE(a.res == nullptr || (a.res_div >= 1 && a.res_mod >= 1 && a.ldres >= a.N), "gemm16: bad residual spec");
  LBM_REQUIRE(a.batch >= 1 && a.batch <= 65535, "gemm16: batch=%d out of range", a.batch);
  const bool ln = a.ln_gamma != nullptr;
  if (ln) LBM_REQUIRE(a.N == LBM_C && a.batch == 1 && a.ldy == LBM_C, "gemm16: fused LayerNorm needs N == 256, ldy == 256");

  Gemm16Dev d{};
  d.X = a.X; d.Wh = (const _Float16*)a.Wh; d.Wl = (const _Float16*)a.Wl; d.bias = a.bias; d.Y = a.Y; d.res = a.res;
  d.M = a.M; d.N = a.N; d.K = a.K; d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = a.ldy; d.ldres = a.ldres;
  d.res_div = a.res_div; d.res_mod = a.res_mod; d.res_scale = a.res_scale; d.act = a.act;
  d.sX = a.sX; d.sW = a.sW; d.sB = a.sB; d.sY = a.sY; d.sR = a.sR;
  if (dispatch16<false>(d, a.batch, stream)) return -1;
  if (ln) return layernorm_rows(a.Y, a.ln_gamma, a.ln_beta, a.Y, a.M, a.keep, a.fallback, a.Yn, stream);
  return 0;
}

int conv2d_nhwc16(const ConvArgs& c, hipStream_t stream) {
  LBM_REQUIRE(c.X && c.Wh && c.Wl && c.Y, "conv2d16: null argument (weights must be pre-split)");
  LBM_REQUIRE(c.B >= 1 && c.H >= 1 && c.Wd >= 1 && c.Cout >= 1, "conv2d16: bad shape");
  int lg = 0;
  while ((1 << lg) < c.Cin) ++lg;
  LBM_REQUIRE((1 << lg) == c.Cin && c.Cin >= 4, "conv2d16: Cin=%d must be a power of two >= 4", c.Cin);
  LBM_REQUIRE(c.kh >= 1 && c.kw >= 1 && c.stride >= 1 && c.pad >= 0, "conv2d16: bad kernel spec");
  const int Ho = (c.H + 2 * c.pad - c.kh) / c.stride + 1, Wo = (c.Wd + 2 * c.pad - c.kw) / c.stride + 1;
  LBM_REQUIRE(Ho >= 1 && Wo >= 1, "conv2d16: empty output");
  const int Kt = c.kh * c.kw * c.Cin;
  const int Kp = (Kt + 31) / 32 * 32;
  LBM_REQUIRE(c.ldw >= Kp && c.ldw % 8 == 0, "conv2d16: packed weight rows need %d (zero-padded) columns, have %d", Kp, c.ldw);
  LBM_REQUIRE(((uintptr_t)c.X & 15) == 0 && ((uintptr_t)c.Wh & 15) == 0 && ((uintptr_t)c.Wl & 15) == 0, "conv2d16: operands must be 16-byte aligned");
  LBM_REQUIRE((long long)c.B * Ho * Wo < 0x7fffffffLL, "conv2d16: too many output pixels");
  Gemm16Dev d{};
  d.X = c.X; d.Wh = (const _Float16*)c.Wh; d.Wl = (const _Float16*)c.Wl; d.bias = c.bias; d.Y = c.Y; d.res = c.res;
  d.M = c.B * Ho * Wo; d.N = c.Cout; d.K = Kp; d.ldx = 0; d.ldw = c.ldw; d.ldy = c.Cout; d.ldres = c.Cout;
  d.res_div = 1; d.res_mod = 0x7fffffff; d.res_scale = 1.f; d.act = c.relu ? ACT_RELU_POST : ACT_NONE;
  d.cH = c.H; d.cW = c.Wd; d.cin_log2 = lg; d.ckh = c.kh; d.ckw = c.kw; d.cstride = c.stride; d.cpad = c.pad; d.cHo = Ho; d.cWo = Wo;
  return dispatch16<true>(d, 1, stream);
}

// fp32 [rows, cols] -> two f16 planes (hi, lo * 2^11); used for weights at pack time and for per-frame operands
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                           _Float16* __restrict__ lo, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    half4 h, l;
    split4(ld4(x + i * 4), h, l);
    *reinterpret_cast<half4*>(hi + i * 4) = h;
    *reinterpret_cast<half4*>(lo + i * 4) = l;
  }
}

int split_planes(const float* x, void* hi, void* lo, long long n, hipStream_t stream) {
  if (n == 0) return 0;
  LBM_REQUIRE(x && hi && lo && n % 4 == 0, "split_planes: bad arguments (n must be a multiple of 4)");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  ProfScope prof("split_planes", stream, 0.0, 8.0 * n);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, (_Float16*)hi, (_Float16*)lo, n / 4);
  return check_launch("split_planes");
}

// =================================================================================================================
// gemm_ss: BOTH operands pre-split (f16 hi / lo planes, K-contiguous) and staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no VALU), so the whole loop body is ds_read_b128 + MFMA.
//
// LDS image of one operand plane: [rows][32 f16] = 64-byte rows, linear in the order the DMA writes them (one
// wave-instruction = 16 rows x 64 B = 1 KiB).  Bank conflicts are removed by an XOR swizzle applied on the
// SOURCE side of the DMA and again on the fragment read: the 16-byte chunk c of row r lives in slot
// c ^ ((r >> 2) & 3).  A ds_read_b128 lane group covers 16 rows with distinct residues mod 16, so its 16 slots
// (4 (r & 3) + (c ^ ((r >> 2) & 3))) are all different: conflict-free.
// Pipeline: STAGES LDS buffers; the DMA of chunk kt + STAGES - 1 is in flight while chunk kt is multiplied;
// each wave retires its own DMAs with a counted s_waitcnt vmcnt(N) and a raw s_barrier publishes them
// (cdna_hip_programming.md §5 "Pipelining across barriers").
// =================================================================================================================
struct GemmSSDev {
  const _Float16* A;   // hi plane [M, lda]; lo plane at A + a_plane
  const _Float16* B;   // hi plane [N, ldb]; lo plane at B + b_plane
  long long a_plane, b_plane;
  const float* bias;
  float* Y;            // optional fp32 output [M, ldy]
  _Float16* Ys;        // optional split output: hi plane [M, ldys], lo plane at Ys + ys_plane
  long long ys_plane;
  const float* res;
  int M, N, K;
  int lda, ldb, ldy, ldys, ldres;
  int res_div, res_mod;
  float res_scale;
  int act;
  long long sA, sB, sBias, sY, sYs, sR;   // batch strides (elements)
  int dbg;   // ablation switches (tuning only): 1 = no DMA after the prologue (FULL: none at all), 4 = no output stores, 8 = return at once, 16 = direct epilogue
  // implicit-GEMM convolution (CONV): A is an NHWC activation as split planes [B, cH, cW, Cin], Cin a multiple of 32;
  // row m = (b, oy, ox), column k = (ky, kx, ci).  Taps outside the image read `zeros` (>= 16 zero bytes).
  int cH, cW, cin_log2, ckw, cstride, cpad, cHo, cWo;
  const void* zeros;
  int kt0_per_z;   // K-split over gridDim.z for convolutions: block z starts at chunk z * kt0_per_z
  float* cand_v;   // TOPK epilogue: per-tile candidates [batch * M, ntile_n * topk_k] (values / global columns)
  int* cand_i;
  int topk_k;
  int xcd_remap;   // 1: XCD-contiguous tile order (grid of a multiple of 8 workgroups per z)
  int vec_ok;      // outputs / residual admit 16-byte row pieces (N, leading dimensions, strides multiples of 4; aligned bases)
};

static int vec_ok_of(const GemmSSDev& d) {
  auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
  if (g_dbg16 & 16) return 0;       // tuning: the direct accumulator-layout epilogue
  return d.N % 4 == 0 && (!d.Y || (d.ldy % 4 == 0 && d.sY % 4 == 0 && al(d.Y, 16))) &&
         (!d.Ys || (d.ldys % 4 == 0 && d.sYs % 4 == 0 && d.ys_plane % 4 == 0 && al(d.Ys, 8))) &&
         (!d.res || (d.ldres % 4 == 0 && d.sR % 4 == 0 && al(d.res, 16)));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if constexpr (N == 23) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
  else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
  else if constexpr (N == 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
  else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if constexpr (N == 29) asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
  else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
  else if constexpr (N == 31) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if constexpr (N == 33) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
  else if constexpr (N == 34) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
  else if constexpr (N == 35) asm volatile("s_waitcnt vmcnt(35)" ::: "memory");
  else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
  else if constexpr (N == 37) asm volatile("s_waitcnt vmcnt(37)" ::: "memory");
  else if constexpr (N == 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
  else if constexpr (N == 39) asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
  else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if constexpr (N == 41) asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
  else if constexpr (N == 42) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
  else if constexpr (N == 43) asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
  else if constexpr (N == 44) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
  else if constexpr (N == 45) asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
  else if constexpr (N == 46) asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
  else if constexpr (N == 47) asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
  else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// compile-time unrolled consumer loop of the FULL variant: stage KT is multiplied once everything issued up to it
// has landed (the (STAGES - 1 - KT) younger stages may still be in flight); no buffer is ever reused, so one
// barrier per stage suffices.
template <int KT, int STAGES, int PPW, typename F>
__device__ __forceinline__ void pf_steps(F& f) {
  if constexpr (KT < STAGES) {
    wait_vmcnt<(STAGES - 1 - KT) * PPW>();
    __builtin_amdgcn_s_barrier();
    f(KT);
    pf_steps<KT + 1, STAGES, PPW>(f);
  }
}

// FULL: K / 32 == STAGES exactly; the DMA of the whole K extent is issued up front (STAGES x stage bytes of LDS, one
// workgroup per CU).  This is the small-problem variant: with ~1 workgroup per CU nothing else hides the DMA
// latency, so it is paid once per workgroup instead of once per chunk.
// NPL: operand planes staged and multiplied -- 2 = hi and lo (fp32-grade, three MFMAs per block), 1 = hi only (plain f16
// operands, one MFMA per block, half the staging bytes: the reduced-precision mode, step precision 2).
template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool FULL = false, int TOPK = 0, int NPL = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm_ss_kernel(GemmSSDev a) {
#if defined(__HIP_DEVICE_COMPILE__)   // LDS-DMA builtins and s_waitcnt asm exist on the device side only
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;          // bytes of one plane of one stage
  constexpr int STAGE = NPL * (PLANE_A + PLANE_B);             // [A hi][A lo][B hi][B lo] (NPL = 1: [A hi][B hi])
  constexpr int PIECES = STAGE / 1024;                         // 1-KiB DMA pieces per stage
  static_assert(PIECES % NW == 0, "DMA pieces must divide evenly over the waves");
  constexpr int PPW = PIECES / NW;                             // pieces (= DMA instructions) per wave and stage
  constexpr int TP = BN + 4;                                   // pitch (floats) of the output tile staged in the epilogue
  constexpr int LDS_BYTES = STAGES * STAGE > BM * TP * 4 ? STAGES * STAGE : BM * TP * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int z = blockIdx.z;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_remap) {
    // Workgroups are dealt round-robin over the 8 XCDs (id % 8 names the XCD); give every XCD one CONTIGUOUS run of the
    // tile sequence (column index fastest), so the row panels it touches are fetched into ONE L2 instead of all eight.
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int id = by * gx + bx;
    const int t = (id & 7) * (total >> 3) + (id >> 3);
    by = t / gx;
    bx = t - by * gx;
  }
  const int m0 = by * BM, n0 = bx * BN;
  if (LBM_ABL(a.dbg & 8)) return;                                       // ablation: launch skeleton only

  // ---- DMA source addresses: piece p covers 16 rows of one plane; lane -> (row = lane / 4, slot = lane % 4) ----
  const _Float16* src[PPW];
  int c_iy0[PPW], c_ix0[PPW];      // CONV: top-left input pixel of the A row this lane fetches (or a huge value: no row)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    c_iy0[i] = c_ix0[i] = 0;
    const int p = wave * PPW + i;                              // piece index inside the stage, in LDS order
    const int pb = p * 1024;
    const bool isB = pb >= NPL * PLANE_A;
    const int off = isB ? pb - NPL * PLANE_A : pb;
    const int plane_bytes = isB ? PLANE_B : PLANE_A;
    const int lo = off >= plane_bytes ? 1 : 0;
    const int row = (off - lo * plane_bytes) / 64 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);           // which 16-byte source chunk lands in this lane's slot
    if (isB) {
      const int n = min(n0 + row, a.N - 1);
      src[i] = a.B + z * a.sB + lo * a.b_plane + (long long)n * a.ldb + chunk * 8;
    } else if (CONV) {
      const int m = m0 + row;
      const int hw = a.cHo * a.cWo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / a.cWo, ox = rem - oy * a.cWo;
      c_iy0[i] = m < a.M ? oy * a.cstride - a.cpad : -0x40000000;
      c_ix0[i] = ox * a.cstride - a.cpad;
      // pointer to channel (chunk * 8) of input pixel (b, iy0, ix0); may lie outside the tensor until a tap is added
      src[i] = a.A + lo * a.a_plane + ((((long long)b * a.cH + c_iy0[i]) * a.cW + c_ix0[i]) << a.cin_log2) + chunk * 8;
      if (m >= a.M) src[i] = a.A;
    } else {
      const int m = min(m0 + row, a.M - 1);
      src[i] = a.A + z * a.sA + lo * a.a_plane + (long long)m * a.lda + chunk * 8;
    }
  }
  const int ktz = CONV ? z * a.kt0_per_z : 0;
  auto issue = [&](int stage, int kt) {
    // CONV: the 32 channels of chunk kt belong to one tap (Cin % 32 == 0): (ky, kx, ci0) are block-uniform
    int ky = 0, kx = 0;
    long long tap_off = 0;
    if (CONV) {
      const int k = (ktz + kt) * 32;
      const int tap = k >> a.cin_log2, ci0 = k & ((1 << a.cin_log2) - 1);
      ky = tap / a.ckw;
      kx = tap - ky * a.ckw;
      tap_off = (((long long)ky * a.cW + kx) << a.cin_log2) + ci0;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      unsigned char* dst = lds + stage * STAGE + (wave * PPW + i) * 1024;
      const bool isB = (wave * PPW + i) * 1024 >= NPL * PLANE_A;
      const _Float16* g;
      if (CONV && !isB) {
        const bool ok = (unsigned)(c_iy0[i] + ky) < (unsigned)a.cH && (unsigned)(c_ix0[i] + kx) < (unsigned)a.cW;
        g = ok ? src[i] + tap_off : reinterpret_cast<const _Float16*>(a.zeros);
      } else {
        g = src[i] + (long long)(ktz + kt) * 32;
      }
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)dst, 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN], acx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        acx[i][j][e] = 0.f;
      }

  const int nk = a.K / 32;
#pragma unroll
  for (int s = 0; s < (FULL ? STAGES : STAGES - 1); ++s)
    if ((FULL && !(LBM_ABL(a.dbg & 1))) || (!FULL && s < nk)) issue(s, s);

  // fragment byte offsets inside a plane: row * 64 + ((c ^ ((row >> 2) & 3)) * 16), c = 2 s + h
  int offA[TM][2], offB[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wm * TM + i) * 32 + r;
#pragma unroll
    for (int s = 0; s < 2; ++s) offA[i][s] = row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = (wn * TN + j) * 32 + r;
#pragma unroll
    for (int s = 0; s < 2; ++s) offB[j][s] = row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4);
  }

  // All fragments of the chunk (both 16-deep steps) are requested before the first MFMA, so the second step's LDS
  // reads are in flight under the first step's MFMAs (the compiler counts lgkmcnt per use).  No run-time switches in
  // here: a scalar branch between the reads and the MFMAs stops that overlap (seen in the ISA of the first version).
  auto compute = [&](int stage) {
    const unsigned char* Ah = lds + stage * STAGE;
    const unsigned char* Al = Ah + PLANE_A;                    // (NPL = 1: not staged, never read)
    const unsigned char* Bh = Ah + NPL * PLANE_A;
    const unsigned char* Bl = Bh + PLANE_B;
    half8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fah[s][i] = *reinterpret_cast<const half8*>(Ah + offA[i][s]);
        if constexpr (NPL == 2) fal[s][i] = *reinterpret_cast<const half8*>(Al + offA[i][s]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fbh[s][j] = *reinterpret_cast<const half8*>(Bh + offB[j][s]);
        if constexpr (NPL == 2) fbl[s][j] = *reinterpret_cast<const half8*>(Bl + offB[j][s]);
      }
    }
    // Pin the reads in front of the MFMAs: left alone, hipcc sinks each read to just before its first use (one register set, a
    // full lgkmcnt(0) wait per k-step) and the matrix cores idle for the LDS latency (scripts/micro/mfma_peak.hip, modes 2 / 4).
#if LBM_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbh[s][j], acc[i][j], 0, 0, 0);
          if constexpr (NPL == 2) {
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbl[s][j], acx[i][j], 0, 0, 0);
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[s][i], fbh[s][j], acx[i][j], 0, 0, 0);
          }
        }
  };
  if constexpr (FULL) {
    pf_steps<0, STAGES, PPW>(compute);
  } else {
    // The DMA of chunk kt+STAGES-1 is issued BEFORE this chunk's wait and barrier (its stage was released by the barrier
    // that closes the previous iteration).  A one-barrier form that issues it after the barrier measured 15-25 % slower on
    // the large grids (64 -> 81 us at 16320 x 1536 x 256): the issue-to-landing latency is what the loop is exposed to.
    for (int kt = 0; kt < nk; ++kt) {
      const int pre = kt + STAGES - 1;
      if (pre < nk && !(LBM_ABL(a.dbg & 1))) {
        issue(pre % STAGES, pre);
        wait_vmcnt<(STAGES - 1) * PPW>();        // everything but the STAGES-1 newest stages has landed
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      compute(kt % STAGES);
      // all waves must be done reading this stage before the next iteration's DMA overwrites it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue ----
  if (LBM_ABL(a.dbg & 4)) return;
  float* Y = a.Y ? a.Y + z * a.sY : nullptr;
  _Float16* Yh = a.Ys ? a.Ys + z * a.sYs : nullptr;
  _Float16* Yl = a.Ys ? Yh + a.ys_plane : nullptr;
  const float* __restrict__ bias = a.bias ? a.bias + z * a.sBias : nullptr;
  const float* res = a.res ? a.res + z * a.sR : nullptr;
  if (a.vec_ok) {
    // The accumulator layout gives a lane one column of 16 scattered rows: stored as it stands, a split output goes
    // out as 2-byte pieces (64 B per row and instruction).  Stage act(acc + bias) through LDS instead (the operand
    // stages are dead by now) and write whole rows of the tile, 16 bytes per lane: measured 9.6 -> 7.x us on the
    // 1024 x 1024 x 256 product of the per-point chain, where the stores were 40 % of the kernel.
    float* T = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cl = (wn * TN + j) * 32 + r;
      const float bv = (bias && n0 + cl < a.N) ? bias[n0 + cl] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rl = (wm * TM + i) * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = (NPL == 2 ? acc[i][j][e] + acx[i][j][e] * (1.0f / 2048.0f) : acc[i][j][e]) + bv;
          if (a.act == ACT_GELU) v = gelu_erf(v);
          T[(rl + (e & 3) + 8 * (e >> 2)) * TP + cl] = v;
        }
      }
    }
    __syncthreads();
    if constexpr (TOPK > 0) {
      // Row-wise top-k pre-selection on the staged tile: 4 adjacent lanes share a row (NW * 64 == 4 * BM), lane `sub` walks
      // columns sub, sub + 4, ... (conflict-free: 16 rows x 4 lanes cover the 64 banks) keeping a sorted list of its
      // topk_k best in registers; the four lists are merged by topk_k rounds of a 4-lane arg-max and lane 0 writes the
      // pair.  The 64 KB tile itself never leaves the CU.
      static_assert(NW * 64 == 4 * BM, "TOPK epilogue: four lanes per tile row");
      constexpr int KM = TOPK;                      // list depth: the smallest of 1 / 4 / 9 that holds topk_k
      const int rl = tid >> 2, sub = tid & 3;
      const int row = m0 + rl;
      const int kk = a.topk_k;
      float lv[KM];
      int li[KM];
#pragma unroll
      for (int t = 0; t < KM; ++t) {
        lv[t] = -INFINITY;
        li[t] = 0x7fffffff;
      }
      // lane `sub` owns the 32 consecutive columns [32 sub, 32 sub + 32) of its row, read as eight 16-byte pieces
      const float* Tr = T + rl * TP + sub * (BN / 4);
#pragma unroll
      for (int i4 = 0; i4 < BN / 16; ++i4) {
        const float4 q = *reinterpret_cast<const float4*>(Tr + i4 * 4);
        const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = sub * (BN / 4) + i4 * 4 + e;
          const float v = qv[e];
          if (n0 + c < a.N && v > lv[KM - 1]) {        // columns ascend: a tie never displaces an earlier column
            lv[KM - 1] = v;
            li[KM - 1] = n0 + c;
#pragma unroll
            for (int t = KM - 1; t > 0; --t) {
              if (lv[t] > lv[t - 1]) {                   // strict: equal values keep column order
                const float tv = lv[t]; lv[t] = lv[t - 1]; lv[t - 1] = tv;
                const int ti = li[t]; li[t] = li[t - 1]; li[t - 1] = ti;
              }
            }
          }
        }
      }
      const long long ntile = (a.N + BN - 1) / BN;
      const long long co = ((long long)z * a.M + row) * ntile * kk + (long long)bx * kk;
      for (int t = 0; t < kk; ++t) {
        float bv = lv[0];
        int bi = li[0];
#pragma unroll
        for (int sft = 1; sft < 4; sft <<= 1) {
          const float ov = __shfl_xor(bv, sft);
          const int oi = __shfl_xor(bi, sft);
          if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
          }
        }
        if (li[0] == bi && bi != 0x7fffffff) {          // the winner pops its head
#pragma unroll
          for (int q = 0; q < KM - 1; ++q) {
            float nv = lv[q + 1];
            asm volatile("" : "+v"(nv));      // one register at a time: keeps the shift from becoming v_pk_mov_b32
            lv[q] = nv;
            int ni = li[q + 1];
            asm volatile("" : "+v"(ni));
            li[q] = ni;
          }
          lv[KM - 1] = -INFINITY;
          li[KM - 1] = 0x7fffffff;
        }
        if (sub == 0 && row < a.M) {
          a.cand_v[co + t] = bv;
          a.cand_i[co + t] = bi;
        }
      }
      return;
    }
    constexpr int C4 = BN / 4;
    for (int c = tid; c < BM * C4; c += NW * 64) {
      const int rl = c / C4, c4 = c - rl * C4;
      const int row = m0 + rl, col = n0 + c4 * 4;
      if (row >= a.M || col >= a.N) continue;
      float4 v = *reinterpret_cast<const float4*>(T + rl * TP + c4 * 4);
      if (res) {
        int rr = row;
        if (a.res_div != 1) rr /= a.res_div;
        if (a.res_mod != 0x7fffffff) rr %= a.res_mod;
        const float4 p = ld4(res + (long long)rr * a.ldres + col);
        v.x += a.res_scale * p.x; v.y += a.res_scale * p.y; v.z += a.res_scale * p.z; v.w += a.res_scale * p.w;
      }
      if (a.act == ACT_RELU_POST) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (Y) st4(Y + (long long)row * a.ldy + col, v);
      if (Yh) {
        if (a.ys_plane) st4_split(Yh, a.ys_plane, (long long)row * a.ldys + col, v);
        else st4h(Yh + (long long)row * a.ldys + col, v);       // plain f16 output (hi plane only: f16-storage mode)
      }
    }
    return;
  }
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
          float v = (NPL == 2 ? acc[i][j][e] + acx[i][j][e] * (1.0f / 2048.0f) : acc[i][j][e]) + bv;
          if (a.act == ACT_GELU) v = gelu_erf(v);
          if (res) {
            int rr = row;
            if (a.res_div != 1) rr /= a.res_div;
            if (a.res_mod != 0x7fffffff) rr %= a.res_mod;
            v += a.res_scale * res[(long long)rr * a.ldres + col];
          }
          if (a.act == ACT_RELU_POST) v = fmaxf(v, 0.f);
          if (Y) Y[(long long)row * a.ldy + col] = v;
          if (Yh) {
            const float c = __builtin_amdgcn_fmed3f(v, -65000.f, 65000.f);
            const _Float16 hh = (_Float16)c;
            Yh[(long long)row * a.ldys + col] = hh;
            if (a.ys_plane) Yl[(long long)row * a.ldys + col] = (_Float16)((c - (float)hh) * 2048.f);
          }
        }
      }
    }
  }
#endif
}

template <int BM, int BN, int WM, int WN, int STAGES, bool CONV = false, bool FULL = false, int TOPK = 0, int NPL = 2>
static int launch_ss(const GemmSSDev& d, int batch, hipStream_t stream) {
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, batch);
  static const std::string name = std::string(CONV ? "conv_s" : TOPK ? "gemm_ss_topk" + std::to_string(TOPK) : "gemm_s") + (TOPK ? "" : NPL == 2 ? "s" : "h") + (FULL ? "_full<" : "<") + std::to_string(BM) + "," + std::to_string(BN) + "," + std::to_string(WM) +
                                  "," + std::to_string(WN) + "," + std::to_string(STAGES) + ">";
  const double mnk = (double)d.M * d.N * d.K * batch;
  static const bool by_shape = getenv("LBM_PROF_SHAPES") != nullptr;      // profiling aid: one entry per problem shape
  std::string shaped;
  if (by_shape) shaped = name + " " + std::to_string(d.M) + "x" + std::to_string(d.N) + "x" + std::to_string(d.K) + "x" + std::to_string(batch) +
                         (d.Y ? " f32" : "") + (d.Ys ? " split" : "");
  GemmSSDev dv = d;
  dv.vec_ok = vec_ok_of(d);
  // XCD-contiguous tile order where it pays: many tiles, and a B operand small enough (<= 2 MB of split planes) to sit in every
  // XCD's L2 while the row panels of A stream through ONE L2 each instead of all eight (measured 3-8 % on 16384..1044480-row
  // products, scripts/gemm_big.py).  A wide B (the 12288-column correlation product) would thrash instead: dispatch order
  // keeps one column stripe per XCD there.  Debug flag 32 switches it off.
  dv.xcd_remap = !(g_dbg16 & 32) && !CONV && ((long long)grid.x * grid.y) % 8 == 0 && (long long)grid.x * grid.y >= 512 &&
                 (long long)d.N * d.K * 4 <= (2ll << 20) ? 1 : 0;
  ProfScope prof(by_shape ? intern_name(shaped.c_str()) : name.c_str(), stream, 2.0 * mnk,
                 4.0 * ((double)d.M * d.K * batch + (double)d.N * d.K * (d.sB || batch == 1 ? batch : 1) +
                        (double)d.M * d.N * batch * ((d.Y ? 1 : 0) + (d.Ys ? 1 : 0) + (d.res ? 1 : 0))));
  hipLaunchKernelGGL((gemm_ss_kernel<BM, BN, WM, WN, STAGES, CONV, FULL, TOPK, NPL>), grid, dim3(WM * WN * 64), 0, stream, dv);
  return check_launch(CONV ? "conv_ss" : "gemm_ss");
}

// 64x64 tiles with the whole K extent prefetched (K / 32 chunks resident): the variant for grids of <= ~2 workgroups per CU
template <bool CONV, int NPL = 2>
static int launch_full(const GemmSSDev& d, int batch, hipStream_t stream, bool* handled) {
  *handled = true;
  switch (d.K / 32) {
    case 2: return launch_ss<64, 64, 2, 2, 2, CONV, true, 0, NPL>(d, batch, stream);
    case 4: return launch_ss<64, 64, 2, 2, 4, CONV, true, 0, NPL>(d, batch, stream);
    case 6: return launch_ss<64, 64, 2, 2, 6, CONV, true, 0, NPL>(d, batch, stream);
    case 8: return launch_ss<64, 64, 2, 2, 8, CONV, true, 0, NPL>(d, batch, stream);
    case 9: return launch_ss<64, 64, 2, 2, 9, CONV, true, 0, NPL>(d, batch, stream);
    default: break;
  }
  *handled = false;
  return 0;
}
static bool full_ok(int nk) { return nk == 2 || nk == 4 || nk == 6 || nk == 8 || nk == 9; }

// out = relu?(sum of K-split slabs + bias + res) as fp32 and / or split planes (the epilogue of a K-split convolution)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                          const float* __restrict__ bias, const float* __restrict__ res, int N4,
                                                          long long total4, int relu, float* __restrict__ Y, void* Ys,
                                                          long long ys_plane) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    // bias / residual are requested before the slab sums need anything (absent ones re-read slab 0 and are dropped), and
    // the slab loop is unrolled so four slabs are in flight: one dependent round trip per term was the cost of this kernel
    const float4 pb = ld4(bias ? bias + (i % N4) * 4 : slabs + i * 4);
    const float4 pr = ld4(res ? res + i * 4 : slabs + i * 4);
    float4 v = ld4(slabs + i * 4);
#pragma unroll 4
    for (int sl = 1; sl < nslab; ++sl) {
      const float4 p = ld4(slabs + sl * slab_stride + i * 4);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (bias) {
      v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    }
    if (res) {
      v.x += pr.x; v.y += pr.y; v.z += pr.z; v.w += pr.w;
    }
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (Y) st4(Y + i * 4, v);
    if (Ys) st4_split(Ys, ys_plane, i * 4, v);
  }
}

template <int NPL>
static int conv_launch(const ConvSSArgs& c, GemmSSDev& d, int nsplit, int nk, long long big, long long mid, long long M, int K,
                       hipStream_t stream) {
  if (nsplit == 1) {
    d.bias = c.bias; d.res = c.res; d.act = c.relu ? ACT_RELU_POST : ACT_NONE;
    d.Y = c.Y; d.Ys = (_Float16*)c.Ys; d.ys_plane = c.ys_plane; d.kt0_per_z = 0;
    if (mid <= 640 && full_ok(nk)) {
      bool handled;
      const int rc = launch_full<true, NPL>(d, 1, stream, &handled);
      if (handled) return rc;
    }
    if (big >= 128 && c.Cout > 64) return launch_ss<128, 128, 2, 4, 2, true, false, 0, NPL>(d, 1, stream);
    // 64 output channels over many pixels (layer1 of a batch of streams): 256 x 64 tiles stage 40 DMA pieces for 96 MFMAs where
    // 64 x 64 tiles stage 16 for 24 (scripts/conv_bench.py 64: 362 -> 298 us at 786432 x 64 x 576)
    if constexpr (NPL == 2) {
      if (c.Cout <= 64 && M >= 65536) return launch_ss<256, 64, 4, 2, 2, true, false, 0, 2>(d, 1, stream);
    }
    return mid <= 320 ? launch_ss<64, 64, 2, 2, 4, true, false, 0, NPL>(d, 1, stream)
                      : launch_ss<64, 64, 2, 2, 2, true, false, 0, NPL>(d, 1, stream);
  }
  d.Y = (float*)c.scratch; d.sY = M * c.Cout; d.K = K / nsplit; d.kt0_per_z = nk / nsplit; d.act = ACT_NONE;
  bool handled;
  if (launch_full<true, NPL>(d, nsplit, stream, &handled)) return -1;
  if (!handled && launch_ss<64, 64, 2, 2, 2, true, false, 0, NPL>(d, nsplit, stream)) return -1;
  return 0;
}

int conv2d_nhwc_ss(const ConvSSArgs& c, hipStream_t stream) {
  LBM_REQUIRE(c.X && c.Wh && (c.Wl || c.planes == 1) && (c.Y || c.Ys) && c.zeros, "conv2d_ss: null argument");
  LBM_REQUIRE(c.planes == 1 || c.planes == 2, "conv2d_ss: planes must be 1 or 2");
  LBM_REQUIRE(c.B >= 1 && c.H >= 1 && c.Wd >= 1 && c.Cout >= 1, "conv2d_ss: bad shape");
  int lg = 0;
  while ((1 << lg) < c.Cin) ++lg;
  LBM_REQUIRE((1 << lg) == c.Cin && c.Cin >= 32, "conv2d_ss: Cin=%d must be a power of two >= 32", c.Cin);
  LBM_REQUIRE(c.kh >= 1 && c.kw >= 1 && c.stride >= 1 && c.pad >= 0, "conv2d_ss: bad kernel spec");
  const int Ho = (c.H + 2 * c.pad - c.kh) / c.stride + 1, Wo = (c.Wd + 2 * c.pad - c.kw) / c.stride + 1;
  LBM_REQUIRE(Ho >= 1 && Wo >= 1, "conv2d_ss: empty output");
  const int K = c.kh * c.kw * c.Cin;
  LBM_REQUIRE(c.ldw >= K && c.ldw % 8 == 0, "conv2d_ss: packed weight rows need %d columns, have %d", K, c.ldw);
  LBM_REQUIRE(((uintptr_t)c.X & 15) == 0 && ((uintptr_t)c.Wh & 15) == 0 && (c.Wl == nullptr || ((uintptr_t)c.Wl & 15) == 0) && ((uintptr_t)c.zeros & 15) == 0 &&
                  c.x_plane % 8 == 0,
              "conv2d_ss: operands must be 16-byte aligned");
  const long long M = (long long)c.B * Ho * Wo;
  LBM_REQUIRE(M < 0x7fffffffLL, "conv2d_ss: too many output pixels");
  GemmSSDev d{};
  d.A = (const _Float16*)c.X; d.a_plane = c.x_plane; d.B = (const _Float16*)c.Wh;
  d.b_plane = c.Wl ? ((const char*)c.Wl - (const char*)c.Wh) / 2 : 0; d.ldb = c.ldw;
  d.M = (int)M; d.N = c.Cout; d.K = K; d.ldy = c.Cout; d.ldys = c.Cout; d.ldres = c.Cout;
  d.res_div = 1; d.res_mod = 0x7fffffff; d.res_scale = 1.f;
  d.cH = c.H; d.cW = c.Wd; d.cin_log2 = lg; d.ckw = c.kw; d.cstride = c.stride; d.cpad = c.pad; d.cHo = Ho; d.cWo = Wo;
  d.zeros = c.zeros; d.dbg = g_dbg16;
  // few output pixels (one stream, deep layers): split K over gridDim.z into fp32 slabs and reduce them afterwards
  // layer1-shaped convolutions over many pixels: the patch-streaming kernel (debug key 4: 15 = never, 16 = whenever the shape admits it)
  if (g_force_conv != 15 && conv64_rows_ok(c) && (g_force_conv == 16 || g_force_conv == 0 || g_force_conv >= 21)) return conv64_rows(c, stream);
  // layer2 / layer3-shaped convolutions over many pixels: the patch-resident kernel (debug key 4: 21 = never, 22 = whenever the shape admits it)
  if (g_force_conv != 21 && conv_patch_ok(c) && (g_force_conv == 22 || ((g_force_conv == 0 || g_force_conv == 16) && (M / 256) * (c.Cout / 128) >= 256)))
    return conv_patch(c, stream);
  const int nk = K / 32;
  const long long big = ((M + 127) / 128) * ((c.Cout + 127) / 128);
  const long long mid = ((M + 63) / 64) * ((c.Cout + 63) / 64);
  int nsplit = 1;
  if (g_force_conv >= 1 && g_force_conv <= 14) {      // tuning: unsplit pipelined variants
    d.bias = c.bias; d.res = c.res; d.act = c.relu ? ACT_RELU_POST : ACT_NONE;
    d.Y = c.Y; d.Ys = (_Float16*)c.Ys; d.ys_plane = c.ys_plane; d.kt0_per_z = 0;
    switch (g_force_conv) {
      case 1: return launch_ss<64, 64, 2, 2, 3, true>(d, 1, stream);
      case 2: return launch_ss<64, 64, 2, 2, 4, true>(d, 1, stream);
      case 3: return launch_ss<32, 64, 1, 2, 4, true>(d, 1, stream);
      case 4: return launch_ss<32, 64, 1, 2, 6, true>(d, 1, stream);
      case 5: return launch_ss<64, 64, 2, 2, 6, true>(d, 1, stream);
      case 6: return launch_ss<32, 32, 1, 1, 6, true>(d, 1, stream);
      case 7: return launch_ss<128, 64, 2, 2, 2, true>(d, 1, stream);
      case 8: return launch_ss<256, 64, 4, 2, 2, true>(d, 1, stream);
      case 9: return launch_ss<256, 64, 4, 1, 2, true>(d, 1, stream);
      case 10: return launch_ss<256, 128, 4, 2, 2, true>(d, 1, stream);
      case 11: return launch_ss<128, 256, 2, 4, 2, true>(d, 1, stream);
      case 12: return launch_ss<128, 128, 2, 4, 2, true>(d, 1, stream);
      case 13: return launch_ss<128, 128, 2, 2, 2, true>(d, 1, stream);
      case 14: return launch_ss<64, 64, 2, 2, 2, true>(d, 1, stream);
      default: break;
    }
  }
  // measured (scripts/conv_bench.py, one 384x512 frame): with >= 128 tiles the unsplit 4-stage pipeline beats K-split +
  // slab reduction (17 vs 22 us at 96x128x64 -> 64), and a K of <= 4 chunks is never worth splitting (7.5 vs 11 us)
  const bool keep_whole = nk <= 4 || (mid >= 128 && !full_ok(nk));
  if (c.scratch != nullptr && mid <= 320 && !keep_whole) {
    // few output pixels: cut K into chunks of at most 9 x 32 that the full-prefetch kernel keeps resident in LDS
    for (int cand = 1; cand <= 16; ++cand) {
      if (nk % cand != 0 || (unsigned long long)cand * M * c.Cout * sizeof(float) > c.scratch_bytes) continue;
      if (full_ok(nk / cand)) {
        nsplit = cand;
        if (mid * cand >= 192) break;
      }
    }
    if (!full_ok(nk / nsplit)) nsplit = 1;
  }
  if (c.planes == 1) {
    if (conv_launch<1>(c, d, nsplit, nk, big, mid, M, K, stream)) return -1;
  } else if (conv_launch<2>(c, d, nsplit, nk, big, mid, M, K, stream)) {
    return -1;
  }
  if (nsplit == 1) return 0;
  const long long total4 = M * c.Cout / 4;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  ProfScope prof("slab_reduce", stream, 0.0, 4.0 * M * c.Cout * (nsplit + 2.0));
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)c.scratch, nsplit, M * c.Cout, c.bias,
                     c.res, c.Cout / 4, total4, c.relu, c.Y, c.Ys, c.ys_plane);
  return check_launch("slab_reduce");
}

bool gemm_ss_topk_ok(int M, int N, int K, int batch, int k) {
  // the shapes for which the plain product would run the 128 x 128 tiling anyway
  const long long big = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  return k >= 1 && k <= 9 && N % 4 == 0 && N > 64 && big >= 128 && ((long long)((N + 127) / 128) * k) % 4 == 0 && !(g_dbg16 & 128);
}

int gemm_ss(const GemmSSArgs& a, hipStream_t stream) {
  LBM_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "gemm_ss: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.M == 0) return 0;
  LBM_REQUIRE(a.A && a.B && (a.Y || a.Ys || a.topk_k > 0), "gemm_ss: null operand");
  LBM_REQUIRE(a.K % 32 == 0, "gemm_ss: K=%d must be a multiple of 32", a.K);
  LBM_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.lda >= a.K && a.ldb >= a.K, "gemm_ss: lda=%d / ldb=%d must be multiples of 8 and >= K", a.lda, a.ldb);
  LBM_REQUIRE(a.a_plane % 8 == 0 && a.b_plane % 8 == 0 && a.sA % 8 == 0 && a.sB % 8 == 0, "gemm_ss: plane / batch strides must keep 16-byte alignment");
  LBM_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0, "gemm_ss: operands must be 16-byte aligned");
  LBM_REQUIRE(a.Y == nullptr || a.ldy >= a.N, "gemm_ss: ldy too small");
  LBM_REQUIRE(a.Ys == nullptr || a.ldys >= a.N, "gemm_ss: ldys too small");
  LBM_REQUIRE(a.res == nullptr || (a.res_div >= 1 && a.res_mod >= 1 && a.ldres >= a.N), "gemm_ss: bad residual spec");
  LBM_REQUIRE(a.batch >= 1 && a.batch <= 65535, "gemm_ss: batch=%d out of range", a.batch);
  GemmSSDev d{};
  d.A = (const _Float16*)a.A; d.B = (const _Float16*)a.B; d.a_plane = a.a_plane; d.b_plane = a.b_plane;
  d.bias = a.bias; d.Y = a.Y; d.Ys = (_Float16*)a.Ys; d.ys_plane = a.ys_plane; d.res = a.res;
  d.M = a.M; d.N = a.N; d.K = a.K; d.lda = a.lda; d.ldb = a.ldb; d.ldy = a.ldy; d.ldys = a.ldys; d.ldres = a.ldres;
  d.res_div = a.res_div; d.res_mod = a.res_mod; d.res_scale = a.res_scale; d.act = a.act;
  d.sA = a.sA; d.sB = a.sB; d.sBias = a.sBias; d.sY = a.sY; d.sYs = a.sYs; d.sR = a.sR;
  d.dbg = g_dbg16;
  if (a.topk_k > 0) {
    LBM_REQUIRE(a.cand_v && a.cand_i && gemm_ss_topk_ok(a.M, a.N, a.K, a.batch, a.topk_k) && !a.bias && !a.res && a.act == ACT_NONE,
                "gemm_ss: fused top-k needs candidate buffers, k <= 9, N %% 4 == 0 and a plain product");
    d.cand_v = a.cand_v; d.cand_i = a.cand_i; d.topk_k = a.topk_k; d.Y = nullptr; d.Ys = nullptr;
    if (a.topk_k == 1) return launch_ss<128, 128, 2, 4, 2, false, false, 1>(d, a.batch, stream);
    if (a.topk_k <= 4) return launch_ss<128, 128, 2, 4, 2, false, false, 4>(d, a.batch, stream);
    return launch_ss<128, 128, 2, 4, 2, false, false, 9>(d, a.batch, stream);
  }
  const long long big = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
  int t = g_force_tile16;
  // K = 256 and many rows: the row-streaming kernel (weights stationary in registers), gemm_rs.hip.  Debug flags: 256 = never,
  // 512 = whenever the shape admits it.
  // rows-stationary form (gemm_rs EPI 2): built and measured in round 3, bit-identical, SLOWER on the value projection it was meant
  // for (3.44 against 3.03 ms at 1 044 480 x 1536 x 256: with lanes = rows every store instruction touches 32 different lines) --
  // reachable only with debug flag 1024
  if (t == 0 && (g_dbg16 & 1024) && gemm_rs_rows_ok(a)) return gemm_rs_rows(a, (g_dbg16 >> 12) & 0xffff, stream);
  if (t == 0 && !(g_dbg16 & 256) && gemm_rs_ok(a) && ((g_dbg16 & 512) || a.M >= g_rs_min_rows))
    return gemm_rs(a, (g_dbg16 >> 12) & 0xffff, stream);
  const long long mid = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64) * a.batch;
  LBM_REQUIRE(a.planes == 1 || a.planes == 2, "gemm_ss: planes must be 1 or 2");
  if (a.planes == 1) {           // hi planes only: the same three tilings, one MFMA per block
    if (mid <= 640 && full_ok(a.K / 32)) {
      bool handled;
      const int rc = launch_full<false, 1>(d, a.batch, stream, &handled);
      if (handled) return rc;
    }
    return (big >= 128 && a.N > 64) ? launch_ss<128, 128, 2, 4, 2, false, false, 0, 1>(d, a.batch, stream)
                                    : launch_ss<64, 64, 2, 2, 2, false, false, 0, 1>(d, a.batch, stream);
  }
  if ((t == 0 && mid <= 640 && full_ok(a.K / 32)) || t == 20) {
    bool handled;
    const int rc = launch_full<false>(d, a.batch, stream, &handled);
    if (handled) return rc;
  }
  if (t == 0 || t == 20) t = (big >= 128 && a.N > 64) ? 3 : 2;   // measured: scripts/gemm_bench.py
  switch (t) {
    case 1: return launch_ss<128, 128, 2, 2, 2>(d, a.batch, stream);
    case 2: return launch_ss<64, 64, 2, 2, 2>(d, a.batch, stream);
    case 3: return launch_ss<128, 128, 2, 4, 2>(d, a.batch, stream);
    case 4: return launch_ss<128, 128, 2, 2, 3>(d, a.batch, stream);
    case 5: return launch_ss<64, 64, 2, 2, 3>(d, a.batch, stream);
    case 6: return launch_ss<128, 64, 2, 2, 2>(d, a.batch, stream);
    case 7: return launch_ss<64, 128, 2, 2, 2>(d, a.batch, stream);
    case 8: return launch_ss<128, 128, 2, 4, 3>(d, a.batch, stream);
    case 9: return launch_ss<64, 64, 2, 2, 4>(d, a.batch, stream);
    case 10: return launch_ss<128, 128, 2, 4, 4>(d, a.batch, stream);
    case 11: return launch_ss<128, 256, 2, 4, 3>(d, a.batch, stream);
    case 12: return launch_ss<256, 128, 4, 2, 3>(d, a.batch, stream);
    case 13: return launch_ss<128, 64, 2, 2, 4>(d, a.batch, stream);
    case 14: return launch_ss<64, 128, 2, 2, 4>(d, a.batch, stream);
    case 15: return launch_ss<128, 64, 4, 2, 4>(d, a.batch, stream);
    case 16: return launch_ss<64, 256, 2, 4, 3>(d, a.batch, stream);
    case 17: return launch_ss<128, 64, 2, 2, 3>(d, a.batch, stream);
    case 18: return launch_ss<128, 64, 4, 2, 3>(d, a.batch, stream);
    case 19: return launch_ss<64, 128, 2, 4, 3>(d, a.batch, stream);
    case 21: return launch_ss<128, 128, 4, 2, 2>(d, a.batch, stream);
    case 22: return launch_ss<256, 128, 4, 2, 2>(d, a.batch, stream);
    case 23: return launch_ss<128, 256, 2, 4, 2>(d, a.batch, stream);
    default: break;
  }
  set_error("gemm_ss: unknown tiling %d", t);
  return -1;
}

}  // namespace lbm
