// stem_pool: the ResNet stem in one launch -- 7 x 7 stride-2 convolution 3(+1 zero) -> 64 channels with folded BatchNorm, ReLU, and
// the 3 x 3 stride-2 max-pooling behind it (backbone.py:48-50 via torchvision's conv1 / bn1 / relu / maxpool), split-precision.
//
// Separately the two cost 669 + 247 us for 64 frames of 384 x 512: the convolution splits its fp32 input in the loader (6 vector
// instructions per operand value, 49 x per input value) and writes 1.7 GB of fp32 that the pooling kernel reads straight back.  Here
//   * the input arrives as f16 hi / lo planes [B, H, W, 4] (written by the preprocess kernel): operand fragments are plain
//     ds_read_b128 out of a patch in LDS -- with the taps of a row padded on the LEFT to eight (tap 0 has zero weights) a k-step of 16
//     is four taps x four channels of ONE input row, the two taps a lane needs are 16 contiguous, 16-byte-aligned bytes
//     (pixel 2 ox + tap, 8 B per pixel and plane), and the 32 lanes of a half-wave read consecutive 16-byte slots: no bank conflict,
//     no swizzle;
//   * the weights (64 x 224 padded, both planes) stay in registers: 112 VGPRs per multiplying wave, whole K in one wave;
//   * the conv output never leaves the CU: a workgroup walks DOWN a strip of 32 conv columns (30 new ones + the two the pooling windows
//     share with the neighbours: 15 pooled columns per strip), two conv rows per iteration, the epilogue waves keep the running
//     vertical maximum in registers and write only pooled rows (fp32 and split planes).
// Workgroup = 8 waves, as conv64_rows: 4 multiply (column block x row parity), 2 issue the LDS-DMA of the patches (one per plane, 9
// pieces of 35 lanes per iteration, hand-counted vmcnt), 2 pool and store.  One barrier per iteration.
//
// Values after the ReLU are >= 0 and every pooling window holds at least one conv output, so conv rows / columns outside the image
// (the pooling's padding, and the columns a last strip computes past the image) enter the maximum as 0.
#include "gemm.hpp"

#include <type_traits>

namespace lbm {

struct StemDev {
  const _Float16* X;        // [2 planes][B][H][W][4]
  long long x_plane;
  const _Float16* Wh;       // [64][ldw], columns (ky, kx, c) with c < 4
  const _Float16* Wl;
  int ldw;
  const float* bias;
  float* Y;                 // [B][Hp][Wp][64] or null
  _Float16* Ys;             // split twin or null
  long long ys_plane;
  int B, H, W, Hp, Wp;
  int rows_per_wg, chunks, strips;
  int dbg;   // tuning (timing only, debug key 20): 1 no DMA after the prologue, 2 no pooling / stores, 4 no MFMAs, 8 no hand-over writes
};

int g_stem_dbg = 0;

typedef __attribute__((address_space(3))) void* lds_ptr_st;

[[maybe_unused]] static __device__ __align__(256) unsigned char g_stem_zero_row[640];      // rows above / below the image, pixels left / right of it

__global__ __launch_bounds__(512) void stem_pool_kernel(StemDev a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWB = 35 * 16;             // one input row of the patch: 70 pixels x 4 channels f16
  constexpr int PLANE = 9 * ROWB;           // 9 input rows feed two conv rows
  constexpr int PATCH = 2 * PLANE + 160;    // 10240
  constexpr int NST = 4;
  constexpr int TP = 36;
  constexpr int FIN = 32 * TP * 4;          // a finished conv row piece: 32 columns x 32 channels fp32, pitch TP
  constexpr int OFF_FIN = NST * PATCH;      // [2 buffers][2 row parities][2 column blocks][FIN]
  constexpr int OFF_POOL = OFF_FIN + 8 * FIN;   // [2 column blocks][FIN]: the epilogue waves' private rows
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OFF_POOL + 2 * FIN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int role = wave >> 1, cb = wave & 1;            // 0 / 1: multiply conv row A / B, 2: DMA (cb = plane), 3: pool + store
  const int r = lane & 31, h = lane >> 5;

  int id = blockIdx.x;
  const int chunk = id % a.chunks; id /= a.chunks;
  const int strip = id % a.strips;
  const int b = id / a.strips;
  const int py0 = chunk * a.rows_per_wg;
  const int n = min(a.rows_per_wg, a.Hp - py0);        // pooled rows of this workgroup; iterations 0 .. n multiply, 1 .. n + 1 pool
  const int Wc = a.W >> 1;
  const int ox0 = 30 * strip - 1;                      // first conv column of the strip (-1: the pooling's left padding)

  auto handover = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (role < 2) {
    const int rp = role;
    // ---- weights as MFMA B-fragments.  k-step ks = (ky, half of the padded tap row): lane (r, h) holds column 32 cb + r, padded taps
    // kx' = 4 (ks & 1) + 2 h and + 1, four channels each; padded tap kx' is tap kx' - 1, tap -1 has zero weights ----
    half8 wh[14], wl[14];
    {
      const long long wo = (long long)(cb * 32 + r) * a.ldw;
#pragma unroll
      for (int ks = 0; ks < 14; ++ks) {
        const int ky = ks >> 1, kx0 = 4 * (ks & 1) + 2 * h - 1;       // tap of the fragment's first four values (-1: zero)
        half4 h0 = {0, 0, 0, 0}, l0 = {0, 0, 0, 0};
        if (kx0 >= 0) {
          h0 = *reinterpret_cast<const half4*>(a.Wh + wo + (ky * 7 + kx0) * 4);
          l0 = *reinterpret_cast<const half4*>(a.Wl + wo + (ky * 7 + kx0) * 4);
        }
        const half4 h1 = *reinterpret_cast<const half4*>(a.Wh + wo + (ky * 7 + kx0 + 1) * 4);
        const half4 l1 = *reinterpret_cast<const half4*>(a.Wl + wo + (ky * 7 + kx0 + 1) * 4);
        wh[ks] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        wl[ks] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      }
    }
    const float bv = a.bias ? a.bias[cb * 32 + r] : 0.f;
    const int fo = 16 * r + 16 * h;          // byte offset of this lane's fragment inside an input row (+ 32 for the second half of the taps)
    for (int t = 0; t <= n + 1; ++t) {
      handover();
      if (t > n || (t == n && rp == 1) || (LBM_ABL(a.dbg & 4))) continue;      // (conv row B of the last iteration belongs to the next workgroup)
      const unsigned char* Pb = lds + (t & (NST - 1)) * PATCH + (2 * rp) * ROWB + fo;
      f32x16 acc, acx;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = acx[e] = 0.f;
      half8 fa[2][4];
      auto ldg = [&](const int bsel, const int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ks = 2 * g + q;
          const unsigned char* p = Pb + (ks >> 1) * ROWB + (ks & 1) * 32;
          fa[bsel][2 * q] = *reinterpret_cast<const half8*>(p);
          fa[bsel][2 * q + 1] = *reinterpret_cast<const half8*>(p + PLANE);
        }
      };
      ldg(0, 0);
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        if (g + 1 < 7) ldg((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ks = 2 * g + q;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], wh[ks], acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], wl[ks], acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q + 1], wh[ks], acx, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float* Tw = reinterpret_cast<float*>(lds + OFF_FIN + (((t & 1) * 2 + rp) * 2 + cb) * FIN);
      if (!(LBM_ABL(a.dbg & 8)))
#pragma unroll
        for (int e = 0; e < 16; ++e) Tw[(4 * h + (e & 3) + 8 * (e >> 2)) * TP + r] = fmaxf((acc[e] + acx[e] * (1.0f / 2048.0f)) + bv, 0.f);
    }
  } else if (role == 2) {
    // ---- DMA: iteration t needs input rows 4 (py0 + t) - 5 .. + 8, pixels 60 strip - 6 .. + 69 (two per lane, 35 lanes) ----
    const int x0 = 60 * strip - 6;
    const bool xin = (unsigned)(x0 + 2 * lane) < (unsigned)a.W;       // (W even: a 16-byte unit is inside or outside as a whole)
    const _Float16* zrow = reinterpret_cast<const _Float16*>(g_stem_zero_row) + lane * 8;
    const _Float16* Xb = a.X + cb * a.x_plane + ((long long)b * a.H * a.W + x0 + 2 * lane) * 4;
    auto issue = [&](const int t) __attribute__((always_inline)) {
      unsigned char* stage = lds + (t & (NST - 1)) * PATCH + cb * PLANE;
      const int y0 = 4 * (py0 + t) - 5;
#pragma unroll
      for (int pr = 0; pr < 9; ++pr) {
        const int y = y0 + pr;
        const bool yin = (unsigned)y < (unsigned)a.H;
        const _Float16* src = (yin && xin) ? Xb + (long long)y * a.W * 4 : zrow;
        if (lane < 35) __builtin_amdgcn_global_load_lds(src, (lds_ptr_st)(stage + pr * ROWB), 16, 0, 0);
      }
    };
    issue(0);
    if (n >= 1) issue(1);
    for (int t = 0; t <= n + 1; ++t) {
      if (t + 1 <= n) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // patch t landed, patch t + 1 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      handover();
      if (t + 2 <= n && !(LBM_ABL(a.dbg & 1))) issue(t + 2);            // its stage held patch t - 2: last read in iteration t - 2
    }
  } else {
    // ---- pooling.  Iteration t takes conv rows A = 2 (py0 + k) - 1 and B = 2 (py0 + k) of iteration k = t - 1: pooled row
    // py0 + k - 1 = max(carry, A) goes out, carry = max(A, B).  Lane: conv columns 8 i + (lane >> 3), channels 4 (lane & 7) .. + 4 ----
    const int lrow = lane >> 3, c4 = lane & 7;
    float* Pw = reinterpret_cast<float*>(lds + OFF_POOL + cb * FIN);
    float4 carry[4];
    bool colv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      carry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      colv[i] = (unsigned)(ox0 + i * 8 + lrow) < (unsigned)Wc;
    }
    auto mx = [](const float4 p, const float4 q) { return make_float4(fmaxf(p.x, q.x), fmaxf(p.y, q.y), fmaxf(p.z, q.z), fmaxf(p.w, q.w)); };
    for (int t = 0; t <= n + 1; ++t) {
      handover();
      if (t == 0 || (LBM_ABL(a.dbg & 2))) continue;
      const int k = t - 1;
      const float* Ta = reinterpret_cast<const float*>(lds + OFF_FIN + (((k & 1) * 2 + 0) * 2 + cb) * FIN);
      const float* Tb = reinterpret_cast<const float*>(lds + OFF_FIN + (((k & 1) * 2 + 1) * 2 + cb) * FIN);
      float4 va[4], vb[4];
      const bool a_in = py0 + k > 0;          // conv row -1 is the pooling's top padding
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = *reinterpret_cast<const float4*>(Ta + (i * 8 + lrow) * TP + c4 * 4);
        if (k < n) vb[i] = *reinterpret_cast<const float4*>(Tb + (i * 8 + lrow) * TP + c4 * 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!a_in || !colv[i]) va[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!colv[i]) vb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (k >= 1) {
        // vertical maximum of the three conv rows -> this wave's private row; then the horizontal windows 2 j, 2 j + 1, 2 j + 2
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(Pw + (i * 8 + lrow) * TP + c4 * 4) = mx(carry[i], va[i]);
        const int py = py0 + k - 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = jj * 8 + lrow;
          const int px = 15 * strip + j;
          if (j < 15 && px < a.Wp) {
            const float* q = Pw + (2 * j) * TP + c4 * 4;
            const float4 v = mx(mx(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(q + TP)),
                                *reinterpret_cast<const float4*>(q + 2 * TP));
            const long long o = (((long long)b * a.Hp + py) * a.Wp + px) * 64 + cb * 32 + c4 * 4;
            if (a.Y) st4(a.Y + o, v);
            if (a.Ys) st4_split(a.Ys, a.ys_plane, o, v);
          }
        }
      }
      if (k < n) {
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[i] = mx(va[i], vb[i]);
      }
    }
  }
#endif
}

bool stem_pool_ok(int H, int W, int ldw) { return H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && ldw >= 196 && ldw % 4 == 0; }

// X: split planes [B, H, W, 4]; W: split planes [64, ldw] with columns (ky, kx, c), c < 4; Y / Ys: [B, H / 4, W / 4, 64]
int stem_pool(const void* X, long long x_plane, const void* Wh, const void* Wl, int ldw, const float* bias, float* Y, void* Ys,
              long long ys_plane, int B, int H, int W, hipStream_t stream) {
  LBM_REQUIRE(X && Wh && Wl && (Y || Ys) && B > 0, "stem_pool: null argument");
  LBM_REQUIRE(stem_pool_ok(H, W, ldw), "stem_pool: H and W must be multiples of 4, weight rows >= 196 columns (7 x 7 x 4)");
  LBM_REQUIRE(((uintptr_t)X & 15) == 0 && x_plane % 8 == 0 && ((uintptr_t)Wh & 7) == 0 && ((uintptr_t)Wl & 7) == 0 && ((uintptr_t)Y & 15) == 0 &&
                  ((uintptr_t)Ys & 7) == 0 && ys_plane % 4 == 0, "stem_pool: operands must be 16-byte aligned");
  StemDev d;
  d.X = (const _Float16*)X; d.x_plane = x_plane; d.Wh = (const _Float16*)Wh; d.Wl = (const _Float16*)Wl; d.ldw = ldw; d.bias = bias;
  d.Y = Y; d.Ys = (_Float16*)Ys; d.ys_plane = ys_plane; d.B = B; d.H = H; d.W = W; d.Hp = H / 4; d.Wp = W / 4;
  d.strips = (d.Wp + 14) / 15;
  d.dbg = g_stem_dbg;
  // pooled rows per workgroup: short workgroups while they are few (one frame), up to 32 rows when there are thousands (each
  // workgroup recomputes one conv row of its upper neighbour and loads the weights once)
  const long long units = (long long)B * d.strips * d.Hp;
  long long rows = (units + 1023) / 1024;
  if (rows < 4) rows = 4;
  if (rows > 32) rows = 32;
  if (rows > d.Hp) rows = d.Hp;
  d.rows_per_wg = (int)rows;
  d.chunks = (d.Hp + d.rows_per_wg - 1) / d.rows_per_wg;
  const long long wgs = (long long)B * d.strips * d.chunks;
  LBM_REQUIRE(wgs <= 0x7fffffffLL, "stem_pool: grid too large");
  const double conv_px = (double)B * (H / 2) * (W / 2);
  ProfScope prof("stem_pool", stream, 2.0 * conv_px * 64 * 196,
                 16.0 * B * H * W + 4.0 * B * d.Hp * d.Wp * 64 * ((Y ? 1 : 0) + (Ys ? 1 : 0)) + 4.0 * 64 * 196);
  hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)wgs), dim3(512), 0, stream, d);
  return check_launch("stem_pool");
}

}  // namespace lbm
