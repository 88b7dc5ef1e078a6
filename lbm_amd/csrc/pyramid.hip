// Feature pyramid, correlation maps and top-k selection.
//
//  * pyramid_pool   — transformer.py:58-69: levels 1..3 = avg_pool2d(level 0, 2/4/8), token-major.
//  * corr_maps      — transformer.py:207-227 restructured.  The reference normalises every level,
//                     takes q̂·f̂_l on each level, bilinearly upsamples the four *score* maps and mixes
//                     them with a 1x1 conv plus the level-0 residual.  All of that is linear in f̂, so
//                       corr[n, p] = q̂_n · F[p] + b,   F[p] = (1 + w0) f̂_0[p] + Σ_{l=1..3} w_l · up_l(f̂_l)[p]
//                     and F (one (P,256) map per layer and stream, independent of the points) is built
//                     here once per frame.  The (N, 4, h, w) score stack of the reference — 45 % of its
//                     CPU time — never exists.
//  * topk_rows      — transformer.py:307-308 + coord_utils.py:4-29: top-k cells of each correlation row
//                     (ties → lowest index) and their cell centres in network pixels.
#include "pyramid.hpp"
#define LBM_CB_NS cb
#include "corr_block.inc"

namespace lbm {
using namespace cb;
int launch_corr_block_packed(const void* args, int mode, unsigned blocks, hipStream_t stream);   // corr_block_pk.hip

// ------------------------------------------------------------------------------------------------
// also used (with `toks`) to emit the split-plane twin of the whole token buffer, level 0 included
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, void* xs, long long plane, int B, int P0, int L) {
  const int lane = threadIdx.x & 63;
  const long long rows = (long long)B * P0;
  for (long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); i < rows; i += (long long)gridDim.x * 4) {
    const long long row = (i / P0) * L + (i % P0);   // level-0 rows of stream i / P0
    st4_split(xs, plane, row * LBM_C + lane * 4, ld4(x + row * LBM_C + lane * 4));
  }
}

// One workgroup per 8x8 block of level-0 cells, one wave per 4x4 quadrant, lane = 4 channels: the 16 level-0 rows of
// a quadrant are loaded once (all loads in flight), giving its four level-1 cells and its level-2 cell in registers;
// the level-3 cell is the mean of the four quadrant sums (through LDS).  With `toks` the split twin of every row
// touched, level 0 included, is emitted on the way.  (The first version walked each output cell's s x s window with
// dependent loads: 40 us at one 96x128 frame, all of it the 64-load chains of the 192 level-3 cells.)
// TAIL: level 0 does not exist yet -- it is formed here as GELU(q1 + up2(q2) + up4(q3)) + pos from the three feature
// levels already multiplied by (fusion o projection) at their own resolution (backbone.py:53-63 with the 1x1
// projections, the bilinear up-sampling and the fusion 1x1 + BatchNorm commuted: all linear), and written out as well.
struct TailArgs {
  const float* q1;    // [B, h*w, 256]   (bias of the folded fusion already added)
  const float* q2;    // [B, (h/2)*(w/2), 256]
  const float* q3;    // [B, (h/4)*(w/4), 256]
  const float* pos;   // [h*w, 256] or null
  // MODE 2: q1 is not read but formed here as f1 . W1^T + b1 (the 64 -> 256 product of the finest level, split operands)
  const _Float16* f1;   // split planes [B, h*w, 64]
  long long f1_plane;
  const _Float16* w1h;  // [256, ldw1]
  const _Float16* w1l;
  int ldw1;
  const float* b1;
  float* inv;           // [B, L] or null: 1 / max(|row|, 1e-12) of every token row written here (what corr_maps would compute)
  int stream_minor;     // workgroup order: 1 = the same 8 x 8 block of every stream back to back (they share its rows of `pos`), 0 = stream by stream
  int dbg;              // tuning build only (debug key 24, bits 4..): 16 = no level-0 product, 32 = no q2 / q3 / pos loads, 64 = no stores, 128 = no GELU
};

// F.normalize's factor for one 256-channel row held by a wave (lane = 4 channels); every rounding spelled out: the encoder tail and
// inv_norm_kernel must give the same bits
__device__ __forceinline__ float row_inv_norm(const float4 x) {
  const float ss = __fmaf_rn(x.w, x.w, __fmaf_rn(x.z, x.z, __fmaf_rn(x.y, x.y, __fmul_rn(x.x, x.x))));
  return __fdiv_rn(1.0f, fmaxf(sqrtf(wave_sum(ss)), 1e-12f));     // F.normalize eps
}

// MODE 0: pooling only; 1: TAIL (q1 from memory); 2: TAIL with q1 = f1 . W1^T + b1 computed by the workgroup (16 waves = 2 x 8
// blocks of 32 cells x 32 channels, 12 MFMAs each, operands straight from memory into registers) and handed over through LDS: the
// [B h w, 256] product is neither written nor read back (0.8 GB each way at 64 frames).  Same MFMA order and final combination as
// gemm_ss: the tokens are bit-identical to MODE 1 fed by that kernel.
// With many blocks MODE 2 runs with NW = 8 waves per workgroup (each wave takes two of the sixteen 2 x 2 groups and two of the sixteen
// MFMA blocks in turn): at ~94 VGPRs a 16-wave workgroup is alone on its CU and its load / multiply / pool / store phases run back to back; two
// 8-wave workgroups per CU overlap them (1027 -> 922 us at 64 frames; one frame has fewer blocks than CUs and keeps 16 waves).
// PATCH (TAIL forms): the bilinear footprints of a 2 x 2 group's four cells overlap almost entirely -- a 3 x 3 patch of q2 rows and
// ONE 2 x 2 patch of q3 rows (a 2-aligned pair of fine cells never straddles a coarse source pair) -- so the group fetches 13 rows
// where cell-by-cell evaluation fetched 32 (the same observation as corr_maps_block_kernel).  Weights stay up_coeff's, per cell;
// positions come from the patch: at the borders up_coeff's clamped pair and the patch pair differ only where the weight is 0, so the
// tokens are the same bits (debug key 24 bit 1 selects the cell-by-cell form for the A/B and the bit-identity test).
template <int MODE, int NW = 16, bool PATCH = true>
__global__ __launch_bounds__(NW * 64) void pyramid_pool_kernel(float* __restrict__ tok, int B, int h, int w, int L, void* toks,
                                                            long long toks_plane, TailArgs ta) {
  constexpr bool TAIL = MODE != 0;
  constexpr int GPW = 16 / NW;       // 2 x 2 groups (and MFMA blocks) per wave
  constexpr int QP = 260;                       // pitch of the q1 block in LDS (floats): rows stay 16-byte aligned
  // (the q1 block and the pooling's partial sums share LDS -- 66.5 KB, two workgroups per CU: with both resident, 82.5 KB, a CU held
  // ONE workgroup and its load / multiply / pool / store phases ran back to back: 986 us instead of 823 + 236 at 64 frames)
  __shared__ __attribute__((aligned(16))) float q1s[MODE == 2 ? 64 * QP : 16 * 64 * 4];
  float4 (*part)[64] = reinterpret_cast<float4 (*)[64]>(q1s);
  // 16 waves per 8x8 block: wave q owns the 2x2 cells at (2 (q / 4), 2 (q % 4)) -- all of its loads are independent and
  // in flight together -- and produces their level-1 cell; the 2x2 sums meet in LDS for levels 2 (waves 0..3) and 3.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int bw = w >> 3, bh = h >> 3;
  const int P0 = h * w;
  // Stream-minor order for a batch of streams: the 64 rows of the positional embedding a block adds are the same for every
  // stream; stream by stream they come back after a whole stream's worth of other rows has passed through the L2.
  const bool sm_order = TAIL && ta.stream_minor;
  const int b = sm_order ? blockIdx.x % B : blockIdx.x / (bh * bw);
  const int rem = sm_order ? blockIdx.x / B : blockIdx.x - b * (bh * bw);
  const int by = rem / bw, bx = rem - by * bw;
  float* base = tok + (long long)b * L * LBM_C + lane * 4;
  const long long sbase = (long long)b * L * LBM_C + lane * 4;
  if constexpr (MODE == 2) {
#if defined(__HIP_DEVICE_COMPILE__)
   // (NW = 8: a wave's two blocks are the two row blocks of ONE column block -- q & 7 == wv for both -- so its weight fragments
   // and bias are fetched once)
   half8 bh[4], bl[4];
   float bv = 0.f;
#pragma unroll 1
   for (int gi = 0; gi < (LBM_ABL(ta.dbg & 16) ? 0 : GPW); ++gi) {
    const int q = wv + gi * NW;
    const int r = lane & 31, hh = lane >> 5, rb = q >> 3, cbk = q & 7;
    const int c = 32 * rb + r;                                          // cell of the 8 x 8 block, row-major
    const long long row = (long long)b * P0 + (long long)(by * 8 + (c >> 3)) * w + bx * 8 + (c & 7);
    const _Float16* ap = ta.f1 + row * 64 + 8 * hh;
    const long long wo = (long long)(32 * cbk + r) * ta.ldw1 + 8 * hh;
    half8 ah[4], al[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ah[ks] = *reinterpret_cast<const half8*>(ap + ks * 16);
      al[ks] = *reinterpret_cast<const half8*>(ap + ta.f1_plane + ks * 16);
    }
    if (gi == 0 || NW != 8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bh[ks] = *reinterpret_cast<const half8*>(ta.w1h + wo + ks * 16);
        bl[ks] = *reinterpret_cast<const half8*>(ta.w1l + wo + ks * 16);
      }
      bv = ta.b1 ? ta.b1[32 * cbk + r] : 0.f;
    }
    f32x16 acc, acx;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acx[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc, 0, 0, 0);
      acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acx, 0, 0, 0);
      acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acx, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) q1s[(32 * rb + 4 * hh + (e & 3) + 8 * (e >> 2)) * QP + 32 * cbk + r] = (acc[e] + acx[e] * (1.0f / 2048.0f)) + bv;
   }
    __syncthreads();
#endif
  }
  auto add = [](float4 a, const float4 c) { a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; return a; };
  auto mul = [](const float4 a, float f) { return make_float4(a.x * f, a.y * f, a.z * f, a.w * f); };
  auto put = [&](long long cell, const float4 o) {
    if (LBM_ABL(ta.dbg & 64) && o.x != 12345.678f) return;
    st4(base + cell * LBM_C, o);
    if (toks) st4_split(toks, toks_plane, sbase + cell * LBM_C, o);
    if constexpr (TAIL) {
      if (ta.inv) {
        const float sn = row_inv_norm(o);
        if (lane == 0) ta.inv[(long long)b * L + cell] = sn;
      }
    }
  };
  const int w1 = w >> 1, w2l = w >> 2, w3 = w >> 3;
  float4 s4g[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
  const int q = wv + gi * NW;
  const int y0 = by * 8 + (q >> 2) * 2, x0 = bx * 8 + (q & 3) * 2;
  float4 v[2][2];
  if constexpr (!TAIL) {
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) v[dy][dx] = ld4(base + (long long)((y0 + dy) * w + x0 + dx) * LBM_C);
  } else {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    const float* q1 = ta.q1 + (long long)b * P0 * LBM_C + lane * 4;
    const float* q2 = ta.q2 + (long long)b * h2 * w2 * LBM_C + lane * 4;
    const float* q3 = ta.q3 + (long long)b * h4 * w4 * LBM_C + lane * 4;
    auto bil = [](const float* qq, int wl, int ya, int yb, int xa, int xb, float lya, float lyb, float lxa, float lxb) {
      const float4 v00 = ld4(qq + (long long)(ya * wl + xa) * LBM_C), v01 = ld4(qq + (long long)(ya * wl + xb) * LBM_C);
      const float4 v10 = ld4(qq + (long long)(yb * wl + xa) * LBM_C), v11 = ld4(qq + (long long)(yb * wl + xb) * LBM_C);
      // every rounding spelled out: the cell-by-cell and the patch form must give the same bits, whatever hipcc would contract
      auto one = [&](float a00, float a01, float a10, float a11) {
        const float top = __fmaf_rn(lxb, a01, __fmul_rn(lxa, a00)), bot = __fmaf_rn(lxb, a11, __fmul_rn(lxa, a10));
        return __fmaf_rn(lyb, bot, __fmul_rn(lya, top));
      };
      return make_float4(one(v00.x, v01.x, v10.x, v11.x), one(v00.y, v01.y, v10.y, v11.y), one(v00.z, v01.z, v10.z, v11.z),
                         one(v00.w, v01.w, v10.w, v11.w));
    };
    // PATCH: every row of the group requested before anything is used
    float4 p2[3][3], p3[2][2];
    if (LBM_ABL(ta.dbg & 32)) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v2 = 0; v2 < 3; ++v2) p2[u][v2] = make_float4(0.1f * lane, 0.2f, 0.3f, 0.4f);
      p3[0][0] = p3[0][1] = p3[1][0] = p3[1][1] = make_float4(0.5f, 0.1f * lane, 0.7f, 0.8f);
    } else if constexpr (PATCH) {
      const int iy = y0 >> 1, ix = x0 >> 1;
      const int ry[3] = {max(iy - 1, 0), iy, min(iy + 1, h2 - 1)};
      const int rx[3] = {max(ix - 1, 0), ix, min(ix + 1, w2 - 1)};
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v2 = 0; v2 < 3; ++v2) p2[u][v2] = ld4(q2 + (long long)(ry[u] * w2 + rx[v2]) * LBM_C);
      int yc, yd, xc, xd;
      float t0, t1;
      up_coeff(y0, 2, h4, yc, yd, t0, t1);
      up_coeff(x0, 2, w4, xc, xd, t0, t1);
      p3[0][0] = ld4(q3 + (long long)(yc * w4 + xc) * LBM_C);
      p3[0][1] = ld4(q3 + (long long)(yc * w4 + xd) * LBM_C);
      p3[1][0] = ld4(q3 + (long long)(yd * w4 + xc) * LBM_C);
      p3[1][1] = ld4(q3 + (long long)(yd * w4 + xd) * LBM_C);
    }
    auto bilv = [](const float4& v00, const float4& v01, const float4& v10, const float4& v11, float lya, float lyb, float lxa, float lxb) {
      // every rounding spelled out: the cell-by-cell and the patch form must give the same bits, whatever hipcc would contract
      auto one = [&](float a00, float a01, float a10, float a11) {
        const float top = __fmaf_rn(lxb, a01, __fmul_rn(lxa, a00)), bot = __fmaf_rn(lxb, a11, __fmul_rn(lxa, a10));
        return __fmaf_rn(lyb, bot, __fmul_rn(lya, top));
      };
      return make_float4(one(v00.x, v01.x, v10.x, v11.x), one(v00.y, v01.y, v10.y, v11.y), one(v00.z, v01.z, v10.z, v11.z),
                         one(v00.w, v01.w, v10.w, v11.w));
    };
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      int ya, yb, yc, yd;
      float lya, lyb, lyc, lyd;
      up_coeff(y0 + dy, 1, h2, ya, yb, lya, lyb);
      up_coeff(y0 + dy, 2, h4, yc, yd, lyc, lyd);
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        int xa, xb, xc, xd;
        float lxa, lxb, lxc, lxd;
        up_coeff(x0 + dx, 1, w2, xa, xb, lxa, lxb);
        up_coeff(x0 + dx, 2, w4, xc, xd, lxc, lxd);
        const long long cell = (long long)(y0 + dy) * w + x0 + dx;
        float4 a1, a2, a3;
        if constexpr (MODE == 2) a1 = *reinterpret_cast<const float4*>(q1s + ((y0 + dy - by * 8) * 8 + (x0 + dx - bx * 8)) * QP + lane * 4);
        else a1 = ld4(q1 + cell * LBM_C);
        if constexpr (PATCH) {
          // level 1: row y0 + dy reads patch rows (dy, dy + 1), column x0 + dx patch columns (dx, dx + 1)
          a2 = bilv(p2[dy][dx], p2[dy][dx + 1], p2[dy + 1][dx], p2[dy + 1][dx + 1], lya, lyb, lxa, lxb);
          a3 = bilv(p3[0][0], p3[0][1], p3[1][0], p3[1][1], lyc, lyd, lxc, lxd);
        } else {
          a2 = bil(q2, w2, ya, yb, xa, xb, lya, lyb, lxa, lxb);
          a3 = bil(q3, w4, yc, yd, xc, xd, lyc, lyd, lxc, lxd);
        }
        float4 t = make_float4(a1.x + a2.x + a3.x, a1.y + a2.y + a3.y, a1.z + a2.z + a3.z, a1.w + a2.w + a3.w);
        if (!(LBM_ABL(ta.dbg & 128))) t = make_float4(gelu_erf(t.x), gelu_erf(t.y), gelu_erf(t.z), gelu_erf(t.w));
        if (ta.pos && !(LBM_ABL(ta.dbg & 32))) {
          const float4 p = ld4(ta.pos + cell * LBM_C + lane * 4);
          t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        v[dy][dx] = t;
        if (LBM_ABL(ta.dbg & 64) && t.x != 12345.678f) continue;
        st4(base + cell * LBM_C, t);
        if (ta.inv) {
          const float sn = row_inv_norm(t);
          if (lane == 0) ta.inv[(long long)b * L + cell] = sn;
        }
      }
    }
  }
  if (toks && !(LBM_ABL(ta.dbg & 64) && v[0][0].x != 12345.678f)) {
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) st4_split(toks, toks_plane, sbase + (long long)((y0 + dy) * w + x0 + dx) * LBM_C, v[dy][dx]);
  }
  // level 1: this group's 2x2 window, row-major
  const float4 s4 = add(add(add(v[0][0], v[0][1]), v[1][0]), v[1][1]);
  put(P0 + (long long)(y0 >> 1) * w1 + (x0 >> 1), mul(s4, 0.25f));
  s4g[gi] = s4;
  }
  if constexpr (MODE == 2) __syncthreads();      // every wave has read its cells of the q1 block: the space is the partial sums' now
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) part[wv + gi * NW][lane] = s4g[gi];
  __syncthreads();
  const int q = wv;
  if (q < 4) {
    // level 2: quadrant q = (q / 2, q % 2) of the block, the four 2x2 sums inside it
    const int qy = q >> 1, qx = q & 1;
    const int w0 = (qy * 2) * 4 + qx * 2;            // wave holding its top-left 2x2
    const float4 s16 = add(add(add(part[w0][lane], part[w0 + 1][lane]), part[w0 + 4][lane]), part[w0 + 5][lane]);
    put(P0 + (P0 >> 2) + (long long)(by * 2 + qy) * w2l + bx * 2 + qx, mul(s16, 1.0f / 16.0f));
    if (q == 0) {
      float4 s64 = part[0][lane];
#pragma unroll
      for (int t = 1; t < 16; ++t) s64 = add(s64, part[t][lane]);
      put(P0 + (P0 >> 2) + (P0 >> 4) + (long long)by * w3 + bx, mul(s64, 1.0f / 64.0f));
    }
  }
}

int pyramid_pool(float* tokens, int B, int h, int w, hipStream_t stream, void* toks, long long toks_plane) {
  LBM_REQUIRE(tokens && B > 0 && h % 8 == 0 && w % 8 == 0, "pyramid_pool: grid %dx%d must be a multiple of 8", h, w);
  const int P0 = h * w;
  const int L = P0 + P0 / 4 + P0 / 16 + P0 / 64;
  const long long blocks = (long long)B * (h / 8) * (w / 8);
  LBM_REQUIRE(blocks < 0x7fffffffLL, "pyramid_pool: too many blocks");
  ProfScope prof("pyramid_pool", stream, 0.0, 4.0 * LBM_C * B * ((double)P0 + (L - P0)) * (toks ? 2.0 : 1.0));
  hipLaunchKernelGGL(pyramid_pool_kernel<0>, dim3((unsigned)blocks), dim3(1024), 0, stream, tokens, B, h, w, L, toks, toks_plane,
                     TailArgs{});
  return check_launch("pyramid_pool");
}

extern int g_tail_order;
int encoder_tail(const float* q1, const float* q2, const float* q3, const float* pos, float* tokens, int B, int h, int w,
                 hipStream_t stream, void* toks, long long toks_plane) {
  LBM_REQUIRE(q1 && q2 && q3 && tokens && B > 0 && h % 8 == 0 && w % 8 == 0, "encoder_tail: bad arguments (grid %dx%d)", h, w);
  const int P0 = h * w;
  const int L = P0 + P0 / 4 + P0 / 16 + P0 / 64;
  const long long blocks = (long long)B * (h / 8) * (w / 8);
  LBM_REQUIRE(blocks < 0x7fffffffLL, "encoder_tail: too many blocks");
  ProfScope prof("encoder_tail", stream, 0.0, 4.0 * LBM_C * B * ((double)P0 * (1.0 + 0.25 + 0.0625 + (pos ? 1.0 : 0.0)) + (double)L * (toks ? 2.0 : 1.0)));
  TailArgs ta{};
  ta.q1 = q1; ta.q2 = q2; ta.q3 = q3; ta.pos = pos;
  if (g_tail_order & 2) hipLaunchKernelGGL((pyramid_pool_kernel<1, 16, false>), dim3((unsigned)blocks), dim3(1024), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
  else hipLaunchKernelGGL((pyramid_pool_kernel<1, 16, true>), dim3((unsigned)blocks), dim3(1024), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
  return check_launch("encoder_tail");
}

int g_tail_order = 1;   // debug key 24: bit 0 = workgroups of the fused encoder tail stream-minor (0: stream by stream), bit 1 = cell-by-cell bilinear taps (the form before the patch)

// encoder_tail with the finest level's 64 -> 256 product inside: f1 split planes [B, h*w, 64], W1 split [256, ldw1], b1 [256] or null
int encoder_tail_fused(const void* f1, long long f1_plane, const void* w1h, const void* w1l, int ldw1, const float* b1, const float* q2,
                       const float* q3, const float* pos, float* tokens, int B, int h, int w, hipStream_t stream, void* toks,
                       long long toks_plane, float* inv_norm) {
  LBM_REQUIRE(f1 && w1h && w1l && q2 && q3 && tokens && B > 0 && h % 8 == 0 && w % 8 == 0, "encoder_tail_fused: bad arguments (grid %dx%d)", h, w);
  LBM_REQUIRE(((uintptr_t)f1 & 15) == 0 && f1_plane % 8 == 0 && ((uintptr_t)w1h & 15) == 0 && ((uintptr_t)w1l & 15) == 0 && ldw1 >= 64 && ldw1 % 8 == 0,
              "encoder_tail_fused: operands must be 16-byte aligned, weight rows >= 64 columns");
  const int P0 = h * w;
  const int L = P0 + P0 / 4 + P0 / 16 + P0 / 64;
  const long long blocks = (long long)B * (h / 8) * (w / 8);
  LBM_REQUIRE(blocks < 0x7fffffffLL, "encoder_tail_fused: too many blocks");
  ProfScope prof("encoder_tail", stream, 2.0 * B * P0 * 256.0 * 64.0,
                 4.0 * LBM_C * B * ((double)P0 * (0.25 + 0.25 + 0.0625 + (pos ? 1.0 : 0.0)) + (double)L * (toks ? 2.0 : 1.0)));
  TailArgs ta{};
  ta.q2 = q2; ta.q3 = q3; ta.pos = pos;
  ta.f1 = (const _Float16*)f1; ta.f1_plane = f1_plane; ta.w1h = (const _Float16*)w1h; ta.w1l = (const _Float16*)w1l; ta.ldw1 = ldw1; ta.b1 = b1; ta.inv = inv_norm;
  ta.stream_minor = ((g_tail_order & 1) && pos != nullptr && B >= 8) ? 1 : 0;
  ta.dbg = LBM_ABL(g_tail_order & ~15);
  const bool patch = !(g_tail_order & 2);
  if (blocks >= 512) {
    if (patch) hipLaunchKernelGGL((pyramid_pool_kernel<2, 8, true>), dim3((unsigned)blocks), dim3(512), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
    else hipLaunchKernelGGL((pyramid_pool_kernel<2, 8, false>), dim3((unsigned)blocks), dim3(512), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
  } else {
    if (patch) hipLaunchKernelGGL((pyramid_pool_kernel<2, 16, true>), dim3((unsigned)blocks), dim3(1024), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
    else hipLaunchKernelGGL((pyramid_pool_kernel<2, 16, false>), dim3((unsigned)blocks), dim3(1024), 0, stream, tokens, B, h, w, L, toks, toks_plane, ta);
  }
  return check_launch("encoder_tail_fused");
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ tok, float* __restrict__ inv, long long rows) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float sn = row_inv_norm(ld4(tok + row * LBM_C + lane * 4));
    if (lane == 0) inv[row] = sn;
  }
}

__global__ __launch_bounds__(256) void corr_maps_kernel(CorrMapArgs a) {
  const int lane = threadIdx.x & 63;
  const int P0 = a.h * a.w;
  const long long total = (long long)a.B * P0;
  for (long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); cell < total; cell += (long long)gridDim.x * 4) {
    const int b = (int)(cell / P0);
    const int p = (int)(cell % P0);
    const int y = p / a.w, x = p % a.w;
    const float* tb = a.tok + (long long)b * a.L * LBM_C + lane * 4;
    // (the row norms come from inv_norm_kernel: computing them here, 13 wave sums per cell, cost 9 us to save a 5 us launch)
    const float* ib = a.inv + (long long)b * a.L;
    float4 lv[4];
    {
      const float4 v = ld4(tb + (long long)p * LBM_C);
      const float s = ib[p];
      lv[0] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    }
    int start = P0;
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      const int hl = a.h >> l, wl = a.w >> l;
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      up_coeff(y, l, hl, y0, y1, ly0, ly1);
      up_coeff(x, l, wl, x0, x1, lx0, lx1);
      const int t00 = start + y0 * wl + x0, t01 = start + y0 * wl + x1;
      const int t10 = start + y1 * wl + x0, t11 = start + y1 * wl + x1;
      const float4 v00 = ld4(tb + (long long)t00 * LBM_C), v01 = ld4(tb + (long long)t01 * LBM_C);
      const float4 v10 = ld4(tb + (long long)t10 * LBM_C), v11 = ld4(tb + (long long)t11 * LBM_C);
      const float s00 = ib[t00], s01 = ib[t01], s10 = ib[t10], s11 = ib[t11];
      float4 r;
      r.x = ly0 * (lx0 * (v00.x * s00) + lx1 * (v01.x * s01)) + ly1 * (lx0 * (v10.x * s10) + lx1 * (v11.x * s11));
      r.y = ly0 * (lx0 * (v00.y * s00) + lx1 * (v01.y * s01)) + ly1 * (lx0 * (v10.y * s10) + lx1 * (v11.y * s11));
      r.z = ly0 * (lx0 * (v00.z * s00) + lx1 * (v01.z * s01)) + ly1 * (lx0 * (v10.z * s10) + lx1 * (v11.z * s11));
      r.w = ly0 * (lx0 * (v00.w * s00) + lx1 * (v01.w * s01)) + ly1 * (lx0 * (v10.w * s10) + lx1 * (v11.w * s11));
      lv[l] = r;
      start += hl * wl;
    }
    for (int i = 0; i < a.n_layers; ++i) {
      const float w0 = a.wmix[i][0] + 1.0f, w1 = a.wmix[i][1], w2 = a.wmix[i][2], w3 = a.wmix[i][3];
      float4 o;
      o.x = w0 * lv[0].x + w1 * lv[1].x + w2 * lv[2].x + w3 * lv[3].x;
      o.y = w0 * lv[0].y + w1 * lv[1].y + w2 * lv[2].y + w3 * lv[3].y;
      o.z = w0 * lv[0].z + w1 * lv[1].z + w2 * lv[2].z + w3 * lv[3].z;
      o.w = w0 * lv[0].w + w1 * lv[1].w + w2 * lv[2].w + w3 * lv[3].w;
      if (a.split) {
        const long long plane = (long long)a.B * P0 * LBM_C;
        st4_split(reinterpret_cast<_Float16*>(a.F) + (long long)i * 2 * plane, plane, ((long long)b * P0 + p) * LBM_C + lane * 4, o);
      } else {
        st4(a.F + (((long long)i * a.B + b) * P0 + p) * LBM_C + lane * 4, o);
      }
    }
  }
}

// lbm_hip_debug_set key 9: 0 (default) / 5 = 2x2-block kernel of THIS file (no packed-fp32 instructions); 6 = cell-by-cell
// kernel; 1..4 = the block kernel built WITH packed-fp32 instructions (corr_block_pk.hip: 1 plain -- the form that showed
// wrong cells beside a second queue --, 2 + trace, 3 every loaded row tied behind its wait, 4 the same + idle cycles)
int g_corr_variant = 0;
float* g_corr_trace = nullptr;   // lbm_hip_debug_ptr key 1
int g_corr_skip_norm = 0;        // debug key 10: 1 = the caller's inv_norm buffer is used as it is (scripts/stale_offline.py)

int corr_maps(const float* tokens, float* inv_norm, float* F, const float* wmix_host, int n_layers, int B, int h, int w,
              bool split, hipStream_t stream, bool have_norm) {
  LBM_REQUIRE(tokens && inv_norm && F && wmix_host, "corr_maps: null argument");
  LBM_REQUIRE(n_layers >= 1 && n_layers <= 8, "corr_maps: n_layers=%d out of range", n_layers);
  LBM_REQUIRE(B > 0 && h % 8 == 0 && w % 8 == 0, "corr_maps: grid %dx%d must be a multiple of 8", h, w);
  const int P0 = h * w;
  const int L = P0 + P0 / 4 + P0 / 16 + P0 / 64;
  long long rows = (long long)B * L;
  long long blocks = (rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  if (!g_corr_skip_norm && !have_norm) {
    ProfScope prof("inv_norm", stream, 0.0, 4.0 * LBM_C * rows);
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tokens, inv_norm, rows);
  }
  if (check_launch("inv_norm")) return -1;
  CorrMapArgs a;
  a.tok = tokens; a.inv = inv_norm; a.F = F; a.n_layers = n_layers; a.B = B; a.h = h; a.w = w; a.L = L; a.split = split ? 1 : 0;
  for (int i = 0; i < 8; ++i)
    for (int l = 0; l < 4; ++l) a.wmix[i][l] = i < n_layers ? wmix_host[i * 4 + l] : 0.f;
  blocks = ((long long)B * P0 + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  ProfScope prof("corr_maps", stream, 0.0, 4.0 * LBM_C * B * ((double)L + (double)n_layers * P0));
  a.trace = g_corr_trace;
  if (g_corr_variant == 6) {
    hipLaunchKernelGGL(corr_maps_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  } else {
    blocks = ((long long)B * (P0 / 4) + 3) / 4;      // one wave per 2x2 block of cells
    if (blocks > 16384) blocks = 16384;
    if (g_corr_variant >= 1 && g_corr_variant <= 4)
      launch_corr_block_packed(&a, g_corr_variant - 1, (unsigned)blocks, stream);
    else
      hipLaunchKernelGGL(corr_maps_block_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  }
  return check_launch("corr_maps");
}

// ------------------------------------------------------------------------------------------------
// top-k of each row: per-lane sorted candidate list (static registers), then K rounds of a wave arg-max.
// WPR waves share one row (each scans a contiguous quarter, the partial top-k lists are merged through LDS by wave 0):
// with ~1000 rows one wave per row leaves most of the chip idle and is bound by the latency of a 48 KB serial scan.
template <int KMAX, int WPR>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ corr, long long ld, int P, int k, int rows,
                                                       int grid_w, float cell, int* __restrict__ idx_out,
                                                       float* __restrict__ ref_out, float* __restrict__ val_out) {
  __shared__ float cv[4][16];
  __shared__ int ci[4][16];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = WPR == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (row >= rows) return;
  float lv[KMAX];
  int li[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) {
    lv[t] = -INFINITY;
    li[t] = 0x7fffffff;
  }
  const float* src = corr + (long long)row * ld;
  auto offer = [&](float v, int i) {
    if (v > lv[KMAX - 1] || (v == lv[KMAX - 1] && i < li[KMAX - 1])) {
      lv[KMAX - 1] = v;
      li[KMAX - 1] = i;
#pragma unroll
      for (int t = KMAX - 1; t > 0; --t) {
        if (lv[t] > lv[t - 1] || (lv[t] == lv[t - 1] && li[t] < li[t - 1])) {
          const float tv = lv[t]; lv[t] = lv[t - 1]; lv[t - 1] = tv;
          const int ti = li[t]; li[t] = li[t - 1]; li[t - 1] = ti;
        }
      }
    }
  };
  // this wave's share of the row, in float4 pieces
  const int nvec = P >> 2;
  const int per = (nvec + WPR - 1) / WPR;
  const int f0 = WPR == 1 ? 0 : wave * per;
  const int f1 = WPR == 1 ? nvec : min(nvec, f0 + per);
  const bool tail = WPR == 1 || wave == WPR - 1;
  // pass 1 (k > 1): the k-th largest of the 64 per-lane maxima is a lower bound of the k-th largest element, so only
  // elements >= tau can be among the k best; pass 2 then runs the sorted-list insertion for those few only.
  float tau = -INFINITY;
  if (KMAX > 1 && (f1 - f0) >= 64) {
    float mx = -INFINITY;
    for (int f = f0 + lane; f < f1; f += 64) {
      const float4 v = ld4(src + f * 4);
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int t = 0; t < k; ++t) {        // k rounds: wave max, then the (first) lane holding it drops out
      tau = wave_max(mx);
      const unsigned long long holders = __ballot(mx == tau);
      if (lane == __builtin_ctzll(holders)) mx = -INFINITY;
    }
  }
  for (int f = f0 + lane; f < f1; f += 64) {
    const float4 v = ld4(src + f * 4);
    if (fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) >= tau) {
      offer(v.x, f * 4);
      offer(v.y, f * 4 + 1);
      offer(v.z, f * 4 + 2);
      offer(v.w, f * 4 + 3);
    }
  }
  if (tail)
    for (int i = (nvec << 2) + lane; i < P; i += 64) offer(src[i], i);

  auto extract = [&](float& bv, int& bi) {   // wave arg-max of the list heads; the winning lane pops its head
    bv = lv[0];
    bi = li[0];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const float ov = __shfl_xor(bv, s);
      const int oi = __shfl_xor(bi, s);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (li[0] == bi) {
#pragma unroll
      for (int q = 0; q < KMAX - 1; ++q) {
        lv[q] = lv[q + 1];
        li[q] = li[q + 1];
      }
      lv[KMAX - 1] = -INFINITY;
      li[KMAX - 1] = 0x7fffffff;
    }
  };
  auto emit = [&](int t, float bv, int bi) {
    const long long o = (long long)row * k + t;
    idx_out[o] = bi;
    if (ref_out) {
      ref_out[o * 2 + 0] = (float)(bi % grid_w) * cell + 0.5f * cell;   // coord_utils.py:20-21
      ref_out[o * 2 + 1] = (float)(bi / grid_w) * cell + 0.5f * cell;
    }
    if (val_out) val_out[o] = bv;
  };

  if (WPR == 1) {
    for (int t = 0; t < k; ++t) {
      float bv;
      int bi;
      extract(bv, bi);
      if (lane == 0) emit(t, bv, bi);
    }
    return;
  }
  // each wave parks its k best in LDS; wave 0 merges the WPR * k (<= 64) candidates, one per lane
  for (int t = 0; t < k; ++t) {
    float bv;
    int bi;
    extract(bv, bi);
    if (lane == 0) {
      cv[wave][t] = bv;
      ci[wave][t] = bi;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  float mv = -INFINITY;
  int mi = 0x7fffffff;
  if (lane < WPR * k) {
    mv = cv[lane / k][lane % k];
    mi = ci[lane / k][lane % k];
  }
  for (int t = 0; t < k; ++t) {
    float bv = mv;
    int bi = mi;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const float ov = __shfl_xor(bv, s);
      const int oi = __shfl_xor(bi, s);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (mi == bi) {
      mv = -INFINITY;
      mi = 0x7fffffff;
    }
    if (lane == 0) emit(t, bv, bi);
  }
}

// Few rows (one stream, ~1000 points): one 256-thread workgroup per row, the row held in registers (<= 16 float4 per
// thread), so it is read ONCE.  tau = the largest, over the four waves, of the k-th largest per-lane maximum is a lower
// bound of the k-th largest element; the elements >= tau are compacted into LDS.  Up to 256 of them (peaky maps) are
// ranked directly (rank = number of candidates that precede it in (value desc, index asc) order).  Broad maxima leave
// more: then the k-th largest value T is found exactly by bisection on the order-preserving integer image of the
// values (largest T with >= k elements >= T) and the ties at T are cut by bisection on the index (lowest first) --
// over the <= 1024 candidates (4 per thread) when they fit, over the whole row in registers otherwise (flat rows).
__device__ __forceinline__ unsigned topk_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float topk_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// CAND: the row is a list of (value, column) candidates (the per-tile pre-selection of the fused correlation product,
// gemm16.hip) instead of a dense row; columns then come from cand_i and run up to idx_max.
template <int NV, bool CAND>
__global__ __launch_bounds__(256) void topk_rows_wide_kernel(const float* __restrict__ corr, long long ld, int P, int k, int rows,
                                                            int grid_w, float cell, int* __restrict__ idx_out,
                                                            float* __restrict__ ref_out, float* __restrict__ val_out,
                                                            const int* __restrict__ cand_i, int idx_max) {
  constexpr int CAP = 1024;
  __shared__ float s_tau[4];
  __shared__ float s_red[2][4];
  __shared__ int s_cnt;
  __shared__ float s_v[CAP];
  __shared__ int s_i[CAP];
  __shared__ float s_fv[64];
  __shared__ int s_fi[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nvec = P >> 2;
  const float* src = corr + (long long)row * ld;
  float4 buf[NV];
  int4 ibuf[CAND ? NV : 1];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int f = tid + j * 256;
    buf[j] = f < nvec ? ld4(src + f * 4) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (CAND)
      ibuf[j] = f < nvec ? *reinterpret_cast<const int4*>(cand_i + (long long)row * ld + f * 4)
                         : make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
    mx = fmaxf(fmaxf(mx, fmaxf(buf[j].x, buf[j].y)), fmaxf(buf[j].z, buf[j].w));
  }
  auto idx_of = [&](int j, int c) {
    if (CAND) return c == 0 ? ibuf[j].x : c == 1 ? ibuf[j].y : c == 2 ? ibuf[j].z : ibuf[j].w;
    return (tid + j * 256) * 4 + c;
  };
  const float rmax_w = wave_max(mx);
  float tau = -INFINITY;
  for (int t = 0; t < k; ++t) {        // k rounds: wave max, then the (first) lane holding it drops out
    tau = wave_max(mx);
    const unsigned long long holders = __ballot(mx == tau);
    if (lane == __builtin_ctzll(holders)) mx = -INFINITY;
  }
  if (lane == 0) {
    s_tau[wave] = tau;
    s_red[0][wave] = rmax_w;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  tau = fmaxf(fmaxf(s_tau[0], s_tau[1]), fmaxf(s_tau[2], s_tau[3]));
  const float rmax = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
  auto push = [&](float v, int i) {
    if (v >= tau) {
      const int p = atomicAdd(&s_cnt, 1);
      if (p < CAP) {
        s_v[p] = v;
        s_i[p] = i;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (fmaxf(fmaxf(buf[j].x, buf[j].y), fmaxf(buf[j].z, buf[j].w)) >= tau) {
      push(buf[j].x, idx_of(j, 0));
      push(buf[j].y, idx_of(j, 1));
      push(buf[j].z, idx_of(j, 2));
      push(buf[j].w, idx_of(j, 3));
    }
  }
  __syncthreads();
  const int n = s_cnt;
  auto emit = [&](int t, float bv, int bi) {
    const long long o = (long long)row * k + t;
    idx_out[o] = bi;
    if (ref_out) {
      ref_out[o * 2 + 0] = (float)(bi % grid_w) * cell + 0.5f * cell;   // coord_utils.py:20-21
      ref_out[o * 2 + 1] = (float)(bi / grid_w) * cell + 0.5f * cell;
    }
    if (val_out) val_out[o] = bv;
  };
  // rank the m <= 64 entries of (fv, fi) and write the first k in order
  auto rank_emit = [&](const float* fv, const int* fi, int m) {
    if (tid < m) {
      const float mv = fv[tid];
      const int mi = fi[tid];
      int rank = 0;
      for (int q = 0; q < m; ++q) {
        const float ov = fv[q];
        const int oi = fi[q];
        rank += (ov > mv || (ov == mv && oi < mi)) ? 1 : 0;
      }
      if (rank < k) emit(rank, mv, mi);
    }
  };
  if (n <= 64) {
    rank_emit(s_v, s_i, n);
    return;
  }
  if (n <= 256) {
    if (tid < n) {
      const float mv = s_v[tid];
      const int mi = s_i[tid];
      int rank = 0;
      for (int q = 0; q < n; ++q) {
        const float ov = s_v[q];
        const int oi = s_i[q];
        rank += (ov > mv || (ov == mv && oi < mi)) ? 1 : 0;
      }
      if (rank < k) emit(rank, mv, mi);
    }
    return;
  }

  int flip = 1;                                   // s_red[0] was read above; every thread is past the barrier after it
  auto block_sum = [&](int c) {
    const float w = wave_sum((float)c);           // <= 64 * 64 per wave: exact in fp32
    if (lane == 0) s_red[flip][wave] = w;
    __syncthreads();
    const int tot = (int)(s_red[flip][0] + s_red[flip][1] + s_red[flip][2] + s_red[flip][3]);
    flip ^= 1;                                    // the next sum writes the other slot: one barrier per sum suffices
    return tot;
  };
  // exact selection among this thread's NE elements (keys ke[], indices ie[]; key 0 / index INT_MAX = no element)
  auto select = [&](const auto& ke, const auto& ie) {
    constexpr int NE = sizeof(ke) / sizeof(ke[0]);
    auto count_ge = [&](unsigned t) {
      int c = 0;
#pragma unroll
      for (int e = 0; e < NE; ++e) c += ke[e] >= t ? 1 : 0;
      return block_sum(c);
    };
    unsigned lo = topk_key(tau), hi = topk_key(rmax);
    while (lo < hi) {
      const unsigned mid = lo + (hi - lo + 1) / 2;
      if (count_ge(mid) >= k) lo = mid; else hi = mid - 1;
    }
    const unsigned T = lo;
    const int need = k - (T == 0xffffffffu ? 0 : count_ge(T + 1));     // how many of the elements equal to T are taken
    int jlo = 0, jhi = CAND ? idx_max : P - 1;
    while (jlo < jhi) {
      const int jm = (jlo + jhi) >> 1;
      int c = 0;
#pragma unroll
      for (int e = 0; e < NE; ++e) c += (ke[e] == T && ie[e] <= jm) ? 1 : 0;
      if (block_sum(c) >= need) jhi = jm; else jlo = jm + 1;
    }
    const int J = jlo;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (ke[e] > T || (ke[e] == T && ie[e] <= J)) {
        const int p = atomicAdd(&s_cnt, 1);
        if (p < 64) {
          s_fv[p] = topk_unkey(ke[e]);
          s_fi[p] = ie[e];
        }
      }
    }
    __syncthreads();
    rank_emit(s_fv, s_fi, min(s_cnt, 64));
  };
  if (n <= CAP) {
    unsigned ke[CAP / 256];
    int ie[CAP / 256];
#pragma unroll
    for (int q = 0; q < CAP / 256; ++q) {
      const int p = tid + q * 256;
      ke[q] = p < n ? topk_key(s_v[p]) : 0u;
      ie[q] = p < n ? s_i[p] : 0x7fffffff;
    }
    select(ke, ie);
  } else {
    unsigned ke[NV * 4];
    int ie[NV * 4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool in = (tid + j * 256) * 4 < P;
      ke[4 * j + 0] = in ? topk_key(buf[j].x) : 0u; ie[4 * j + 0] = in ? idx_of(j, 0) : 0x7fffffff;
      ke[4 * j + 1] = in ? topk_key(buf[j].y) : 0u; ie[4 * j + 1] = in ? idx_of(j, 1) : 0x7fffffff;
      ke[4 * j + 2] = in ? topk_key(buf[j].z) : 0u; ie[4 * j + 2] = in ? idx_of(j, 2) : 0x7fffffff;
      ke[4 * j + 3] = in ? topk_key(buf[j].w) : 0u; ie[4 * j + 3] = in ? idx_of(j, 3) : 0x7fffffff;
    }
    select(ke, ie);
  }
}

template <int KMAX>
static void launch_topk(bool wide, dim3 grid, hipStream_t stream, const float* corr, long long ld, int P, int k, int rows, int grid_w,
                        float cell, int* idx_out, float* ref_out, float* val_out) {
  if (wide)
    hipLaunchKernelGGL((topk_rows_kernel<KMAX, 4>), grid, dim3(256), 0, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
  else
    hipLaunchKernelGGL((topk_rows_kernel<KMAX, 1>), grid, dim3(256), 0, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
}

static int g_topk_mode = 0;
void topk_force_mode(int m) { g_topk_mode = m; }

int topk_rows(const float* corr, long long ld, int P, int k, int rows, int grid_w, int stride_px, int* idx_out,
              float* ref_out, float* val_out, hipStream_t stream) {
  if (rows == 0) return 0;
  LBM_REQUIRE(corr && idx_out, "topk_rows: null argument");
  LBM_REQUIRE(k >= 1 && k <= 16 && k <= P, "topk_rows: k=%d out of range", k);
  LBM_REQUIRE(ld % 4 == 0 && ((uintptr_t)corr & 15) == 0, "topk_rows: rows must be 16-byte aligned");
  const bool wide = rows < 4096 && P >= 1024;        // few rows: four waves per row
  dim3 grid(wide ? rows : (rows + 3) / 4);
  const float cell = (float)stride_px;
  ProfScope prof(k == 1 ? "topk_rows_k1" : k <= 4 ? "topk_rows_k4" : k <= 9 ? "topk_rows_k9" : "topk_rows_k16", stream, 0.0,
                 4.0 * rows * ((double)P + 4.0 * k));
  if (wide && P % 4 == 0 && P <= 16 * 1024 && g_topk_mode == 0) {   // the row fits the registers of one workgroup
    if (P <= 4 * 1024)
      hipLaunchKernelGGL((topk_rows_wide_kernel<4, false>), grid, dim3(256), 0, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out, nullptr, 0);
    else if (P <= 12 * 1024)
      hipLaunchKernelGGL((topk_rows_wide_kernel<12, false>), grid, dim3(256), 0, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out, nullptr, 0);
    else
      hipLaunchKernelGGL((topk_rows_wide_kernel<16, false>), grid, dim3(256), 0, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out, nullptr, 0);
    return check_launch("topk_rows");
  }
  if (k == 1) launch_topk<1>(wide, grid, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
  else if (k <= 4) launch_topk<4>(wide, grid, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
  else if (k <= 9) launch_topk<9>(wide, grid, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
  else launch_topk<16>(wide, grid, stream, corr, ld, P, k, rows, grid_w, cell, idx_out, ref_out, val_out);
  return check_launch("topk_rows");
}

// Final selection over the per-tile candidates of the fused correlation product: rows of n_cand (value, column) pairs.
int topk_candidates(const float* cand_v, const int* cand_i, int n_cand, int P, int k, int rows, int grid_w, int stride_px, int* idx_out,
                    float* ref_out, float* val_out, hipStream_t stream) {
  if (rows == 0) return 0;
  LBM_REQUIRE(cand_v && cand_i && idx_out, "topk_candidates: null argument");
  LBM_REQUIRE(k >= 1 && k <= 16 && k <= n_cand && n_cand % 4 == 0 && n_cand <= 4096, "topk_candidates: k=%d, n_cand=%d out of range", k, n_cand);
  LBM_REQUIRE(((uintptr_t)cand_v & 15) == 0 && ((uintptr_t)cand_i & 15) == 0, "topk_candidates: rows must be 16-byte aligned");
  const float cell = (float)stride_px;
  ProfScope prof(k == 1 ? "topk_cand_k1" : k <= 4 ? "topk_cand_k4" : "topk_cand_k9", stream, 0.0, 8.0 * rows * ((double)n_cand + k));
  const dim3 grid(rows);
  if (n_cand <= 1024)
    hipLaunchKernelGGL((topk_rows_wide_kernel<1, true>), grid, dim3(256), 0, stream, cand_v, (long long)n_cand, n_cand, k, rows, grid_w, cell,
                       idx_out, ref_out, val_out, cand_i, P - 1);
  else
    hipLaunchKernelGGL((topk_rows_wide_kernel<4, true>), grid, dim3(256), 0, stream, cand_v, (long long)n_cand, n_cand, k, rows, grid_w, cell,
                       idx_out, ref_out, val_out, cand_i, P - 1);
  return check_launch("topk_candidates");
}

}  // namespace lbm
