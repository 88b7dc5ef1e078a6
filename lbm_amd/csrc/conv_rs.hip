// conv64_rows: 3 x 3, stride 1, pad 1 convolution 64 -> 64 channels (ResNet layer1, backbone.py:48-65) for MANY pixels, NHWC split planes.
//
// The tiled implicit GEMM (gemm16.hip, CONV) fetches every input pixel nine times (once per tap) as im2col rows through LDS-DMA and,
// with only 64 output channels, stages 40 DMA pieces per 96 MFMAs.  Here the weights (64 x 576, 147 KB split) stay in registers and
// the IMAGE streams through: a step is 32 consecutive output pixels of one image row, its input is the 3 x 34 pixel patch around
// them (30 KB with padding, fetched once), and the nine taps read their operand fragments straight out of that patch in LDS.
//
// Workgroup = 8 waves.  Four multiply (one per SIMD): wave (kh, cb) owns output columns [32 cb, 32 cb + 32) and half kh of the K = 576 extent (18 of
// the 36 k-steps: 144 VGPRs of weight fragments).  The two halves of a column block form a chain so that the accumulation ORDER is the
// tiled kernel's (results bit-identical): wave (0, cb) multiplies k-steps 0..17 of step j and hands its two accumulators over through
// LDS; wave (1, cb) picks them up one step later, adds k-steps 18..35 and runs the epilogue (bias, residual, ReLU, fp32 / split
// output) is NOT theirs: four helper waves share the SIMDs with them -- two issue the LDS-DMA of the patches (hand-counted vmcnt, no
// other memory traffic in those waves), two take the finished 32 x 32 blocks from LDS one more step later and do bias-free residual /
// ReLU / output stores (ordinary loads and stores, compiler-managed waits) -- so address arithmetic, DMA issue and stores sit in the
// issue slots the MFMAs leave free.  Patches live in a ring of 4 stages (step j is read in iterations j and j + 1, steps j + 1 and
// j + 2 are in flight); one barrier per step.
//
// Patch image in LDS: [ky 0..2][plane][pixel 0..33][64 channels] f16, a pixel = 128 B = eight 16-byte slots, slot s of pixel p stored at
// s ^ ((p >> 1) & 7): the 16 lanes of a ds_read_b128 group read 16 consecutive pixels -> 16 different (p & 1, (p >> 1) & 7) -> no
// bank conflict.  The swizzle is applied on the source side of the DMA (the LDS side is lane-linear).
#include "gemm.hpp"

#include <cstdlib>
#include <type_traits>

namespace lbm {

int g_conv64_dbg = 0;   // debug key 18

struct Conv64Dev {
  const _Float16* X;
  long long x_plane;
  const _Float16* Wh;
  const _Float16* Wl;
  int ldw;
  const float* bias;
  const float* res;
  float* Y;
  _Float16* Ys;
  long long ys_plane;
  const _Float16* zeros;
  int B, H, W, relu;
  int steps_total, spg;
  int dbg;   // tuning (timing only): 1 no DMA after the prologue, 2 no epilogue, 4 no producer MFMAs, 8 no consumer MFMAs
};

typedef __attribute__((address_space(3))) void* lds_ptr_c64;

[[maybe_unused]] static __device__ __align__(256) unsigned char g_zero_row[34 * 128];      // what rows above / below the image and pixels -1 / W read (never written)

__global__ __launch_bounds__(512) void conv64_rows_kernel(Conv64Dev a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWB = 34 * 128;            // bytes of one (ky, plane) pixel row of the patch: 34 pixels, no padding
  constexpr int PATCH = 6 * ROWB;           // 26112
  constexpr int NST = 4;
  constexpr int HAND = 8192;                // one wave's two accumulators
  constexpr int TP = 36;
  constexpr int FIN = 32 * TP * 4;          // one wave's finished 32 x 32 block, row-major with pitch TP
  constexpr int OFF_HAND = NST * PATCH;     // [2 buffers][2 column blocks][HAND]
  constexpr int OFF_FIN = OFF_HAND + 4 * HAND;   // [2 buffers][2 column blocks][FIN]
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OFF_FIN + 4 * FIN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int role = wave >> 1, cb = wave & 1;          // 0 first half of K, 1 second half, 2 DMA, 3 epilogue; waves w and w + 4 share a SIMD
  const int r = lane & 31, h = lane >> 5;
  const int s0 = blockIdx.x * a.spg;
  const int ns = min(a.spg, a.steps_total - s0);
  if (ns <= 0) return;
  const int spr = a.W >> 5;                 // steps per image row

  auto handover = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (role < 2) {
    const int kh = role;
    // ---- this wave's half of the weights as MFMA B-fragments: lane (r, h) = output column 32 cb + r, k = 16 ks + 8 h .. + 8 ----
    half8 wh[18], wl[18];
    {
      const long long wo = (long long)(cb * 32 + r) * a.ldw + kh * 288 + 8 * h;
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        wh[i] = *reinterpret_cast<const half8*>(a.Wh + wo + i * 16);
        wl[i] = *reinterpret_cast<const half8*>(a.Wl + wo + i * 16);
      }
    }
    // fragment byte offsets inside a (ky, plane) row for tap column kx and channel quarter q: pixel r + kx, slot (2 q + h) ^ swizzle
    int foff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int p = r + kx, sw = (p >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) foff[kx][q] = p * 128 + (((2 * q + h) ^ sw) << 4);
    }
    // k-steps [18 KH, 18 KH + 18) of the patch at Pb, fragments of the next pair of k-steps requested before this pair's MFMAs
    f32x16 acc, acx;
    auto mac = [&](const unsigned char* Pb, auto khc) __attribute__((always_inline)) {
      constexpr int KH = decltype(khc)::value;          // compile-time half: taps and channel quarters of every k-step are constants
      half8 fa[2][4];
      auto ldg = [&](const int bsel, const int g) __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int ks = 18 * KH + 2 * g + qq;
          const int tap = ks >> 2, q = ks & 3;
          const int ky = tap / 3, kx = tap - 3 * ky;
          const unsigned char* p = Pb + (ky * 2) * ROWB + foff[kx][q];
          fa[bsel][2 * qq] = *reinterpret_cast<const half8*>(p);
          fa[bsel][2 * qq + 1] = *reinterpret_cast<const half8*>(p + ROWB);
        }
      };
      ldg(0, 0);
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) ldg((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int i = 2 * g + qq;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * qq], wh[i], acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * qq], wl[i], acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * qq + 1], wh[i], acx, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (kh == 0) {
      // ---- first half of K for step k, accumulators handed to the second half through LDS ----
      for (int k = 0; k <= ns + 1; ++k) {
        handover();
        if (k >= ns) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = acx[e] = 0.f;
        if (!(LBM_ABL(a.dbg & 4))) mac(lds + (k & (NST - 1)) * PATCH, std::integral_constant<int, 0>{});
        f32x4* hb = reinterpret_cast<f32x4*>(lds + OFF_HAND + ((k & 1) * 2 + cb) * HAND) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hb[i * 64] = f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
          hb[(4 + i) * 64] = f32x4{acx[4 * i], acx[4 * i + 1], acx[4 * i + 2], acx[4 * i + 3]};
        }
      }
    } else {
      // ---- second half of K for step k - 1; the finished block goes to the epilogue wave, row-major, bias added ----
      const float bv = a.bias ? a.bias[cb * 32 + r] : 0.f;
      for (int k = 0; k <= ns + 1; ++k) {
        handover();
        if (k == 0 || k > ns) continue;
        const int j = k - 1;
        const f32x4* hb = reinterpret_cast<const f32x4*>(lds + OFF_HAND + ((j & 1) * 2 + cb) * HAND) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 u = hb[i * 64], v = hb[(4 + i) * 64];
          acc[4 * i] = u[0]; acc[4 * i + 1] = u[1]; acc[4 * i + 2] = u[2]; acc[4 * i + 3] = u[3];
          acx[4 * i] = v[0]; acx[4 * i + 1] = v[1]; acx[4 * i + 2] = v[2]; acx[4 * i + 3] = v[3];
        }
        if (!(LBM_ABL(a.dbg & 8))) mac(lds + (j & (NST - 1)) * PATCH, std::integral_constant<int, 1>{});
        float* Tw = reinterpret_cast<float*>(lds + OFF_FIN + ((j & 1) * 2 + cb) * FIN);
#pragma unroll
        for (int e = 0; e < 16; ++e) Tw[(4 * h + (e & 3) + 8 * (e >> 2)) * TP + r] = (acc[e] + acx[e] * (1.0f / 2048.0f)) + bv;
      }
    }
  } else if (role == 2) {
    // ---- DMA: the patch of step k + 2.  Wave cb fetches plane cb: 3 rows x 5 pieces of 8 pixels (the fifth piece of a row holds
    // pixels 32 and 33 only: its other lanes are switched off, the next row starts right behind).  Everything per lane is constant
    // over the steps (byte offset of its 16-byte slot from the row's pixel -1); everything that changes is uniform (the row's base
    // address, or the zero row for a row above / below the image) -- two vector instructions per piece instead of an address
    // computation, since these waves share their SIMDs' issue slots with the multiplying waves.
    int voff[5];
#pragma unroll
    for (int seg = 0; seg < 5; ++seg) {
      const int pq = seg * 8 + (lane >> 3);
      voff[seg] = pq * 64 + (((lane & 7) ^ ((pq >> 1) & 7)) << 3);      // in f16 elements
    }
    const _Float16* zrow = reinterpret_cast<const _Float16*>(g_zero_row);
    const _Float16* Xp = a.X + cb * a.x_plane;
    // raster position of the next step to issue
    int ib, iy, ix;
    {
      const int row = s0 / spr;
      ix = (s0 - row * spr) << 5;
      ib = row / a.H; iy = row - ib * a.H;
    }
    auto issue = [&](const int j) __attribute__((always_inline)) {
      unsigned char* stage = lds + (j & (NST - 1)) * PATCH + cb * ROWB;
      const bool eL = ix == 0, eR = ix + 32 == a.W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int y = iy - 1 + ky;
        const _Float16* rb = (unsigned)y < (unsigned)a.H ? Xp + (((long long)ib * a.H + y) * a.W + ix - 1) * 64 : zrow;
#pragma unroll
        for (int seg = 0; seg < 5; ++seg) {
          const _Float16* src = rb + voff[seg];
          if (seg == 0 && eL && lane < 8) src = zrow + voff[0];                   // pixel -1 of the image row
          if (seg == 4 && eR && (lane >> 3) == 1) src = zrow + voff[4];           // pixel W
          unsigned char* dst = stage + ky * 2 * ROWB + seg * 1024;
          if (seg < 4 || lane < 16) __builtin_amdgcn_global_load_lds(src, (lds_ptr_c64)dst, 16, 0, 0);
        }
      }
      ix += 32;
      if (ix == a.W) { ix = 0; if (++iy == a.H) { iy = 0; ++ib; } }
    };
    issue(0);
    if (ns > 1) issue(1);
    for (int k = 0; k <= ns + 1; ++k) {
      if (k + 1 < ns) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");    // patch k landed, patch k + 1 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      handover();
      if (k + 2 < ns && !(LBM_ABL(a.dbg & 1))) issue(k + 2);      // its stage held patch k - 2: last read in iteration k - 1
    }
  } else {
    // ---- epilogue of step k - 2: residual, ReLU, 16-byte row pieces of the fp32 and / or split output.  The residual rows of a
    // step are requested two iterations before they are used (three register sets, loop unrolled by three so that no set is ever
    // moved): a memory round trip per step on this wave's path made the residual form twice as slow as the plain one ----
    const int lrow = lane >> 3, c4 = lane & 7;
    const long long lo = (long long)lrow * 64 + cb * 32 + c4 * 4;      // this lane's element inside a step's 32 x 64 block (+ 512 i)
    f32x4 pr[3][4];
    const int n_st = (a.Y ? 4 : 0) + (a.Ys ? 8 : 0);              // vector-memory operations of one step's stores
    // The residual loads are issued from inline asm so that the compiler does not wait for them (it would wait for the NEWEST
    // set on every back edge); the wait below counts, in issue order, what was issued after the set that is about to be used.
    auto fetch = [&](auto ic, const int j) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      if (a.res == nullptr || j >= ns) return;
      const float* rp = a.res + ((long long)(s0 + j) << 11) + lo;     // steps are 32 consecutive pixels of the NHWC raster (W % 32 == 0)
      f32x4 &p0 = pr[I][0], &p1 = pr[I][1], &p2 = pr[I][2], &p3 = pr[I][3];
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:2048\n\t"
                   "global_load_dwordx4 %2, %5, off\n\tglobal_load_dwordx4 %3, %5, off offset:2048"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(rp), "v"(rp + 1024) : "memory");
    };
    auto wait_newer = [&](const int n) __attribute__((always_inline)) {       // n: a multiple of 4 in [0, 32]
      switch (n >> 2) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
      }
    };
    auto finish = [&](auto ic, const int j) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      const float* Tw = reinterpret_cast<const float*>(lds + OFF_FIN + ((j & 1) * 2 + cb) * FIN);
      const long long o0 = ((long long)(s0 + j) << 11) + lo;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(Tw + (i * 8 + lrow) * TP + c4 * 4);
      if (a.res) {
        // issued after this step's residual set (k = j + 2): stores of step j - 2 and j - 1, residual sets of steps j + 1 and j + 2
        const int k = j + 2;
        wait_newer((k - 1 < ns ? 4 : 0) + (k < ns ? 4 : 0) + n_st * ((k >= 4 ? 1 : 0) + (k >= 3 ? 1 : 0)));
        f32x4 &p0 = pr[I][0], &p1 = pr[I][1], &p2 = pr[I][2], &p3 = pr[I][3];
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (a.res) { v[i].x += pr[I][i][0]; v[i].y += pr[I][i][1]; v[i].z += pr[I][i][2]; v[i].w += pr[I][i][3]; }
        if (a.relu) v[i] = make_float4(fmaxf(v[i].x, 0.f), fmaxf(v[i].y, 0.f), fmaxf(v[i].z, 0.f), fmaxf(v[i].w, 0.f));
        if (a.Y) st4(a.Y + o0 + i * 512, v[i]);
        if (a.Ys) st4_split(a.Ys, a.ys_plane, o0 + i * 512, v[i]);
      }
    };
    // iteration k: request the residual of step k (set k % 3), finish step k - 2 (set (k + 1) % 3)
    auto iter = [&](auto ic, const int k) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      handover();
      fetch(ic, k);
      if (k >= 2 && !(LBM_ABL(a.dbg & 2))) finish(std::integral_constant<int, (I + 1) % 3>{}, k - 2);
    };
    for (int k = 0; k <= ns + 1; k += 3) {
      iter(std::integral_constant<int, 0>{}, k);
      if (k + 1 > ns + 1) break;
      iter(std::integral_constant<int, 1>{}, k + 1);
      if (k + 2 > ns + 1) break;
      iter(std::integral_constant<int, 2>{}, k + 2);
    }
  }
#endif
}

bool conv64_rows_ok(const ConvSSArgs& c) {
  return c.planes == 2 && c.Cin == 64 && c.Cout == 64 && c.kh == 3 && c.kw == 3 && c.stride == 1 && c.pad == 1 && c.Wd % 32 == 0 &&
         c.Wl != nullptr && c.ldw >= 576 && c.ldw % 8 == 0 && c.ys_plane % 4 == 0 && ((uintptr_t)c.Y & 15) == 0 && ((uintptr_t)c.Ys & 7) == 0 &&
         ((uintptr_t)c.res & 15) == 0 && c.x_plane % 8 == 0;
}

int conv64_rows(const ConvSSArgs& c, hipStream_t stream) {
  LBM_REQUIRE(conv64_rows_ok(c) && c.X && c.Wh && (c.Y || c.Ys) && c.zeros, "conv64_rows: shape / alignment not supported");
  const long long M = (long long)c.B * c.H * c.Wd;
  LBM_REQUIRE(M / 32 < 0x7fffffffLL, "conv64_rows: too many pixels");
  Conv64Dev d;
  d.X = (const _Float16*)c.X; d.x_plane = c.x_plane; d.Wh = (const _Float16*)c.Wh; d.Wl = (const _Float16*)c.Wl; d.ldw = c.ldw;
  d.bias = c.bias; d.res = c.res; d.Y = c.Y; d.Ys = (_Float16*)c.Ys; d.ys_plane = c.ys_plane; d.zeros = (const _Float16*)c.zeros;
  d.B = c.B; d.H = c.H; d.W = c.Wd; d.relu = c.relu;
  d.dbg = g_conv64_dbg & 255;
  d.steps_total = (int)(M / 32);
  int wgs = d.steps_total < 256 ? d.steps_total : 256;
  d.spg = (d.steps_total + wgs - 1) / wgs;
  if (g_conv64_dbg >> 8) d.spg = g_conv64_dbg >> 8;      // experiments: steps per workgroup
  wgs = (d.steps_total + d.spg - 1) / d.spg;
  const double mnk = (double)M * 64 * 576;
  ProfScope prof("conv64_rows", stream, 2.0 * mnk, 4.0 * ((double)M * 64 * (1 + (c.Y ? 1 : 0) + (c.Ys ? 1 : 0) + (c.res ? 1 : 0)) + 64.0 * 576));
  hipLaunchKernelGGL(conv64_rows_kernel, dim3(wgs), dim3(512), 0, stream, d);
  return check_launch("conv64_rows");
}

}  // namespace lbm
