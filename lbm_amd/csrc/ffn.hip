// Fused feed-forward for the per-point chain: slabs[c] = GELU(A · F1[c]ᵀ + b1[c]) · F2[:, c]ᵀ for hidden chunk c.
//
// transformer.py:343-355 (MLP 256 -> 1024 -> 256) runs 15 times per frame on [R, 256] rows.  As two products it costs two
// launches and a round trip of the [R, 1024] hidden tensor; with ~1000 rows both are latency, not work.  Here one
// workgroup owns 32 rows and one chunk of CH hidden channels: it forms its [32, CH] piece of the hidden layer, keeps it
// in LDS as split f16 planes, and multiplies it straight into a partial [32, 256] output -- the K-split slab the
// LayerNorm kernel that follows reduces anyway (bias of the second layer and the residual are added there).
//
// Weights are read in FRAGMENT-MAJOR order (pack_fragment_major below): for every (32-row block, 16-deep k step) the
// 64 lanes' MFMA B-operand fragments of the hi plane, then of the lo plane, lie in 2 contiguous KiB.  A wave therefore
// fetches each fragment with one fully coalesced 16-byte-per-lane load straight into registers: no LDS staging, no
// barrier and no bank conflicts on the weight side, and all of a wave's weight loads are in flight before its first
// MFMA.  The only LDS traffic is the 32-row activation tile and the hidden piece.
#include <algorithm>

#include "common.hpp"
#include "gemm.hpp"

namespace lbm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// planes [2][N][K] f16 (hi, lo) -> out [N/32][K/16][2][64][8]: lane = 32 * (k half of the step) + row within the block
__global__ __launch_bounds__(256) void pack_fragment_major_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                                  int N, int K, _Float16* __restrict__ out) {
  const long long total = (long long)N * K / 8 * 2;      // 16-byte pieces
  const int KS = K / 16;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
    const int lane = (int)(p & 63);
    const int plane = (int)((p >> 6) & 1);
    const long long blk = p >> 7;
    const int ks = (int)(blk % KS), nb = (int)(blk / KS);
    const int row = nb * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8;
    const _Float16* src = (plane ? lo : hi) + (long long)row * K + k;
    *reinterpret_cast<half8*>(out + p * 8) = *reinterpret_cast<const half8*>(src);
  }
}

int pack_fragment_major(const void* hi, const void* lo, int N, int K, void* out, hipStream_t stream) {
  LBM_REQUIRE(hi && lo && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "pack_fragment_major: needs N %% 32 == 0 and K %% 16 == 0");
  const long long pieces = (long long)N * K / 4;
  hipLaunchKernelGGL(pack_fragment_major_kernel, dim3((unsigned)std::min<long long>((pieces + 255) / 256, 4096)), dim3(256), 0, stream,
                     (const _Float16*)hi, (const _Float16*)lo, N, K, (_Float16*)out);
  return check_launch("pack_fragment_major");
}

struct FfnDev {
  const _Float16* A;      // [M, lda] hi plane, lo plane at A + a_plane
  long long a_plane;
  int lda;
  const _Float16* F1;     // fragment-major [1024 / 32][256 / 16][2][64][8]
  const float* b1;        // [1024]
  const _Float16* F2;     // fragment-major [256 / 32][1024 / 16][2][64][8]
  float* slabs;           // [1024 / CH][M, 256]
  long long slab_stride;
  int M;
  int set_rows, nb0;      // two weight sets: rows of the first set and its 32-row blocks (0: one set)
  const _Float16* F1b;
  const float* b1b;
  const _Float16* F2b;
};

constexpr int kC = 256, kHid = 1024, kRows = 32;
constexpr int kAP = kC + 8;                 // LDS pitch (halves) of the activation tile: 528 B = 4 banks past a multiple of 64

template <int CH, int NPL, bool CLAIM>
__global__ __launch_bounds__(256) void ffn_fused_kernel(FfnDev a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT1 = CH / 32 / 4;           // hidden tiles per wave
  constexpr int KS1 = kC / 16;               // 16 k steps over the 256 input channels
  constexpr int KS2 = CH / 16;               // k steps over this chunk of the hidden layer
  constexpr int HP = CH + 8;                 // LDS pitch (halves) of the hidden piece
  __shared__ __attribute__((aligned(16))) _Float16 sA[NPL][kRows * kAP];
  __shared__ __attribute__((aligned(16))) _Float16 sH[NPL][kRows * HP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int c = blockIdx.x;
  // two weight sets: the blocks of the second set start at row set_rows, each set ends at its own last row
  const bool set2 = a.set_rows != 0 && (int)blockIdx.y >= a.nb0;
  const int m0 = set2 ? a.set_rows + ((int)blockIdx.y - a.nb0) * kRows : blockIdx.y * kRows;
  const int mend = (a.set_rows != 0 && !set2) ? a.set_rows : a.M;
  const _Float16* F1 = set2 ? a.F1b : a.F1;
  const _Float16* F2 = set2 ? a.F2b : a.F2;
  const float* b1 = set2 ? a.b1b : a.b1;
  // CLAIM (lbm_hip_debug_set key 12 = 1; off by default): the wave claims the SIMD's whole register file (256 + 256), so no
  // wave of another kernel is resident beside it.  Round 1 shipped this as a guard: built with packed-fp32 instructions the
  // kernel came out with a few points perturbed at the 2^-11 level in ~1 frame of 20 of the frame pipeline.  The library is
  // now built without packed-fp32 instructions (Makefile) and the kernel screens 0 of 1500 frames
  // without the claim; same speed either way (13.2 us per launch at 1024 rows).
  if constexpr (CLAIM) asm volatile("" ::: "v255", "a255");

  // ---- every weight fragment this wave will need, requested up front ----
  half8 b1h[NT1][KS1], b1l[NT1][KS1];
#pragma unroll
  for (int j = 0; j < NT1; ++j) {
    const int nb = c * (CH / 32) + w * NT1 + j;
    const _Float16* p = F1 + ((long long)nb * KS1 * 2) * 512 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      b1h[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2) * 512);
      if constexpr (NPL == 2) b1l[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2 + 1) * 512);
    }
  }
  // ---- activation tile -> LDS (rows past M repeat the last row; their results are never stored) ----
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int piece = it * 256 + tid;              // 32 rows x 32 pieces of 16 bytes
      const int row = piece >> 5, pc = piece & 31;
      const int m = min(m0 + row, mend - 1);
      const half8 v = *reinterpret_cast<const half8*>(a.A + pl * a.a_plane + (long long)m * a.lda + pc * 8);
      *reinterpret_cast<half8*>(&sA[pl][row * kAP + pc * 8]) = v;
    }
  // second-layer fragments: tile 0 now, tile 1 half way through phase 1, when half of the first layer's registers are free
  half8 b2h[2][KS2], b2l[2][KS2];
  auto fetch2 = [&](int j) {
    const int nb = w * 2 + j;
    const _Float16* p = F2 + (((long long)nb * (kHid / 16) + c * KS2) * 2) * 512 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      b2h[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2) * 512);
      if constexpr (NPL == 2) b2l[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2 + 1) * 512);
    }
  };
  fetch2(0);
  __syncthreads();

  // ---- phase 1: this wave's NT1 tiles of the hidden piece ----
  f32x16 acc[NT1], acx[NT1];
#pragma unroll
  for (int j = 0; j < NT1; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = acx[j][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) {
    if (ks == KS1 / 2) fetch2(1);
    const half8 ah = *reinterpret_cast<const half8*>(&sA[0][r * kAP + ks * 16 + h * 8]);
    half8 al;
    if constexpr (NPL == 2) al = *reinterpret_cast<const half8*>(&sA[NPL - 1][r * kAP + ks * 16 + h * 8]);
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h[j][ks], acc[j], 0, 0, 0);
      if constexpr (NPL == 2) {
        acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l[j][ks], acx[j], 0, 0, 0);
        acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h[j][ks], acx[j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT1; ++j) {
    const int cl = (w * NT1 + j) * 32 + r;          // hidden channel inside the chunk: this lane's accumulator column
    const float bv = b1[c * CH + cl];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = (e & 3) + 8 * (e >> 2) + 4 * h;
      float v = (NPL == 2 ? acc[j][e] + acx[j][e] * (1.0f / 2048.0f) : acc[j][e]) + bv;
      v = __builtin_amdgcn_fmed3f(gelu_erf(v), -65000.f, 65000.f);
      const _Float16 hh = (_Float16)v;
      sH[0][rl * HP + cl] = hh;
      if constexpr (NPL == 2) sH[NPL - 1][rl * HP + cl] = (_Float16)((v - (float)hh) * 2048.f);
    }
  }
  __syncthreads();

  // ---- phase 2: partial output, two 32-column tiles per wave ----
  f32x16 o[2], ox[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[j][e] = ox[j][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS2; ++ks) {
    const half8 ah = *reinterpret_cast<const half8*>(&sH[0][r * HP + ks * 16 + h * 8]);
    half8 al;
    if constexpr (NPL == 2) al = *reinterpret_cast<const half8*>(&sH[NPL - 1][r * HP + ks * 16 + h * 8]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2h[j][ks], o[j], 0, 0, 0);
      if constexpr (NPL == 2) {
        ox[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2l[j][ks], ox[j], 0, 0, 0);
        ox[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b2h[j][ks], ox[j], 0, 0, 0);
      }
    }
  }
  float* out = a.slabs + (long long)c * a.slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (w * 2 + j) * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < mend) out[(long long)row * kC + col] = NPL == 2 ? o[j][e] + ox[j][e] * (1.0f / 2048.0f) : o[j][e];
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// The same feed-forward for MANY rows (2048 < M <= 16384: the lattice blocks of a dense grid, a batch of streams).
//
// The few-rows kernel above re-reads its 256 KB of weight fragments for every 32 rows; from a few thousand rows on that is more L2
// traffic than the work is worth (111 against 85 us for the two products at 16384 rows).  Here the weights stay put: a workgroup of
// 8 waves owns one 128-channel chunk of the hidden layer for a whole RANGE of rows and streams the rows through in steps of 32.
// Waves 0-3 hold the first layer's fragments of the chunk (32 hidden channels each, all 256 k: 128 VGPRs) and turn a step's rows
// into the [32, 128] hidden piece (GELU, split planes, LDS); waves 4-7 hold the second layer's fragments (two 32-column output tiles
// each over the chunk's 128 k: 128 VGPRs) and multiply the PREVIOUS step's hidden piece into the step's partial [32, 256] output.
// Producer and consumer of a SIMD (waves w and w + 4) share its matrix pipe, so the producer's GELU / split arithmetic sits under
// the consumer's MFMAs (the activation of step k - 1 is applied at the START of step k, while the consumers multiply step k - 2, and
// the producers multiply while the consumers store); hidden piece and row tile are double-buffered in LDS, one barrier per step.  The next
// step's rows are fetched into registers a step ahead.  Output: the same eight K-split slabs as above, bit for bit.
// ------------------------------------------------------------------------------------------------------------------------
int g_ffn_rows_dbg = 0;   // debug key 16: ablation bits of ffn_rows_kernel

struct FfnRowsDev {
  FfnDev f;
  int steps_total, spg;      // 32-row steps in all / per workgroup
  int dbg;                   // tuning (timing only): 1 no activation arithmetic, 2 no slab stores, 4 no hidden-piece LDS writes, 8 no producer MFMAs, 16 no consumer MFMAs
};

__global__ __launch_bounds__(512) void ffn_rows_kernel(FfnRowsDev aa) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CH = 128, KS1 = kC / 16, KS2 = CH / 16, HP = CH + 8, AP = kC + 8;
  __shared__ __attribute__((aligned(16))) _Float16 sA[2][2][kRows * AP];
  __shared__ __attribute__((aligned(16))) _Float16 sH[2][2][kRows * HP];
  const FfnDev& a = aa.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wq = w & 3;
  const bool producer = w < 4;
  const int r = lane & 31, h = lane >> 5;
  const int c = blockIdx.x;
  const int s0 = blockIdx.y * aa.spg;
  const int ns = min(aa.spg, aa.steps_total - s0);
  if (ns <= 0) return;
  auto handover = [&]() __attribute__((always_inline)) {
    // LDS writes of this wave have landed; loads and stores to global memory stay in flight (no __syncthreads: its fence would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (producer) {
    half8 b1h[KS1], b1l[KS1];
    {
      const int nb = c * (CH / 32) + wq;
      const _Float16* p = a.F1 + ((long long)nb * KS1 * 2) * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        b1h[ks] = *reinterpret_cast<const half8*>(p + (ks * 2) * 512);
        b1l[ks] = *reinterpret_cast<const half8*>(p + (ks * 2 + 1) * 512);
      }
    }
    const int cl = wq * 32 + r;                       // hidden channel inside the chunk: this lane's accumulator column
    const float bv = a.b1[c * CH + cl];
    // row tile: 32 rows x 2 planes x 32 pieces of 16 bytes over the 256 producer lanes = 8 pieces per lane
    const int ptid = tid;                             // 0..255
    half8 xr[2][4];
    auto fetch = [&](const int step) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int piece = it * 256 + ptid;
          const int row = piece >> 5, pc = piece & 31;
          const int m = min((s0 + step) * kRows + row, a.M - 1);
          xr[pl][it] = *reinterpret_cast<const half8*>(a.A + pl * a.a_plane + (long long)m * a.lda + pc * 8);
        }
    };
    auto stash = [&](const int b) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int piece = it * 256 + ptid;
          const int row = piece >> 5, pc = piece & 31;
          *reinterpret_cast<half8*>(&sA[b][pl][row * AP + pc * 8]) = xr[pl][it];
        }
    };
    fetch(0);
    stash(0);
    if (ns > 1) fetch(1);
    float sv[16];                                     // step k - 1 before the activation: its GELU runs while the consumers multiply
    for (int k = 0; k <= ns + 1; ++k) {
      handover();
      if (k >= 1 && k <= ns) {
        _Float16* Hh = &sH[(k - 1) & 1][0][cl];
        _Float16* Hl = &sH[(k - 1) & 1][1][cl];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v = (LBM_ABL(aa.dbg & 1)) ? sv[e] : __builtin_amdgcn_fmed3f(gelu_erf(sv[e]), -65000.f, 65000.f);
          const _Float16 hh = (_Float16)v;
          if (!(LBM_ABL(aa.dbg & 4))) {
            Hh[rl * HP] = hh;
            Hl[rl * HP] = (_Float16)((v - (float)hh) * 2048.f);
          }
        }
      }
      if (k >= ns) continue;
      if (k + 1 < ns) {
        stash((k + 1) & 1);                           // (that buffer was last read in step k - 1)
        if (k + 2 < ns) fetch(k + 2);
      }
      const _Float16* Ah = &sA[k & 1][0][r * AP + h * 8];
      const _Float16* Al = &sA[k & 1][1][r * AP + h * 8];
      f32x16 acc, acx;
      half8 fa[2][4];
      auto ldg = [&](const int bsel, const int g) __attribute__((always_inline)) {
        fa[bsel][0] = *reinterpret_cast<const half8*>(Ah + (2 * g) * 16);
        fa[bsel][1] = *reinterpret_cast<const half8*>(Al + (2 * g) * 16);
        fa[bsel][2] = *reinterpret_cast<const half8*>(Ah + (2 * g + 1) * 16);
        fa[bsel][3] = *reinterpret_cast<const half8*>(Al + (2 * g + 1) * 16);
      };
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (!(LBM_ABL(aa.dbg & 8))) ldg(0, 0);
#pragma unroll
      for (int g = 0; g < ((LBM_ABL(aa.dbg & 8)) ? 0 : KS1 / 2); ++g) {
        if (g + 1 < KS1 / 2) ldg((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ks = 2 * g + q;
          const bool first = ks == 0;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], b1h[ks], first ? zero16 : acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], b1l[ks], first ? zero16 : acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q + 1], b1h[ks], acx, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) sv[e] = (acc[e] + acx[e] * (1.0f / 2048.0f)) + bv;
    }
  } else {
    half8 b2h[2][KS2], b2l[2][KS2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = wq * 2 + j;
      const _Float16* p = a.F2 + (((long long)nb * (kHid / 16) + c * KS2) * 2) * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        b2h[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2) * 512);
        b2l[j][ks] = *reinterpret_cast<const half8*>(p + (ks * 2 + 1) * 512);
      }
    }
    float* out = a.slabs + (long long)c * a.slab_stride;
    for (int k = 0; k <= ns + 1; ++k) {
      handover();
      if (k < 2) continue;
      const _Float16* Hh = &sH[k & 1][0][r * HP + h * 8];          // the hidden piece of step k - 2
      const _Float16* Hl = &sH[k & 1][1][r * HP + h * 8];
      f32x16 o[2], ox[2];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      half8 fh[2][4];
      auto ldg = [&](const int bsel, const int g) __attribute__((always_inline)) {
        fh[bsel][0] = *reinterpret_cast<const half8*>(Hh + (2 * g) * 16);
        fh[bsel][1] = *reinterpret_cast<const half8*>(Hl + (2 * g) * 16);
        fh[bsel][2] = *reinterpret_cast<const half8*>(Hh + (2 * g + 1) * 16);
        fh[bsel][3] = *reinterpret_cast<const half8*>(Hl + (2 * g + 1) * 16);
      };
      if (!(LBM_ABL(aa.dbg & 16))) ldg(0, 0);
#pragma unroll
      for (int g = 0; g < ((LBM_ABL(aa.dbg & 16)) ? 0 : KS2 / 2); ++g) {
        if (g + 1 < KS2 / 2) ldg((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ks = 2 * g + q;
          const bool first = ks == 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[g & 1][2 * q], b2h[j][ks], first ? zero16 : o[j], 0, 0, 0);
            ox[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[g & 1][2 * q], b2l[j][ks], first ? zero16 : ox[j], 0, 0, 0);
            ox[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[g & 1][2 * q + 1], b2h[j][ks], ox[j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int m0 = (s0 + k - 2) * kRows;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (wq * 2 + j) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < a.M && !(LBM_ABL(aa.dbg & 2))) out[(long long)row * kC + col] = o[j][e] + ox[j][e] * (1.0f / 2048.0f);
        }
      }
    }
  }
#endif
}

int ffn_rows(const FfnArgs& f, hipStream_t stream) {
  if (f.M == 0) return 0;
  LBM_REQUIRE(f.A && f.F1 && f.b1 && f.F2 && f.slabs && f.M > 0, "ffn_rows: null argument");
  LBM_REQUIRE(f.chunk == 128 && f.planes == 2, "ffn_rows: chunk must be 128, planes 2");
  LBM_REQUIRE(f.lda % 8 == 0 && f.a_plane % 8 == 0 && ((uintptr_t)f.A & 15) == 0 && ((uintptr_t)f.F1 & 15) == 0 && ((uintptr_t)f.F2 & 15) == 0,
              "ffn_rows: operands must admit 16-byte pieces");
  LBM_REQUIRE(f.slab_stride >= (long long)f.M * kC, "ffn_rows: slabs overlap");
  FfnRowsDev d;
  d.f.A = (const _Float16*)f.A; d.f.a_plane = f.a_plane; d.f.lda = f.lda; d.f.F1 = (const _Float16*)f.F1; d.f.b1 = f.b1; d.f.F2 = (const _Float16*)f.F2;
  d.f.slabs = f.slabs; d.f.slab_stride = f.slab_stride; d.f.M = f.M;
  d.f.set_rows = 0; d.f.nb0 = 0; d.f.F1b = nullptr; d.f.b1b = nullptr; d.f.F2b = nullptr;
  LBM_REQUIRE(f.set_rows == 0, "ffn_rows: two weight sets are a feature of the few-rows kernel");
  d.dbg = g_ffn_rows_dbg;
  d.steps_total = (f.M + kRows - 1) / kRows;
  int groups = std::min(d.steps_total, 32);             // 8 chunks x 32 row ranges = one workgroup per CU
  d.spg = (d.steps_total + groups - 1) / groups;
  groups = (d.steps_total + d.spg - 1) / d.spg;
  const dim3 grid(kHid / 128, groups);
  const double flops = 4.0 * f.M * kC * kHid;
  const double bytes = 4.0 * (f.M * kC + 2.0 * kC * kHid) + 4.0 * f.M * kC * 8;
  ProfScope prof("ffn_rows_ss", stream, flops, bytes);
  hipLaunchKernelGGL(ffn_rows_kernel, grid, dim3(512), 0, stream, d);
  return check_launch("ffn_rows");
}

int g_ffn_claim = 0;   // debug key 12: 1 = the wave claims the whole register file (see the kernel)

int ffn_fused(const FfnArgs& f, hipStream_t stream) {
  if (f.M == 0) return 0;
  LBM_REQUIRE(f.A && f.F1 && f.b1 && f.F2 && f.slabs && f.M > 0, "ffn_fused: null argument");
  LBM_REQUIRE(f.chunk == 128, "ffn_fused: chunk must be 128 (a wave keeps all its weight fragments in registers)");
  LBM_REQUIRE(f.planes == 1 || f.planes == 2, "ffn_fused: planes must be 1 or 2");
  LBM_REQUIRE(f.lda % 8 == 0 && f.a_plane % 8 == 0 && ((uintptr_t)f.A & 15) == 0 && ((uintptr_t)f.F1 & 15) == 0 && ((uintptr_t)f.F2 & 15) == 0,
              "ffn_fused: operands must admit 16-byte pieces");
  LBM_REQUIRE(f.slab_stride >= (long long)f.M * kC, "ffn_fused: slabs overlap");
  FfnDev d;
  d.A = (const _Float16*)f.A; d.a_plane = f.a_plane; d.lda = f.lda; d.F1 = (const _Float16*)f.F1; d.b1 = f.b1; d.F2 = (const _Float16*)f.F2;
  d.slabs = f.slabs; d.slab_stride = f.slab_stride; d.M = f.M;
  d.set_rows = 0; d.nb0 = 0; d.F1b = nullptr; d.b1b = nullptr; d.F2b = nullptr;
  int nblk = (f.M + kRows - 1) / kRows;
  if (f.set_rows > 0) {
    LBM_REQUIRE(f.set_rows < f.M && f.F1b && f.b1b && f.F2b && ((uintptr_t)f.F1b & 15) == 0 && ((uintptr_t)f.F2b & 15) == 0,
                "ffn_fused: a two-set launch needs the second weight set and 0 < set_rows < M");
    d.set_rows = f.set_rows; d.nb0 = (f.set_rows + kRows - 1) / kRows;
    d.F1b = (const _Float16*)f.F1b; d.b1b = f.b1b; d.F2b = (const _Float16*)f.F2b;
    nblk = d.nb0 + (f.M - f.set_rows + kRows - 1) / kRows;
  }
  const dim3 grid(kHid / f.chunk, nblk);
  const double flops = 4.0 * f.M * kC * kHid;
  const double bytes = 2.0 * f.planes * (f.M * kC + 2.0 * kC * kHid) + 4.0 * f.M * kC * (kHid / f.chunk);
  ProfScope prof(f.planes == 2 ? "ffn_fused_ss" : "ffn_fused_sh", stream, flops, bytes);
  if (g_ffn_claim) {
    if (f.planes == 2) hipLaunchKernelGGL((ffn_fused_kernel<128, 2, true>), grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((ffn_fused_kernel<128, 1, true>), grid, dim3(256), 0, stream, d);
  } else {
    if (f.planes == 2) hipLaunchKernelGGL((ffn_fused_kernel<128, 2, false>), grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((ffn_fused_kernel<128, 1, false>), grid, dim3(256), 0, stream, d);
  }
  return check_launch("ffn_fused");
}

}  // namespace lbm
