// gemm_rs: "row-streaming" split-precision product for K = 256 — one operand STATIONARY in registers, the other streamed.
//
//   D[s, t] = sum_k S[s, k] * T[t, k]          S: streamed rows (split planes [Ms, 256]),  T: stationary rows (split planes [Nt, 256])
//
// Why a second product kernel: with K = 256 the tiled kernel (gemm16.hip) stages 256 KB per 128 x 128 tile for 2.6 us of
// MFMA work, and every wave spends as long issuing LDS-DMA pieces and fragment reads as it spends in the matrix cores
// (ablation in DESIGN.md section 5: DMA + epilogue are half of the kernel).  Here a workgroup of 8 waves owns 256 rows of T
// for its whole life: wave w keeps the MFMA B-fragments of its 32 rows of T, all 256 k, both planes, in 128 VGPRs.  S
// streams through a 3-stage LDS ring in steps of 32 rows (32 KB per step, LDS-DMA, XOR-swizzled like gemm16.hip); per
// step a wave issues 4 DMA pieces and 32 ds_read_b128 for 48 MFMAs (the tiled kernel: 4 and 12 for 12).  One barrier per
// step.  The epilogue of step s-1 runs out of 16 saved registers while step s is multiplied, and the two waves that share a
// SIMD do it at different points of the step (one before its MFMAs, one in the middle of them), so VALU / LDS / store work
// of one wave sits under the MFMAs of the other.
//
// Two epilogues:
//   EPI 0  linear layer: S = activations [M, 256], T = weights [N, 256]: bias / GELU / residual / ReLU, fp32 and / or split output;
//          every wave transposes its 32 x 32 block through a private LDS patch and writes 16-byte row pieces.
//   EPI 1  correlation + top-k (transformer.py:306-308, coord_utils.py:4-29): S = the correlation map F of a stream [P0, 256], T = the
//          normalised point tokens [N, 256].  Output lane = point, output registers = cells in ascending order, so every lane keeps a
//          sorted list of its KM best (value, cell) in registers across ALL steps of its workgroup — the R x P0 scores exist only in
//          accumulators.  Lists of the cell ranges / half-waves are merged by topk_candidates (pyramid.hip).
// Arithmetic is the tiled kernel's: same three MFMAs per block in the same order, same final combination — results are bit-identical.
#include "gemm.hpp"

#include <cstdlib>
#include <set>
#include <string>

namespace lbm {

struct GemmRSDev {
  const _Float16* S;
  const _Float16* T;
  long long s_plane, t_plane, sS, sT;
  int lds, ldt;
  int Ms, Nt, batch;
  int spc, steps_total, nslice, nchunk;
  int sb, spb;        // slice blocks per group and slices per block (XCD placement, rs_partition)
  const float* bias;
  float* Y;
  _Float16* Ys;
  const float* res;
  long long ys_plane, sBias, sY, sYs, sR;
  int ldy, ldys, ldres, res_div, res_mod;
  float res_scale;
  int act;
  float* cand_v;
  int* cand_i;
  int n_cand;
  int dbg;   // tuning (timing only, results are garbage): 1 no skew between the waves of a SIMD, 2 no DMA after the prologue, 4 no epilogue,
             // 8 no MFMAs, 16 non-temporal stores, 32 no loads of T, 64 return at once, 128 strict waits, 256 stores into 32 rows, 512 no
             // transposition, 1024 transposition without stores
};

typedef __attribute__((address_space(3))) void* lds_ptr_rs;

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter on gfx9");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Output stores of the in-loop epilogue go out through inline asm.  As compiler-visible stores they put a second kind of event on the
// VM counter beside the LDS-DMA loads; hipcc then treats the counter as out of order and drains it (vmcnt(0)) at every register it
// believes pending -- four times per step, which serialises the DMA ring (seen in the ISA).  Completion is covered by the hand-counted
// waits of the step loop; the data registers are protected by the wait states inside the asm string (below).
__device__ __forceinline__ void st16_asm(float* p, const float4 v, const bool nt) {
  const f32x4 d = {v.x, v.y, v.z, v.w};
  // (`s_nop 1` inside the string: a 16-byte store reads its data registers over two more cycles, and hipcc -- which cannot see that
  // this asm is a store -- may rebuild the next value in them at once.  Round 3 met exactly that in the rows-stationary epilogue:
  // columns {0, 1} of some lanes carried the NEXT store's values.)
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void st8_asm(_Float16* p, const half4 d, const bool nt) {
  if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(d) : "memory");
}

// RES: the linear epilogue adds a residual (the only ordinary loads inside the loop: beside them the compiler drains the DMA queue)
// NST: output store instructions a wave issues per step (4 per fp32 output, 8 per split output): lets the step loop's counted wait
// leave the previous epilogue's stores in flight (gfx9 retires the VM counter in issue order)
// NPL: operand planes multiplied.  2 = split operands, three MFMAs per block (fp32-grade); 1 = the hi planes only (plain f16
// operands: the f16-storage mode's value projection) -- the lo planes are neither fetched nor multiplied (EPI 0 only)
// RING: stages of the LDS ring (3 or 4).  4 = steps fetched THREE ahead: the wait at the top of a step then leaves two steps' stores and
// DMA pieces in flight instead of one (the linear epilogue's patches lose their padding so that 4 x 32 KB + 8 x 4 KB is exactly the
// CU's 160 KB)
template <int EPI, int KM, bool RES = false, int NST = 0, int NPL = 2, int RING = 3>
__global__ __launch_bounds__(512) void gemm_rs_kernel(GemmRSDev a) {
  static_assert(NPL == 2 || EPI == 0, "single-plane form: linear epilogue only");
  constexpr int NP = 2 * NPL;                         // LDS-DMA pieces a wave issues per step
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(RING == 3 || RING == 4, "ring of 3 or 4 stages");
  constexpr int STAGE = 32768, NS = RING, D = RING - 1, TP = RING == 4 ? 32 : 36;      // D: steps fetched ahead
  constexpr int LDS_BYTES = NS * STAGE + (EPI == 0 ? 8 * 32 * TP * 4 : 0);
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;

  // Workgroups are dealt round-robin over the 8 XCDs (id % 8).  The workgroups that stream the SAME rows of S (one per 256-row
  // slice of T) are kept on one XCD, dispatched together, so S passes through ONE L2 -- in blocks of spb slices: with fewer
  // groups than XCDs (or a count that is not a multiple of 8) a group's slices are dealt over sb XCDs so that none stays idle.
  const int id = blockIdx.x, xcd = id & 7, tq = id >> 3;
  const int sl = tq % a.spb, unit = (tq / a.spb) * 8 + xcd;
  const int group = unit / a.sb, slice = (unit - group * a.sb) * a.spb + sl;
  if (group >= a.batch * a.nchunk || slice >= a.nslice) return;
  const int z = group / a.nchunk, chunk = group - z * a.nchunk;
  const int s0 = chunk * a.spc;
  const int ns = min(a.spc, a.steps_total - s0);      // <= 0: a chunk past the last step (workgroup-uniform)
  const int n0 = slice * 256 + wave * 32;
  if (ns <= 0) {
    if constexpr (EPI == 1) {
      if (n0 + r < a.Nt) {
        const long long o = ((long long)z * a.Nt + n0 + r) * a.n_cand + (long long)(chunk * 2 + h) * KM;
#pragma unroll
        for (int t = 0; t < KM; ++t) {
          a.cand_v[o + t] = -INFINITY;
          a.cand_i[o + t] = 0x7fffffff;
        }
      }
    }
    return;
  }
  const bool active = n0 < a.Nt;                      // wave-uniform: a ragged last slice leaves waves without rows of T
  const bool skew = ((wave >> 2) & 1) && !(LBM_ABL(a.dbg & 1));
  const bool nt = (LBM_ABL(a.dbg & 16)) != 0;    // tuning: non-temporal output stores   // waves w and w + 4 share a SIMD

  // ---- stationary operand: this wave's 32 rows of T as MFMA B-fragments (lane (r, h): row r, k = 16 ks + 8 h .. + 8) ----
  half8 fbh[16], fbl[16];
  if (LBM_ABL(a.dbg & 64)) return;                 // tuning: launch skeleton only
  if (LBM_ABL(a.dbg & 32)) {                       // tuning: no loads of the stationary operand
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fbh[ks][e] = (_Float16)(0.01f * (lane + ks));
        fbl[ks][e] = (_Float16)(0.02f * (lane - ks));
      }
  } else {
    const int n = min(n0 + r, a.Nt - 1);
    const _Float16* tp = a.T + (long long)z * a.sT + (long long)n * a.ldt + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      fbh[ks] = *reinterpret_cast<const half8*>(tp + ks * 16);
      if constexpr (NPL == 2) fbl[ks] = *reinterpret_cast<const half8*>(tp + a.t_plane + ks * 16);
    }
  }

  // ---- streamed operand: stage = [8 k-chunks of 32][plane][32 rows][64 B]; wave w fetches chunk w (2 planes x 2 half-steps) ----
  const _Float16* Sz = a.S + (long long)z * a.sS + wave * 32;
  const int drow = lane >> 2;
  auto issue = [&](int stage, int step) {
    const int mb = (s0 + step) * 32;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pl = i >> 1, half = i & 1;
      const int row = half * 16 + drow;
      const int slot = (lane & 3) ^ ((row >> 2) & 3);
      const int m = min(mb + row, a.Ms - 1);
      const _Float16* g = Sz + pl * a.s_plane + (long long)m * a.lds + slot * 8;
      unsigned char* dst = lds + stage * STAGE + wave * 4096 + pl * 2048 + half * 1024;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr_rs)dst, 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (ns > j) issue(j, j);

  const int sw = (r >> 2) & 3;
  const int o0 = r * 64 + ((h ^ sw) << 4), o1 = r * 64 + (((2 + h) ^ sw) << 4);

  f32x16 acc, acx;
  float out[16];
  float lv[EPI == 1 ? KM : 1];
  int li[EPI == 1 ? KM : 1];
  if constexpr (EPI == 1) {
#pragma unroll
    for (int t = 0; t < KM; ++t) {
      lv[t] = -INFINITY;
      li[t] = 0x7fffffff;
    }
  }
  float bv = 0.f;
  if constexpr (EPI == 0) {
    if (a.bias && n0 + r < a.Nt) bv = a.bias[(long long)z * a.sBias + n0 + r];
  }

  // One step = 8 groups of two k-steps.  The four fragment reads of group g + 1 are issued BEFORE the six MFMAs of group g and pinned
  // there (sched_barrier): left alone, the compiler re-uses one register set and waits for every read right before its MFMA, which
  // leaves the matrix cores idle for the LDS latency of every k-step (measured: 384 against 520 algorithmic TFLOP/s for the bare loop,
  // scripts/micro/mfma_peak.hip modes 2 and 4).
  half8 fa[2][4];
  auto ldg = [&](const unsigned char* Ab, const int bsel, const int g) __attribute__((always_inline)) {
    const unsigned char* p = Ab + g * 4096;
    fa[bsel][0] = *reinterpret_cast<const half8*>(p + o0);
    if constexpr (NPL == 2) fa[bsel][1] = *reinterpret_cast<const half8*>(p + 2048 + o0);
    fa[bsel][2] = *reinterpret_cast<const half8*>(p + o1);
    if constexpr (NPL == 2) fa[bsel][3] = *reinterpret_cast<const half8*>(p + 2048 + o1);
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto grp = [&](const int g) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ks = 2 * g + q;
      const bool first = g == 0 && q == 0;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], fbh[ks], first ? zero16 : acc, 0, 0, 0);
      if constexpr (NPL == 1) {
      } else if constexpr (EPI == 0) {       // (activations hi x weights lo) then (activations lo x weights hi): the tiled kernel's order
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], fbl[ks], first ? zero16 : acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q + 1], fbh[ks], acx, 0, 0, 0);
      } else {                        // roles swapped (EPI 1: S = map, T = points; EPI 2: S = weights, T = token rows): (stationary hi x streamed lo) then (stationary lo x streamed hi), as there
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q + 1], fbh[ks], first ? zero16 : acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], fbl[ks], acx, 0, 0, 0);
      }
    }
  };
  auto half_step = [&](const unsigned char* Ab, const int g0) __attribute__((always_inline)) {
#pragma unroll
    for (int g = g0; g < g0 + 4; ++g) {
      if (g + 1 < 8) ldg(Ab, (g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      grp(g);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  float* Y = nullptr;
  _Float16* Yh = nullptr;
  const float* res = nullptr;
  if constexpr (EPI == 0) {
    Y = a.Y ? a.Y + (long long)z * a.sY : nullptr;
    Yh = a.Ys ? a.Ys + (long long)z * a.sYs : nullptr;
    res = (RES && a.res) ? a.res + (long long)z * a.sR : nullptr;
  }
  if constexpr (EPI == 2) Y = a.Y + (long long)z * a.sY;
  // epilogue of one finished step, out of the saved registers
  auto epi = [&](const int step) __attribute__((always_inline)) {
    const int mb = (s0 + step) * 32;
    if constexpr (EPI == 0) {
      if (LBM_ABL(a.dbg & 4)) return;
      float* Tw = reinterpret_cast<float*>(lds + NS * STAGE) + wave * (32 * TP);
      if (a.act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = gelu_erf(out[e]);
      }
      if (!(LBM_ABL(a.dbg & 512))) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Tw[(4 * h + (e & 3) + 8 * (e >> 2)) * TP + r] = out[e];
      }
      // The patch is private to the wave and the LDS serves a wave's requests in order, so the transposed reads need no barrier.  They
      // are issued from inline asm: before a compiler-visible ds_read hipcc waits vmcnt(0) here (it cannot tell the patch from the
      // DMA ring), which would drain the ring and the stores four times per step.
      f32x4 tv[4];
      if (LBM_ABL(a.dbg & 512)) {                     // tuning: no transposition (garbage goes out)
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = f32x4{out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]};
      } else {
        const unsigned ta = (unsigned)(uintptr_t)(lds_ptr_rs)(Tw + (lane >> 3) * TP + (lane & 7) * 4);
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_read_b128 %0, %4\n\t"
            "ds_read_b128 %1, %4 offset:%5\n\t"
            "ds_read_b128 %2, %4 offset:%6\n\t"
            "ds_read_b128 %3, %4 offset:%7\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(tv[0]), "=&v"(tv[1]), "=&v"(tv[2]), "=&v"(tv[3])
            : "v"(ta), "n"(8 * TP * 4), "n"(16 * TP * 4), "n"(24 * TP * 4)          // eight rows of the patch per read
            : "memory");
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = i * 8 + (lane >> 3), c4 = lane & 7;
        const int row = mb + rl, col = n0 + c4 * 4;
        const int srow = (LBM_ABL(a.dbg & 256)) ? (row & 31) : row;     // tuning: all stores into 32 rows (L2-resident)
        float4 v = make_float4(tv[i][0], tv[i][1], tv[i][2], tv[i][3]);
        if (row < a.Ms && col < a.Nt && !(LBM_ABL(a.dbg & 1024))) {   // (1024: tuning, transposition without the stores)
          if (RES && res) {
            int rr = row;
            if (a.res_div != 1) rr /= a.res_div;
            if (a.res_mod != 0x7fffffff) rr %= a.res_mod;
            const float4 p = ld4(res + (long long)rr * a.ldres + col);
            v.x += a.res_scale * p.x; v.y += a.res_scale * p.y; v.z += a.res_scale * p.z; v.w += a.res_scale * p.w;
          }
          if (a.act == ACT_RELU_POST) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
          if (Y) st16_asm(Y + (long long)srow * a.ldy + col, v, nt);
          if (Yh) {
            half4 sh, sl;
            split4(v, sh, sl);
            _Float16* yp = Yh + (long long)srow * a.ldys + col;
            st8_asm(yp, sh, nt);
            if (a.ys_plane) st8_asm(yp + a.ys_plane, sl, nt);       // (ys_plane 0: plain f16 output, the hi plane alone)
          }
        }
      }
      asm volatile("" ::: "memory");
    } else if constexpr (EPI == 2) {
      // Rows stationary: lane (r, h) holds, for ITS row n0 + r, the 16 outputs of columns mb + 4 h + (e & 3) + 8 (e >> 2) -- four runs
      // of four consecutive columns, i.e. four 16-byte stores straight out of the accumulator layout: no transposition through LDS.
      // Over the steps a wave walks along its 32 output rows, 128 bytes further per step.
      if (n0 + r < a.Nt) {
        float* yp = Y + (long long)(n0 + r) * a.ldy + mb + 4 * h;
        const float* bp = a.bias ? a.bias + (long long)z * a.sBias + mb + 4 * h : nullptr;
        f32x4 v4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          v4[g4] = f32x4{out[4 * g4], out[4 * g4 + 1], out[4 * g4 + 2], out[4 * g4 + 3]};
          if (bp) {
            const float4 b4 = ld4(bp + 8 * g4);
            v4[g4][0] += b4.x; v4[g4][1] += b4.y; v4[g4][2] += b4.z; v4[g4][3] += b4.w;
          }
        }
        // ONE asm block for the four stores, each from its own register quad, and wait states behind the last: a 16-byte store
        // reads its data registers over several cycles after issue, and the compiler -- which cannot see that this asm is a store
        // -- would otherwise rebuild the next quad in the same registers at once (seen: columns 8 g + {0, 1} of some lanes
        // carrying the NEXT store's values).
        if (nt)
          asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %0, %2, off offset:32 nt\n\t"
                       "global_store_dwordx4 %0, %3, off offset:64 nt\n\tglobal_store_dwordx4 %0, %4, off offset:96 nt\n\ts_nop 3"
                       ::"v"(yp), "v"(v4[0]), "v"(v4[1]), "v"(v4[2]), "v"(v4[3]) : "memory");
        else
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:32\n\t"
                       "global_store_dwordx4 %0, %3, off offset:64\n\tglobal_store_dwordx4 %0, %4, off offset:96\n\ts_nop 3"
                       ::"v"(yp), "v"(v4[0]), "v"(v4[1]), "v"(v4[2]), "v"(v4[3]) : "memory");
      }
      asm volatile("" ::: "memory");
    } else {
      // cells ascend with e inside a lane and with the step: a strict comparison keeps the lowest cell among equal scores.
      // (Also raising the bar to the KM-th best of lane l ^ 32, which holds other cells of the same point, was measured and dropped:
      // the exchange and the second comparison cost more vector issue slots than the skipped insertions return, 453 -> 522 us at C3.)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int idx = mb + 4 * h + (e & 3) + 8 * (e >> 2);
        const float v = out[e];
        if (idx < a.Ms && v > lv[KM - 1]) {
          lv[KM - 1] = v;
          li[KM - 1] = idx;
#pragma unroll
          for (int t = KM - 1; t > 0; --t) {
            if (lv[t] > lv[t - 1]) {
              const float tv = lv[t]; lv[t] = lv[t - 1]; lv[t - 1] = tv;
              const int ti = li[t]; li[t] = li[t - 1]; li[t - 1] = ti;
            }
          }
        }
      }
    }
  };

  // The stationary fragments came by ordinary loads.  Wait for them HERE with a wait the compiler can see: otherwise it must assume
  // them pending on the loop's back edge and drains the whole DMA pipeline (vmcnt(0)) before the first MFMA of every step.
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
  for (int s = 0; s < ns; ++s) {
    // Step s must have landed.  Iteration i of this loop issues, in this order, the NST output stores of the epilogue of step i - 1
    // (i >= 1; full steps only, so their number is fixed) and the NP pieces of step i + D (while there is one); step s itself was
    // issued in iteration s - D (or before the loop).  So behind the pieces of step s this wave's VM queue holds what iterations
    // s - D + 1 .. s - 1 issued: (D - 1) x NST stores and min(D - 1, ns - 1 - s) x NP pieces.  The counter retires in issue order, so
    // "at most that many outstanding" means step s is in while those stay in flight.  Waves without columns issue no stores; with a
    // residual the compiler's own waits for the residual loads drain the queue anyway.
    const int ahead = min(D - 1, ns - 1 - s);          // later steps already requested
    const bool counted = s >= D && active && !RES && NST > 0 && !(LBM_ABL(a.dbg & 128));
    if (counted) {
      if (ahead == D - 1) wait_vm<(D - 1) * (NST + NP)>();
      else if (D == 3 && ahead == 1) wait_vm<(D - 1) * NST + NP>();
      else wait_vm<(D - 1) * NST>();
    } else {
      if (ahead == D - 1) wait_vm<(D - 1) * NP>();
      else if (D == 3 && ahead == 1) wait_vm<NP>();
      else wait_vm<0>();
    }
    if (!(LBM_ABL(a.dbg & 2048))) __builtin_amdgcn_s_barrier();          // every wave's pieces of step s are in; every wave is done reading step s - 1
    const unsigned char* Ab = lds + (s % NS) * STAGE;
    const bool mm = active && !(LBM_ABL(a.dbg & 8));
    if (mm) ldg(Ab, 0, 0);                 // in flight under the epilogue / DMA issue below
    if (!skew) {
      if (active && s > 0) epi(s - 1);
      if (s + D < ns && !(LBM_ABL(a.dbg & 2))) issue((s + D) % NS, s + D);
    }
    if (mm) half_step(Ab, 0);
    if (skew) {
      if (active && s > 0) epi(s - 1);
      if (s + D < ns && !(LBM_ABL(a.dbg & 2))) issue((s + D) % NS, s + D);
    }
    if (mm) half_step(Ab, 4);
    if (active) {
#pragma unroll
      for (int e = 0; e < 16; ++e) out[e] = (NPL == 2 ? acc[e] + acx[e] * (1.0f / 2048.0f) : acc[e]) + bv;
    }
  }
  if (active) epi(ns - 1);

  if constexpr (EPI == 1) {
    if (active && n0 + r < a.Nt) {
      const long long o = ((long long)z * a.Nt + n0 + r) * a.n_cand + (long long)(chunk * 2 + h) * KM;
#pragma unroll
      for (int t = 0; t < KM; ++t) {
        a.cand_v[o + t] = lv[t];
        a.cand_i[o + t] = li[t];
      }
    }
  }
#endif
}

static int g_rs_cus = 256;
int g_rs_ring = 3;     // debug key 25: stages of the LDS ring (3 | 4), bit-identical results; see RING above

// steps of 32 streamed rows per workgroup: about one workgroup per CU, at most `cap` steps each (the 2-step cost of loading the
// stationary fragments is amortised; more, shorter workgroups balance better)
static void rs_partition(GemmRSDev& d, int cap, bool even_chunks) {
  d.steps_total = (d.Ms + 31) / 32;
  d.nslice = (d.Nt + 255) / 256;
  long long nchunk = (long long)g_rs_cus / ((long long)d.nslice * d.batch);
  if (nchunk > d.steps_total) nchunk = d.steps_total;
  const long long need = (d.steps_total + cap - 1) / cap;
  if (nchunk < need) nchunk = need;
  if (nchunk < 1) nchunk = 1;
  if (even_chunks && (nchunk & 1)) ++nchunk;     // (a chunk past the last step is legal: the kernel leaves empty lists for it)
  d.nchunk = (int)nchunk;
  d.spc = (int)((d.steps_total + nchunk - 1) / nchunk);
  const long long groups = nchunk * d.batch;
  d.sb = 1;
  while (d.sb < 8 && d.sb * 2 <= d.nslice && (groups * d.sb) % 8 != 0) d.sb *= 2;
  d.spb = (d.nslice + d.sb - 1) / d.sb;
}
static long long rs_grid(const GemmRSDev& d) {
  const long long units = (long long)d.batch * d.nchunk * d.sb;
  return ((units + 7) / 8) * 8 * d.spb;
}

bool gemm_rs_ok(const GemmSSArgs& a) {
  auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
  if (a.K != 256 || a.topk_k != 0 || a.N % 4 != 0 || a.M < 1) return false;
  // hi planes only: the one shape the f16-storage mode needs -- plain f16 output, no fp32 copy, no residual
  if (a.planes != 2 && !(a.planes == 1 && a.Ys && a.ys_plane == 0 && !a.Y && !a.res)) return false;
  if (a.Y && !(a.ldy % 4 == 0 && a.sY % 4 == 0 && al(a.Y, 16))) return false;
  if (a.Ys && !(a.ldys % 4 == 0 && a.sYs % 4 == 0 && a.ys_plane % 4 == 0 && al(a.Ys, 8))) return false;
  if (a.Ys && a.ys_plane == 0 && a.Y) return false;       // plain f16 output: the store count per step (4) assumes it is the only output
  if (a.res && !(a.ldres % 4 == 0 && a.sR % 4 == 0 && al(a.res, 16))) return false;
  return true;
}

int gemm_rs(const GemmSSArgs& a, int dbg, hipStream_t stream) {
  LBM_REQUIRE(gemm_rs_ok(a), "gemm_rs: shape / alignment not supported (K must be 256, N %% 4 == 0, 16-byte rows)");
  GemmRSDev d{};
  d.S = (const _Float16*)a.A; d.T = (const _Float16*)a.B; d.s_plane = a.a_plane; d.t_plane = a.b_plane; d.sS = a.sA; d.sT = a.sB;
  d.lds = a.lda; d.ldt = a.ldb; d.Ms = a.M; d.Nt = a.N; d.batch = a.batch;
  d.bias = a.bias; d.Y = a.Y; d.Ys = (_Float16*)a.Ys; d.res = a.res; d.ys_plane = a.ys_plane;
  d.sBias = a.sBias; d.sY = a.sY; d.sYs = a.sYs; d.sR = a.sR;
  d.ldy = a.ldy; d.ldys = a.ldys; d.ldres = a.ldres; d.res_div = a.res_div; d.res_mod = a.res_mod; d.res_scale = a.res_scale; d.act = a.act;
  d.dbg = dbg;
  rs_partition(d, 96, false);
  const long long wgs = rs_grid(d);
  LBM_REQUIRE(wgs <= 0x7fffffffLL, "gemm_rs: grid too large");
  static const bool by_shape = getenv("LBM_PROF_SHAPES") != nullptr;
  std::string shaped;
  if (by_shape)
    shaped = std::string(a.planes == 1 ? "gemm_rh " : "gemm_rs ") + std::to_string(a.M) + "x" + std::to_string(a.N) + "x" + std::to_string(a.K) + "x" + std::to_string(a.batch) +
             (a.Y ? " f32" : "") + (a.Ys ? " split" : "");
  const double mnk = (double)a.M * a.N * a.K * a.batch;
  ProfScope prof(by_shape ? intern_name(shaped.c_str()) : a.planes == 1 ? "gemm_rh" : "gemm_rs", stream, 2.0 * mnk,
                 4.0 * ((double)a.M * a.K * a.batch + (double)a.N * a.K * (a.sB || a.batch == 1 ? a.batch : 1) +
                        (double)a.M * a.N * a.batch * ((a.Y ? 1 : 0) + (a.Ys ? 1 : 0) + (a.res ? 1 : 0))));
  const dim3 grid((unsigned)wgs), block(512);
  const bool r4 = g_rs_ring == 4;
#define LBM_RS_LAUNCH(...)                                                                          \
  do {                                                                                              \
    if (r4) hipLaunchKernelGGL((gemm_rs_kernel<__VA_ARGS__, 4>), grid, block, 0, stream, d);        \
    else hipLaunchKernelGGL((gemm_rs_kernel<__VA_ARGS__, 3>), grid, block, 0, stream, d);           \
  } while (0)
  if (a.planes == 1) LBM_RS_LAUNCH(0, 1, false, 4, 1);
  else if (a.res) hipLaunchKernelGGL((gemm_rs_kernel<0, 1, true, 0>), grid, block, 0, stream, d);
  else if (a.Ys && a.ys_plane == 0) LBM_RS_LAUNCH(0, 1, false, 4, 2);     // (never together with Y: see gemm_rs_ok)
  else if (a.Y && a.Ys) LBM_RS_LAUNCH(0, 1, false, 12, 2);
  else if (a.Ys) LBM_RS_LAUNCH(0, 1, false, 8, 2);
  else LBM_RS_LAUNCH(0, 1, false, 4, 2);
  return check_launch("gemm_rs");
}

// Linear layer with the ROWS stationary (EPI 2): a workgroup keeps 256 rows of the activations in registers for its whole life --
// they are read from memory exactly once, by ordinary loads -- and the weight matrix (small: it stays in every L2) streams through
// the LDS ring; a lane owns an output row, so the results leave as 16-byte pieces of whole rows with no transposition.  For products
// with many rows and many columns (the value projection: 1 044 480 x 1536 x 256): there the weights-stationary form reads the
// activations once per 256-column slice (2.36 GB for 1.07) and writes 1-KB pieces 6 KB apart.
bool gemm_rs_rows_ok(const GemmSSArgs& a) {
  auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
  return a.planes == 2 && a.K == 256 && a.topk_k == 0 && a.N % 32 == 0 && a.N >= 32 && a.M >= 256 && a.Y && !a.Ys && !a.res &&
         a.act == ACT_NONE && a.ldy % 4 == 0 && a.sY % 4 == 0 && al(a.Y, 16) && (!a.bias || al(a.bias, 16)) && a.sBias % 4 == 0;
}

int gemm_rs_rows(const GemmSSArgs& a, int dbg, hipStream_t stream) {
  LBM_REQUIRE(gemm_rs_rows_ok(a), "gemm_rs_rows: shape / alignment not supported");
  GemmRSDev d{};
  // roles: streamed S = the weights [N, 256], stationary T = the activation rows [M, 256]
  d.S = (const _Float16*)a.B; d.T = (const _Float16*)a.A; d.s_plane = a.b_plane; d.t_plane = a.a_plane; d.sS = a.sB; d.sT = a.sA;
  d.lds = a.ldb; d.ldt = a.lda; d.Ms = a.N; d.Nt = a.M; d.batch = a.batch;
  d.bias = a.bias; d.sBias = a.sBias; d.Y = a.Y; d.sY = a.sY; d.ldy = a.ldy;
  d.dbg = dbg;
  rs_partition(d, 96, false);
  // (with few row slices the weight rows are cut into chunks so that every CU has work: each chunk writes its own columns)
  const long long wgs = rs_grid(d);
  LBM_REQUIRE(wgs <= 0x7fffffffLL, "gemm_rs_rows: grid too large");
  const double mnk = (double)a.M * a.N * a.K * a.batch;
  ProfScope prof("gemm_rs_rows", stream, 2.0 * mnk, 4.0 * ((double)a.M * a.K * a.batch + (double)a.N * a.K + (double)a.M * a.N * a.batch));
  hipLaunchKernelGGL((gemm_rs_kernel<2, 1, false, 4>), dim3((unsigned)wgs), dim3(512), 0, stream, d);
  return check_launch("gemm_rs_rows");
}

// Correlation + per-point top-k pre-selection.  points: split planes [batch, N, 256]; map: split planes [batch, P0, 256].
// Leaves n_cand = 2 * nchunk * KM candidates (value, cell) per point in cand_v / cand_i [batch * N, n_cand]; *n_cand_out tells the caller.
int gemm_rs_topk_cands(int P0, int N, int batch, int k) {
  GemmRSDev d{};
  d.Ms = P0; d.Nt = N; d.batch = batch;
  rs_partition(d, 96, true);
  const int KM = k == 1 ? 1 : k <= 4 ? 4 : 9;
  return 2 * d.nchunk * KM;
}

int gemm_rs_topk(const GemmSSArgs& a, int dbg, hipStream_t stream) {
  // a.A / M: points, a.B / N: the map (as gemm_ss takes them); roles are swapped here
  LBM_REQUIRE(a.planes == 2 && a.K == 256 && a.topk_k >= 1 && a.topk_k <= 9 && a.cand_v && a.cand_i, "gemm_rs_topk: bad arguments");
  GemmRSDev d{};
  d.S = (const _Float16*)a.B; d.T = (const _Float16*)a.A; d.s_plane = a.b_plane; d.t_plane = a.a_plane; d.sS = a.sB; d.sT = a.sA;
  d.lds = a.ldb; d.ldt = a.lda; d.Ms = a.N; d.Nt = a.M; d.batch = a.batch;
  d.cand_v = a.cand_v; d.cand_i = a.cand_i;
  d.dbg = dbg;
  rs_partition(d, 96, true);
  const int KM = a.topk_k == 1 ? 1 : a.topk_k <= 4 ? 4 : 9;
  d.n_cand = 2 * d.nchunk * KM;
  const long long wgs = rs_grid(d);
  LBM_REQUIRE(wgs <= 0x7fffffffLL, "gemm_rs_topk: grid too large");
  const double mnk = (double)a.M * a.N * a.K * a.batch;
  ProfScope prof(KM == 1 ? "gemm_rs_topk1" : KM == 4 ? "gemm_rs_topk4" : "gemm_rs_topk9", stream, 2.0 * mnk,
                 4.0 * ((double)a.M * a.K * a.batch + (double)a.N * a.K * a.batch) + 8.0 * (double)a.M * a.batch * d.n_cand);
  const dim3 grid((unsigned)wgs), block(512);
  const bool r4 = g_rs_ring == 4;
  if (KM == 1) LBM_RS_LAUNCH(1, 1, false, 0, 2);
  else if (KM == 4) LBM_RS_LAUNCH(1, 4, false, 0, 2);
  else LBM_RS_LAUNCH(1, 9, false, 0, 2);
#undef LBM_RS_LAUNCH
  return check_launch("gemm_rs_topk");
}

}  // namespace lbm
