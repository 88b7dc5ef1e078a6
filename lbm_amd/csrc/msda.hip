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

#ifndef LBM_MSDA_HPW
#define LBM_MSDA_HPW 1       // heads (= waves) per workgroup of the window kernel: 4, 2 or 1 (measured: 1 is fastest in every case, 10 waves per CU by LDS)
#endif
#ifndef LBM_MSDA_CAP
#define LBM_MSDA_CAP 48      // cells of the staged window per head and level (x 256 B); 2 workgroups per CU at 48
#endif

namespace lbm {

__device__ __forceinline__ int sel4(const int (&v)[4], int l) {
  return l == 0 ? v[0] : (l == 1 ? v[1] : (l == 2 ? v[2] : v[3]));
}

// One bilinear tap of one sample, as both kernels evaluate it (same expressions, same order: the windowed kernel is bit-identical
// to the direct one).  (lx, ly) normalised location, (tx, ty) the tap of the 2 x 2 patch; returns the clamped cell and the tap's
// weight wx * wy * w (zero outside the map: padding_mode='zeros', align_corners=False, msdeformattn.py:223-228).
struct Tap {
  int xc, yc;
  float tw;
  bool inb;
};
__device__ __forceinline__ Tap bilinear_tap(float lx, float ly, float w, int wl, int hl, int tx, int ty) {
  const float fw = (float)wl, fh = (float)hl;
  // grid = 2*loc - 1 ; pixel = ((grid + 1) * size - 1) / 2     (grid_sample, align_corners=False)
  const float px = (((2.0f * lx - 1.0f) + 1.0f) * fw - 1.0f) * 0.5f;
  const float py = (((2.0f * ly - 1.0f) + 1.0f) * fh - 1.0f) * 0.5f;
  const float x0 = floorf(px), y0 = floorf(py);
  const float wx = tx ? (px - x0) : (x0 + 1.0f - px);
  const float wy = ty ? (py - y0) : (y0 + 1.0f - py);
  const int xi = (int)x0 + tx, yi = (int)y0 + ty;
  Tap t;
  t.inb = (xi >= 0) & (xi < wl) & (yi >= 0) & (yi < hl);
  t.xc = min(max(xi, 0), wl - 1);
  t.yc = min(max(yi, 0), hl - 1);
  t.tw = t.inb ? wx * wy * w : 0.f;
  return t;
}

// PT: compile-time number of points per level (0 = a.P at run time).  blockIdx.y selects the reference point: the k
// lattice cells of a point are independent gathers, and one workgroup walking them in turn was bound by the latency of
// its ~36 dependent-address loads per cell (106 us at 1024 points x 9 cells); the cheap stage 1 is simply repeated.
template <bool EXPLICIT, int PT, typename VT = float>
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
  const VT* vb = reinterpret_cast<const VT*>(a.value) + (long long)b * a.v_batch_stride + set * a.set_value_off + hd * LBM_HD + cq * 4;
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
#pragma unroll
      for (int p = 0; p < (PT ? PT : P); ++p) {
        const int i = l * P + p;
        const float lx = rx + sm[hd][0][i];
        const float ly = ry + sm[hd][1][i];
        const float w = sm[hd][2][i];
        const Tap t = bilinear_tap(lx, ly, w, wl, hl, tx, ty);
        const float tw = t.tw;
        const float4 v = ld4(vb + (long long)(st + t.yc * wl + t.xc) * a.ldv);
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

// ---- LDS-staged form for the lattice launches (k = 9, 4 reference cells per point) ---------------------------------------------
// The k reference cells of a point are the best cells of ONE correlation row (transformer.py:306-313) -- with a peaked row a
// 3 x 3 / 2 x 2 neighbourhood -- and every one of them samples the same P offsets around itself, so their k * P * 4 taps per level
// fall into a window of a few dozen cells.  The direct kernel fetches every tap from L2 (k * P * 4 x 256 B per level and head) and,
// worse, has all 64 lanes redo the sample's address arithmetic (~45 vector instructions per sample, which is what bounds it).
// Here one workgroup owns a point, one wave a head, and per level the wave
//   1. places a window of <= CAP cells over the bounding box of its taps (cheap: extreme reference cell + extreme offset),
//   2. works out the taps ONCE, one (reference cell, sample) pair per lane -- clamped cell, weight, and whether the cell lies in
//      the window -- and parks (LDS slot | global cell, weight) per tap in LDS,
//   3. stages the window's 256-byte head segments into LDS (each read from L2 once instead of up to k * P times),
//   4. accumulates, for every reference cell in turn, its P samples in the same order and with the same arithmetic as the direct
//      kernel -- bit-identical rows -- reading a tap from the window, or straight from the value map when the learned offsets
//      (unbounded) put it outside.
// Waves never talk to each other: no workgroup barrier anywhere.
template <int K, int PT, int CAP, int HPW, typename VT = float>
__global__ __launch_bounds__(64 * HPW) void msda_window_kernel(MsdaArgs a) {
  constexpr int NI = K * PT;                       // (reference cell, sample) pairs per level
  constexpr int NPASS = (NI + 63) / 64;
  constexpr int NLD = CAP / 4;                     // staging loads per lane and level
  __shared__ float4 win_all[HPW][CAP * 16];  // window: CAP cells x 64 channels of this head
  __shared__ int2 meta_all[HPW][NI * 4];     // per tap: (x >= 0: float4 index into win | x < 0: -(cell + 1), weight bits)
  __shared__ __attribute__((aligned(16))) float smp[HPW][3][64];      // dx, dy, weight per sample
  __shared__ __attribute__((aligned(16))) float ext_all[HPW][32];     // [4 l + w]: offset extremes of level l; [16 + w]: reference extremes
  __shared__ float rf_all[HPW][2][16];       // the K reference points
  const int lane = threadIdx.x & 63;
  constexpr int WPP = LBM_HEADS / HPW;             // workgroups per point (HPW heads = waves each)
  const int hw = threadIdx.x >> 6;                 // wave of the workgroup: indexes the LDS regions
  long long r = blockIdx.x / WPP;                  // the point
  int hg = (int)(blockIdx.x % WPP);                // which group of HPW heads
  if (a.xcd_streams) {                             // XCD c works through streams c, c + 8, ... one at a time (see the direct kernel)
    const int c = (int)(blockIdx.x & 7);
    const long long t = blockIdx.x >> 3;
    const long long per = (long long)a.N * WPP;    // workgroups of one stream
    const long long j = t / per;
    const long long u = t - j * per;
    r = (j * 8 + c) * a.N + u / WPP;
    hg = (int)(u % WPP);
  }
  const int hd = hg * HPW + hw;                    // head
  float4* win = win_all[hw];
  int2* meta = meta_all[hw];
  float* ext = ext_all[hw];
  const int b = (int)(r / a.N);
  constexpr int S = LBM_LEVELS * PT;

  // ---- stage 1: per-sample normalised offset and soft-maxed weight (as the direct kernel), the K reference points ----
  {
    float dx = 0.f, dy = 0.f, logit = -INFINITY;
    if (lane < S) {
      const int l = lane / PT;
      const float* o = a.offlog + r * a.ld_offlog;
      const int oi = (hd * S + lane) * 2;
      dx = o[oi] / (float)sel4(a.lvl_w, l);
      dy = o[oi + 1] / (float)sel4(a.lvl_h, l);
      logit = o[a.logit_col + hd * S + lane];
    }
    float rxl = 0.f, ryl = 0.f;
    if (lane < K) {
      const float* rp = a.ref + (r * K + lane) * 2;
      rxl = fminf(fmaxf(rp[0] / a.net_w, 0.f), 1.f);
      ryl = fminf(fmaxf(rp[1] / a.net_h, 0.f), 1.f);
    }
    const float mx = wave_max(logit);
    const float e = lane < S ? expf(logit - mx) : 0.f;
    const float wt = e / wave_sum(e);
    smp[hw][0][lane] = dx;
    smp[hw][1][lane] = dy;
    smp[hw][2][lane] = wt;
    if (lane < K) {
      rf_all[hw][0][lane] = rxl;
      rf_all[hw][1][lane] = ryl;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // extremes for the window placement, one per lane (a loop over <= 9 LDS values each instead of 20 wave reductions):
  // lanes 0..15 the offsets of level lane / 4 (x lo, x hi, y lo, y hi), lanes 16..19 the reference points
  if (lane < 20) {
    const int w = lane & 3, l = lane >> 2;
    const float* src = lane < 16 ? &smp[hw][w >> 1][l * PT] : &rf_all[hw][w >> 1][0];
    const int n = lane < 16 ? PT : K;
    float v = src[0];
    for (int i = 1; i < n; ++i) v = (w & 1) ? fmaxf(v, src[i]) : fminf(v, src[i]);
    ext[lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float4 rext = reinterpret_cast<const float4*>(ext)[4];              // rx lo, rx hi, ry lo, ry hi

  const int tap = lane >> 4, cq = lane & 15;
  const VT* vb = reinterpret_cast<const VT*>(a.value) + (long long)b * a.v_batch_stride + hd * LBM_HD + cq * 4;
  float4 acc[K];
#pragma unroll
  for (int kk = 0; kk < K; ++kk) acc[kk] = make_float4(0.f, 0.f, 0.f, 0.f);

  // Window of a level: bounding box of the taps from the extreme reference points and the extreme offsets (a heuristic only --
  // whether a tap is inside is decided exactly, per tap, when the tap table is built).  At most 8 cells wide (rows of the LDS
  // copy have a pitch of 8 cells: no division anywhere) and CAP / 8 high; a box beyond that is covered from its low corner, and
  // one that is hopelessly large (scattered reference cells: a flat correlation row) is not staged at all.
  struct Place { int bx0, by0, WB, HB; };
  auto place = [&](int l) {
    const int hl = a.lvl_h[l], wl = a.lvl_w[l];
    const float4 e4 = reinterpret_cast<const float4*>(ext)[l];              // dx lo, dx hi, dy lo, dy hi
    int bx0 = (int)floorf((rext.x + e4.x) * (float)wl - 0.5f), bx1 = (int)floorf((rext.y + e4.y) * (float)wl - 0.5f) + 1;
    int by0 = (int)floorf((rext.z + e4.z) * (float)hl - 0.5f), by1 = (int)floorf((rext.w + e4.w) * (float)hl - 0.5f) + 1;
    bx0 = min(max(bx0, 0), wl - 1); bx1 = min(max(bx1, 0), wl - 1);
    by0 = min(max(by0, 0), hl - 1); by1 = min(max(by1, 0), hl - 1);
    Place pl{bx0, by0, bx1 - bx0 + 1, by1 - by0 + 1};
    if ((long long)pl.WB * pl.HB > 4 * CAP) {
      pl.WB = 0; pl.HB = 0;
    } else {
      pl.WB = min(pl.WB, 8);
      pl.HB = min(pl.HB, CAP / 8);
    }
    return pl;
  };
  // The window's loads: 16 lanes fetch one cell's 256-byte head segment, four cells (half a window row) per instruction.
  // Branch-free on purpose -- under a per-lane `if` the compiler waits for each load in turn; slots outside WB x HB re-read the
  // window's last cell and are never referenced.  Issued one level AHEAD (registers), so their latency passes behind the tap
  // table and the accumulation of the level before.
  decltype(ldraw4(static_cast<const VT*>(nullptr))) stg[NLD];      // raw: f16 maps are widened when the window is written
  auto fetch = [&](int l, const Place& pl) {
    const int wl = a.lvl_w[l], st = a.lvl_start[l];
    if (pl.WB > 0 && !(LBM_ABL(a.dbg & 2))) {
      const int WBs = __builtin_amdgcn_readfirstlane(pl.WB), HBs = __builtin_amdgcn_readfirstlane(pl.HB);
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        // half row `it` of the window: skipped (a scalar branch) when the window does not reach it
        if ((it >> 1) < HBs && (it & 1) * 4 < WBs) {
          const int c = it * 4 + tap;
          const int cy = c >> 3, cx = min(c & 7, pl.WB - 1);
          stg[it] = ldraw4(vb + (long long)(st + (pl.by0 + cy) * wl + pl.bx0 + cx) * a.ldv);
        }
      }
    }
  };
  Place nxt = place(0);
  fetch(0, nxt);

#pragma unroll 1
  for (int l = 0; l < LBM_LEVELS; ++l) {
    const int hl = a.lvl_h[l], wl = a.lvl_w[l], st = a.lvl_start[l];
    const Place pl = nxt;
    // 1. this level's window into LDS (fetched during the previous level), the next level's on its way
    if (pl.WB > 0 && !(LBM_ABL(a.dbg & 2))) {
#pragma unroll
      for (int it = 0; it < NLD; ++it) win[(it * 4 + tap) * 16 + cq] = widen4(stg[it]);
    }
    if (l + 1 < LBM_LEVELS) {
      nxt = place(l + 1);
      fetch(l + 1, nxt);
    }
    if (LBM_ABL(a.dbg & 1))                       // (ablation: a valid, constant tap table -- slot 0, weight 0)
      for (int i = lane; i < NI * 4; i += 64) meta[i] = make_int2(0, 0);
    // 2. the taps, once: lane = (reference cell, sample)
    // (not unrolled: with both passes of k = 9 interleaved the kernel needs 184 registers -- two waves per SIMD; 166 leave room for
    // a third: 729 -> 719 us on clustered cells at C3 size)
#pragma unroll 1
    for (int ps = 0; ps < (LBM_ABL(a.dbg & 1) ? 0 : NPASS); ++ps) {
      const int item = ps * 64 + lane;
      if (item < NI) {
        const int kk = item / PT, p = item - kk * PT;
        const int i = l * PT + p;
        const float lx = rf_all[hw][0][kk] + smp[hw][0][i];
        const float ly = rf_all[hw][1][kk] + smp[hw][1][i];
        const float w = smp[hw][2][i];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const Tap t = bilinear_tap(lx, ly, w, wl, hl, t4 & 1, t4 >> 1);
          const int ux = t.xc - pl.bx0, uy = t.yc - pl.by0;
          const bool in_win = (ux >= 0) & (ux < pl.WB) & (uy >= 0) & (uy < pl.HB);
          const int code = in_win ? (uy * 8 + ux) * 16 : -(st + t.yc * wl + t.xc + 1);
          meta[item * 4 + t4] = make_int2(code, __float_as_int(t.tw));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 3. accumulate: samples in order, the K reference cells side by side (K reads in flight; every accumulator sees its
    //    samples in the direct kernel's order).  Two forms of the step, chosen per sample for the whole wave: every tap in the
    //    window -> LDS reads only; otherwise every lane reads both (the value-map read of an in-window lane goes to the level's
    //    first cell, a line every lane shares) and selects -- no per-lane branch, whose two sides would write the same registers
    //    and make every LDS read wait for the loads before it.
#pragma unroll 1
    for (int p = 0; p < (LBM_ABL(a.dbg & 4) ? 0 : PT); ++p) {
      int2 m[K];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) m[kk] = meta[(kk * PT + p) * 4 + tap];
      int any = 0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) any |= m[kk].x;
      float4 v[K];
      if (pl.WB == 0) {                            // nothing staged on this level: the direct gather, with the taps ready-made
#pragma unroll
        for (int kk = 0; kk < K; ++kk) v[kk] = ld4(vb + (long long)(-m[kk].x - 1) * a.ldv);
      } else if (__builtin_amdgcn_ballot_w64(any < 0) == 0) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) v[kk] = win[m[kk].x + cq];
      } else {
        float4 g[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) g[kk] = ld4(vb + (long long)(m[kk].x < 0 ? -m[kk].x - 1 : st) * a.ldv);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
          const float4 w4 = win[max(m[kk].x, 0) + cq];
          v[kk] = m[kk].x < 0 ? g[kk] : w4;
        }
      }
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        const float tw = __int_as_float(m[kk].y);
        acc[kk].x += tw * v[kk].x;
        acc[kk].y += tw * v[kk].y;
        acc[kk].z += tw * v[kk].z;
        acc[kk].w += tw * v[kk].w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // the next level overwrites window and tap table
  }
#pragma unroll
  for (int kk = 0; kk < K; ++kk) {
    float4 s4 = acc[kk];
    s4.x += __shfl_xor(s4.x, 16); s4.y += __shfl_xor(s4.y, 16); s4.z += __shfl_xor(s4.z, 16); s4.w += __shfl_xor(s4.w, 16);
    s4.x += __shfl_xor(s4.x, 32); s4.y += __shfl_xor(s4.y, 32); s4.z += __shfl_xor(s4.z, 32); s4.w += __shfl_xor(s4.w, 32);
    if (lane < 16) {
      const long long oo = (r * K + kk) * LBM_C + hd * LBM_HD + cq * 4;
      if (a.out) st4(a.out + oo, s4);
      if (a.out_s) st4_split(a.out_s, a.out_plane, oo, s4);
    }
  }
}

// ---- single reference cell (k = 1: last layer, the three heads) with the taps worked out once --------------------------------
// The direct kernel has all 64 lanes redo every sample's address arithmetic (~45 vector instructions per sample, the bound of
// that kernel); here lane = sample does it once, parks (cell, weight) per tap in LDS, and the gather loop is a table read, a
// load and four multiply-adds per sample.  Same expressions, same order of the sums: bit-identical rows.  One wave (= head) per
// workgroup, or HPW of them; two-module launches (set_rows) as in the direct kernel.
template <int PT, int HPW, typename VT = float>
__global__ __launch_bounds__(64 * HPW) void msda_taps_kernel(MsdaArgs a) {
  constexpr int S = LBM_LEVELS * PT;
  constexpr int WPP = LBM_HEADS / HPW;
  __shared__ int2 meta_all[HPW][S * 4];
  const int lane = threadIdx.x & 63;
  const int hw = threadIdx.x >> 6;
  long long rq = blockIdx.x / WPP;                 // output row
  int hg = (int)(blockIdx.x % WPP);
  if (a.xcd_streams) {
    const int c = (int)(blockIdx.x & 7);
    const long long t = blockIdx.x >> 3;
    const long long per = (long long)a.N * WPP;
    const long long j = t / per;
    const long long u = t - j * per;
    rq = (j * 8 + c) * a.N + u / WPP;              // (two-module launches: unit j * 8 + c runs over (module, stream))
    hg = (int)(u % WPP);
  }
  const int hd = hg * HPW + hw;
  int2* meta = meta_all[hw];
  const int set = (a.set_rows != 0 && rq >= a.set_rows) ? 1 : 0;
  const long long r = rq - (long long)set * a.set_rows;
  const int b = (int)(r / a.N);
  {
    float dx = 0.f, dy = 0.f, logit = -INFINITY;
    int l = 0;
    if (lane < S) {
      l = lane / PT;
      const float* o = a.offlog + r * a.ld_offlog + set * a.set_offlog_off;
      const int oi = (hd * S + lane) * 2;
      dx = o[oi] / (float)sel4(a.lvl_w, l);
      dy = o[oi + 1] / (float)sel4(a.lvl_h, l);
      logit = o[a.logit_col + hd * S + lane];
    }
    const float* rp = a.ref + r * 2;
    const float rx = fminf(fmaxf(rp[0] / a.net_w, 0.f), 1.f);
    const float ry = fminf(fmaxf(rp[1] / a.net_h, 0.f), 1.f);
    const float mx = wave_max(logit);
    const float e = lane < S ? expf(logit - mx) : 0.f;
    const float wt = e / wave_sum(e);
    if (lane < S) {
      const int hl = sel4(a.lvl_h, l), wl = sel4(a.lvl_w, l), st = sel4(a.lvl_start, l);
      const float lx = rx + dx, ly = ry + dy;
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const Tap t = bilinear_tap(lx, ly, wt, wl, hl, t4 & 1, t4 >> 1);
        meta[lane * 4 + t4] = make_int2(st + t.yc * wl + t.xc, __float_as_int(t.tw));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int tap = lane >> 4, cq = lane & 15;
  const VT* vb = reinterpret_cast<const VT*>(a.value) + (long long)b * a.v_batch_stride + set * a.set_value_off + hd * LBM_HD + cq * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int i0 = 0; i0 < S; i0 += PT) {
    int2 m[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) m[p] = meta[(i0 + p) * 4 + tap];
    float4 v[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) v[p] = ld4(vb + (long long)m[p].x * a.ldv);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const float tw = __int_as_float(m[p].y);
      acc.x += tw * v[p].x;
      acc.y += tw * v[p].y;
      acc.z += tw * v[p].z;
      acc.w += tw * v[p].w;
    }
  }
  acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
  if (lane < 16) {
    const long long oo = rq * LBM_C + hd * LBM_HD + cq * 4;
    if (a.out) st4(a.out + oo, acc);
    if (a.out_s) st4_split(a.out_s, a.out_plane, oo, acc);
  }
}

int g_msda_xcd = 1;   // debug key 21: 0 = dispatch order
int g_msda_dbg = 0;      // debug key 23 (tuning build): ablation bits of the window kernel
int g_msda_window = 3;   // debug key 22, bits: 1 = LDS-staged kernel for the k = 9 / 4 lattice launches, 2 = tap-table kernel for k = 1; 0 = always the direct kernel (default 3)

int msda_sample(const MsdaArgs& a, hipStream_t stream) {
  const long long R = (long long)a.B * a.N;
  if (R == 0) return 0;
  LBM_REQUIRE(a.value && (a.out || a.out_s), "msda_sample: null value/out");
  LBM_REQUIRE(a.P >= 1 && a.P <= 16, "msda_sample: P=%d must be in [1,16]", a.P);
  LBM_REQUIRE(a.k >= 1, "msda_sample: k=%d", a.k);
  LBM_REQUIRE(a.ldv % 4 == 0 && a.v_batch_stride % 4 == 0 && ((uintptr_t)a.value & (a.value_f16 ? 7 : 15)) == 0, "msda_sample: value rows must be 16-byte (f16: 8-byte) aligned");
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
  const bool use_window = !explicit_mode && (g_msda_window & 1) && a.set_rows == 0 && a.P == 9 && (a.k == 9 || a.k == 4);
  const bool use_taps = !explicit_mode && (g_msda_window & 2) && a.k == 1 && (a.P == 9 || a.P == 4);
  ProfScope prof(use_window ? "msda_window" : use_taps ? "msda_taps" : "msda_sample", stream, nq * 16.0 * a.P * 4 * 64 * 2, nq * (16.0 * a.P * 4 * 64 * 4 + 4.0 * LBM_C));
  if (explicit_mode) {
    LBM_REQUIRE(!a.value_f16, "msda_sample: explicit mode reads fp32 value maps");
    LBM_REQUIRE(a.attw != nullptr && a.k == 1, "msda_sample: explicit mode needs attw and k == 1");
    hipLaunchKernelGGL((msda_sample_kernel<true, 0>), dim3((unsigned)R, 1), dim3(256), 0, stream, a);
  } else {
    LBM_REQUIRE(a.offlog && a.ref && a.net_w > 0.f && a.net_h > 0.f, "msda_sample: implicit mode needs offlog/ref/net size");
    LBM_REQUIRE(a.ld_offlog >= a.logit_col + 16 * a.P && a.logit_col >= 32 * a.P, "msda_sample: offlog layout too small");
    LBM_REQUIRE(a.k <= 65535, "msda_sample: k=%d too large", a.k);
    LBM_REQUIRE(a.set_rows == 0 || a.set_rows == R, "msda_sample: a two-module launch covers every point twice (set_rows == B * N)");
    const dim3 grid((unsigned)(a.set_rows ? 2 * R : R), (unsigned)a.k);
    MsdaArgs b = a;
    b.dbg = g_msda_dbg;
    b.xcd_streams = (g_msda_xcd && a.B >= 8 && a.B % 8 == 0) ? 1 : 0;
    if (use_window) {
      constexpr int HPW = LBM_MSDA_HPW;
      LBM_REQUIRE(R * (LBM_HEADS / HPW) <= 0x7fffffffLL, "msda_sample: too many rows");
      const dim3 wgrid((unsigned)(R * (LBM_HEADS / HPW)));
      if (a.value_f16) {
        if (a.k == 9) hipLaunchKernelGGL((msda_window_kernel<9, 9, LBM_MSDA_CAP, HPW, _Float16>), wgrid, dim3(64 * HPW), 0, stream, b);
        else hipLaunchKernelGGL((msda_window_kernel<4, 9, LBM_MSDA_CAP, HPW, _Float16>), wgrid, dim3(64 * HPW), 0, stream, b);
      } else if (a.k == 9) hipLaunchKernelGGL((msda_window_kernel<9, 9, LBM_MSDA_CAP, HPW>), wgrid, dim3(64 * HPW), 0, stream, b);
      else hipLaunchKernelGGL((msda_window_kernel<4, 9, LBM_MSDA_CAP, HPW>), wgrid, dim3(64 * HPW), 0, stream, b);
      return check_launch("msda_window");
    }
    if (use_taps) {
      const long long rows = a.set_rows ? 2 * R : R;
      LBM_REQUIRE(rows * LBM_HEADS <= 0x7fffffffLL, "msda_sample: too many rows");
      const dim3 tgrid((unsigned)(rows * LBM_HEADS));
      if (a.value_f16) {
        if (a.P == 9) hipLaunchKernelGGL((msda_taps_kernel<9, 1, _Float16>), tgrid, dim3(64), 0, stream, b);
        else hipLaunchKernelGGL((msda_taps_kernel<4, 1, _Float16>), tgrid, dim3(64), 0, stream, b);
      } else if (a.P == 9) hipLaunchKernelGGL((msda_taps_kernel<9, 1>), tgrid, dim3(64), 0, stream, b);
      else hipLaunchKernelGGL((msda_taps_kernel<4, 1>), tgrid, dim3(64), 0, stream, b);
      return check_launch("msda_taps");
    }
    if (a.value_f16) hipLaunchKernelGGL((msda_sample_kernel<false, 0, _Float16>), grid, dim3(256), 0, stream, b);   // (a fallback only: the f16 mode's launches are window / tap-table shapes)
    else if (a.P == 9) hipLaunchKernelGGL((msda_sample_kernel<false, 9>), grid, dim3(256), 0, stream, b);
    else if (a.P == 4) hipLaunchKernelGGL((msda_sample_kernel<false, 4>), grid, dim3(256), 0, stream, b);
    else hipLaunchKernelGGL((msda_sample_kernel<false, 0>), grid, dim3(256), 0, stream, b);
  }
  return check_launch("msda_sample");
}

}  // namespace lbm
