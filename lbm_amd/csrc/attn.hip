// Single-query multi-head attention read-out over a short per-point sequence.
//
// Restates nn.MultiheadAttention for a length-1 query (transformer.py:272, 294, 333) without ever
// projecting the keys/values:  score_jh = u_h · (seq_j + temb_j)   with u_h = Wk_hᵀ (Wq_h x + bq_h)/8
// (the key bias adds the same constant to every j and cancels in the softmax), and
//   out_h = Σ_j softmax_j(score_jh) · seq_j      (the value projection Wv_h is applied afterwards by a GEMM,
// which is exact because the softmax weights sum to one).  The per-point sequence (M memory slots of a
// ring buffer, or the k lattice tokens) is therefore read exactly once from HBM: 2·M·C·4 bytes per point
// and layer for the two memories — the algorithmic minimum of SURVEY §8(d).
//
// Lane mapping: a wave reads one 1 KiB sequence row per load instruction, lane l owning channels 4l .. 4l+3
// (every byte fetched once, 16 B per lane, fully coalesced).  Four slots are in flight per wave; their
// 4 slots x 4 heads = 16 partial dot products are summed over the wave by a butterfly reduce-scatter
// (17 cross-lane moves instead of 96) and come back as wave-uniform scalars, so the soft-max bookkeeping
// is shared by all lanes.  Softmax is online (running max / sum), so any S works.
// WPP = 4 (few points): the four waves of a workgroup share one point, each taking every fourth slot, and merge
// their (max, sum, accumulator) states through LDS — with ~1000 points one wave per point leaves the chip idle.
#include "attn.hpp"

namespace lbm {

// sum p[0..15] over the 64 lanes; afterwards tot[q] (wave-uniform) holds the total of p[q]
__device__ __forceinline__ float reduce16_lane(const float (&p)[16], int lane);
__device__ __forceinline__ void reduce16(const float (&p)[16], float (&tot)[16], int lane) {
  const float a1 = reduce16_lane(p, lane);
  // lane with bits (5,4,3,2) = (q3,q2,q1,q0) holds the total of index q = 8 q3 + 4 q2 + 2 q1 + q0
#pragma unroll
  for (int q = 0; q < 16; ++q) tot[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), q << 2));
}
// the same, left where the butterfly puts it: lane L holds the total of index (L >> 2) & 15
__device__ __forceinline__ float reduce16_lane(const float (&p)[16], int lane) {
  float a8[8], a4[4], a2[2], a1;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (b5 ? p[i + 8] : p[i]) + __shfl_xor(b5 ? p[i] : p[i + 8], 32);
#pragma unroll
  for (int i = 0; i < 4; ++i) a4[i] = (b4 ? a8[i + 4] : a8[i]) + __shfl_xor(b4 ? a8[i] : a8[i + 4], 16);
#pragma unroll
  for (int i = 0; i < 2; ++i) a2[i] = (b3 ? a4[i + 2] : a4[i]) + __shfl_xor(b3 ? a4[i] : a4[i + 2], 8);
  a1 = (b2 ? a2[1] : a2[0]) + __shfl_xor(b2 ? a2[0] : a2[1], 4);
  a1 += __shfl_xor(a1, 2);
  a1 += __shfl_xor(a1, 1);
  return a1;
}

// XL: the group's exponentials evaluated across the lanes (see the loop); false = every lane evaluates all twenty (the first form,
// kept for the bit-identity test: debug key 3, value + 16)
template <int WPP, bool TEMB, bool XL = true, typename ST = float>
__global__ __launch_bounds__(256) void attn_readout_kernel(AttnArgs a) {
  __shared__ float sm_acc[WPP == 1 ? 1 : 4][LBM_HEADS][LBM_C];
  __shared__ float sm_ml[WPP == 1 ? 1 : 4][2][LBM_HEADS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // WPP waves share a point (4 / WPP points per workgroup); sub = this wave's place among them
  const int row_ = blockIdx.x * (4 / WPP) + wave / WPP;
  const int sub = wave % WPP;
  if (WPP == 1 && row_ >= a.R) return;
  const bool live = row_ < a.R;                    // (WPP > 1: a wave past the end works on the last row and stores nothing --
  const int row = live ? row_ : a.R - 1;           //  every wave of the workgroup reaches the barrier of the merge)

  float4 u[LBM_HEADS];
#pragma unroll
  for (int hh = 0; hh < LBM_HEADS; ++hh) u[hh] = ld4(a.U + (long long)row * a.ldu + hh * LBM_C + lane * 4);

  const ST* sp = reinterpret_cast<const ST*>(a.seq) + (long long)row * a.point_stride + lane * 4;
  const uint8_t* ma = a.mask_a ? a.mask_a + (long long)row * a.S : nullptr;
  const uint8_t* mb = a.mask_b ? a.mask_b + (long long)row * a.S : nullptr;

  float m[LBM_HEADS], l[LBM_HEADS];
  float m_lane = -INFINITY;                      // m[(lane >> 2) & 3], kept per lane for the group soft-max
  float4 acc[LBM_HEADS];
#pragma unroll
  for (int hh = 0; hh < LBM_HEADS; ++hh) {
    m[hh] = -INFINITY;
    l[hh] = 0.f;
    acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // this wave's slots: j0, j0 + WPP, ...  in groups of four; the loads of group g + 1 (rows and their temporal
  // embeddings) are issued before group g is reduced, so they fly while the soft-max arithmetic runs.  The ignore
  // masks are read 64 slots at a time, one byte per lane, and kept as a wave-uniform bit set: a per-slot byte load
  // would have to return before the slot's row load could be issued behind it (loads retire in order), which
  // serialised the whole read-out.  A group never straddles a 64-slot chunk (4·WPP divides 64).
  const int step = WPP;
  const int jb0 = sub;
  unsigned long long ign = 0;
  int ign_chunk = -1;
  auto load_masks = [&](int chunk) {
    const int jl = chunk * 64 + lane;
    const bool in = jl < a.S;
    int phys = a.ring_head + (in ? jl : 0);
    if (phys >= a.S) phys -= a.S;
    bool ig = !in;
    if (ma) ig |= ma[phys] != 0;
    if (mb) ig |= mb[phys] != 0;
    ign = __ballot(ig);
    ign_chunk = chunk;
  };
  float4 vn[4], tn[4];
  bool usen[4];
  auto fetch = [&](int jb) {
    if ((jb >> 6) != ign_chunk) load_masks(jb >> 6);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = jb + g * step;
      const int jc = j < a.S ? j : 0;
      int phys = a.ring_head + jc;
      if (phys >= a.S) phys -= a.S;
      usen[g] = !((ign >> (j & 63)) & 1ull);
      if (TEMB) tn[g] = ld4(a.temb + (long long)jc * LBM_C + lane * 4);
      vn[g] = ld4(sp + (long long)phys * a.slot_stride);
    }
  };
  if (jb0 < a.S) fetch(jb0);
  for (int jb = jb0; jb < a.S; jb += 4 * step) {
    float4 v[4], kxs[4];
    bool use[4];
    float p[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v[g] = vn[g];
      use[g] = usen[g];
      kxs[g] = TEMB ? make_float4(vn[g].x + tn[g].x, vn[g].y + tn[g].y, vn[g].z + tn[g].z, vn[g].w + tn[g].w) : vn[g];
    }
    if (jb + 4 * step < a.S) fetch(jb + 4 * step);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 kx = kxs[g];
#pragma unroll
      for (int hh = 0; hh < LBM_HEADS; ++hh) p[g * 4 + hh] = u[hh].x * kx.x + u[hh].y * kx.y + u[hh].z * kx.z + u[hh].w * kx.w;
    }
    if constexpr (XL) {
    // Soft-max bookkeeping of the group.  The 16 scores come out of the butterfly one per lane quad (lane L: slot (L >> 4) & 3,
    // head (L >> 2) & 3), so the 16 weights exp(s - max) and the 4 rescale factors exp(m_old - max) are TWO exponentials
    // evaluated across the lanes and then broadcast as scalars -- not twenty evaluated by all 64 lanes alike, which made the
    // kernel instruction-bound (~450 vector instructions per 4 KB of memory read) rather than HBM-bound.  Same values, same
    // order of the sums: bit-identical.
    const float sl = reduce16_lane(p, lane);
    const int gl = (lane >> 4) & 3;
    const bool use_l = gl == 0 ? use[0] : (gl == 1 ? use[1] : (gl == 2 ? use[2] : use[3]));
    float mx = use_l ? sl : -INFINITY;
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn_l = fmaxf(m_lane, mx);             // running maximum of this lane's head
    const float w_l = expf(sl - mn_l);
    const float sc_l = expf(m_lane - mn_l);           // m = -inf -> 0
    m_lane = mn_l;
    auto bcast = [](float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); };
#pragma unroll
    for (int hh = 0; hh < LBM_HEADS; ++hh) {
      const float mn = bcast(mn_l, hh << 2);
      if (mn == -INFINITY) continue;             // nothing attended so far (wave-uniform)
      const float sc = bcast(sc_l, hh << 2);
      float ls = l[hh] * sc;
      float4 ac = make_float4(acc[hh].x * sc, acc[hh].y * sc, acc[hh].z * sc, acc[hh].w * sc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (use[g]) {                            // wave-uniform: masks are per (row, slot)
          const float w = bcast(w_l, (g << 4) | (hh << 2));
          ls += w;
          ac.x += w * v[g].x; ac.y += w * v[g].y; ac.z += w * v[g].z; ac.w += w * v[g].w;
        }
      }
      m[hh] = mn;
      l[hh] = ls;
      acc[hh] = ac;
    }
    } else {
    float s[16];
    reduce16(p, s, lane);
#pragma unroll
    for (int hh = 0; hh < LBM_HEADS; ++hh) {
      float mn = m[hh];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (use[g]) mn = fmaxf(mn, s[g * 4 + hh]);
      if (mn == -INFINITY) continue;             // nothing attended so far (wave-uniform)
      const float sc = expf(m[hh] - mn);         // m = -inf -> 0
      float ls = l[hh] * sc;
      float4 ac = make_float4(acc[hh].x * sc, acc[hh].y * sc, acc[hh].z * sc, acc[hh].w * sc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (use[g]) {                            // wave-uniform: masks are per (row, slot)
          const float w = expf(s[g * 4 + hh] - mn);
          ls += w;
          ac.x += w * v[g].x; ac.y += w * v[g].y; ac.z += w * v[g].z; ac.w += w * v[g].w;
        }
      }
      m[hh] = mn;
      l[hh] = ls;
      acc[hh] = ac;
    }
    }
  }

  if (WPP > 1) {   // merge the online-softmax states of the point's WPP waves (in wave order: fixed summation order)
#pragma unroll
    for (int hh = 0; hh < LBM_HEADS; ++hh) {
      st4(&sm_acc[wave][hh][lane * 4], acc[hh]);
      if (lane == 0) {
        sm_ml[wave][0][hh] = m[hh];
        sm_ml[wave][1][hh] = l[hh];
      }
    }
    __syncthreads();
    if (sub != 0 || !live) return;
#pragma unroll
    for (int hh = 0; hh < LBM_HEADS; ++hh) {
      float mm = -INFINITY;
#pragma unroll
      for (int w2 = 0; w2 < WPP; ++w2) mm = fmaxf(mm, sm_ml[wave + w2][0][hh]);
      float ls = 0.f;
      float4 ac = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mm != -INFINITY) {
#pragma unroll
        for (int w2 = 0; w2 < WPP; ++w2) {
          const float mw = sm_ml[wave + w2][0][hh];
          if (mw == -INFINITY) continue;
          const float sc = expf(mw - mm);
          ls += sm_ml[wave + w2][1][hh] * sc;
          const float4 q = ld4(&sm_acc[wave + w2][hh][lane * 4]);
          ac.x += q.x * sc; ac.y += q.y * sc; ac.z += q.z * sc; ac.w += q.w * sc;
        }
      }
      m[hh] = mm;
      l[hh] = ls;
      acc[hh] = ac;
    }
  }

#pragma unroll
  for (int hh = 0; hh < LBM_HEADS; ++hh) {
    const float inv = l[hh] > 0.f ? 1.0f / l[hh] : 0.f;
    const float4 o = make_float4(acc[hh].x * inv, acc[hh].y * inv, acc[hh].z * inv, acc[hh].w * inv);
    const long long oo = ((long long)row * LBM_HEADS + hh) * LBM_C + lane * 4;
    if (a.out) st4(a.out + oo, o);
    if (a.out_s) st4_split(a.out_s, a.out_plane, oo, o);
  }
  if (a.valid && lane == 0) a.valid[row] = l[0] > 0.f ? 1 : 0;
}

static int g_attn_mode = 0;   // tuning knob (debug key 3): 0 auto, 1 = one wave per point, 2 = two, 4 = four waves per point; + 16 = every lane evaluates every exponential
void attn_force_mode(int m) { g_attn_mode = m; }

int attn_readout(const AttnArgs& a, hipStream_t stream) {
  if (a.R == 0) return 0;
  LBM_REQUIRE(a.U && a.seq && (a.out || a.out_s), "attn_readout: null argument");
  LBM_REQUIRE(a.S >= 1 && a.S <= 4096, "attn_readout: sequence length %d out of range", a.S);
  LBM_REQUIRE(a.ring_head >= 0 && a.ring_head < a.S, "attn_readout: ring head %d outside [0,%d)", a.ring_head, a.S);
  LBM_REQUIRE(a.slot_stride % 4 == 0 && a.point_stride % 4 == 0 && a.ldu % 4 == 0 && a.ldu >= 4 * LBM_C,
              "attn_readout: strides must be multiples of 4 floats");
  // algorithmic traffic: the S sequence rows once, 4 per-head queries in, 4 per-head read-outs out, the masks
  ProfScope prof("attn_readout", stream, (double)a.R * a.S * 4.0 * LBM_C * 4,
                 (double)a.R * (LBM_C * ((a.seq_f16 ? 2.0 : 4.0) * a.S + 4.0 * 8) + (a.mask_a ? a.S : 0) + (a.mask_b ? a.S : 0)));
  const int mode = g_attn_mode & 15;
  const bool wide = mode == 4 || (mode == 0 && a.R < 4096 && a.S >= 8);
  // two waves per point (forced with debug key 3 = 2 only): measured no faster than one at 10 000 points x 48 or 96 slots
  // (124.1 against 121.7 us, 224 against 222), so the automatic choice never takes it
  const bool pair = !wide && mode == 2 && !(g_attn_mode & 16);
  const dim3 grid(wide ? a.R : pair ? (a.R + 1) / 2 : (a.R + 3) / 4);
#define LBM_ATTN_LAUNCH(W, T, X) hipLaunchKernelGGL((attn_readout_kernel<W, T, X>), grid, dim3(256), 0, stream, a)
  if (a.seq_f16) {             // f16 memories (f16-storage mode): the same kernel reading 512-byte slot rows
    LBM_REQUIRE(a.temb != nullptr && !pair, "attn_readout: f16 sequences are the masked memory read-outs");
    if (wide) hipLaunchKernelGGL((attn_readout_kernel<4, true, true, _Float16>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_readout_kernel<1, true, true, _Float16>), grid, dim3(256), 0, stream, a);
  } else if (g_attn_mode & 16) {      // the first form of the soft-max step (bit-identity test)
    if (wide) { if (a.temb) LBM_ATTN_LAUNCH(4, true, false); else LBM_ATTN_LAUNCH(4, false, false); }
    else { if (a.temb) LBM_ATTN_LAUNCH(1, true, false); else LBM_ATTN_LAUNCH(1, false, false); }
  } else {
    if (wide) { if (a.temb) LBM_ATTN_LAUNCH(4, true, true); else LBM_ATTN_LAUNCH(4, false, true); }
    else if (pair) { if (a.temb) LBM_ATTN_LAUNCH(2, true, true); else LBM_ATTN_LAUNCH(2, false, true); }
    else { if (a.temb) LBM_ATTN_LAUNCH(1, true, true); else LBM_ATTN_LAUNCH(1, false, true); }
  }
#undef LBM_ATTN_LAUNCH
  return check_launch("attn_readout");
}

}  // namespace lbm
