// conv_patch: 3 x 3, stride 1, pad 1 convolution with Cin = 128 / 256 (ResNet layer2 / layer3, backbone.py:20-26,48-65) for MANY pixels,
// NHWC split planes.
//
// The tiled implicit GEMM (gemm16.hip, CONV) is bound by what it pulls through LDS-DMA: per 128 x 128 x 32 chunk it stages 32 KB for
// 96 MFMA blocks, every input pixel nine times (once per tap), and with two stages per workgroup only 64 KB per CU are ever in flight
// (scripts/conv_abl64.py: 215 us per layer-2 convolution of 64 frames, 130 us with the DMA switched off).  Here a workgroup of 16 waves
// owns a tile of 256 output pixels -- whole image rows, 4 rows of 64 or 8 rows of 32 -- x 128 output channels and keeps the INPUT PATCH
// of one 32-channel chunk ((rows + 2) x (W + 2) pixels, both planes: 50 / 44 KB) in LDS for all nine taps; the taps read their operand
// fragments straight out of it at shifted pixel offsets.  Only the weights still stream chunk by chunk (16 KB per tap and channel
// chunk, ring of 3 stages, ONE barrier per chunk): 21.6 KB of DMA per 192 MFMA blocks instead of 32 KB per 96.  The next channel
// chunk's patch is fetched into a second buffer while the current one is multiplied.
//
// LDS images: a pixel (patch) or weight row = 32 channels = 64 B = four 16-byte slots, slot s of row p stored at s ^ ((p >> 2) & 3)
// (applied on the source side of the DMA; the LDS side is lane-linear): the 16 lanes of a ds_read_b128 group read 16 consecutive rows
// -- from ANY starting row, so the tap shifts stay conflict-free.
//
// K is walked channel chunk by channel chunk, taps inside (the tiled kernel walks tap by tap): the same products summed in another
// order -- results agree to fp32 rounding, not bit for bit.
#include "gemm.hpp"

namespace lbm {

int g_conv_patch_dbg = 0;   // debug key 26 (bits 0-7: timing-only ablation, tuning build)
int g_conv_patch_tile = 0;  // debug key 26 >> 8: 0 = choose, 192 / 256 = pixels per tile

struct ConvPatchDev {
  const _Float16* X;      // [B, H, WO, Cin] hi plane; lo plane x_plane elements further
  long long x_plane;
  const _Float16* Wh;     // [Cout, ldw], columns (ky, kx, ci)
  long long w_plane;      // distance to the lo plane (elements)
  int ldw;
  const float* bias;
  const float* res;
  float* Y;
  _Float16* Ys;
  long long ys_plane;
  const _Float16* zeros;
  int H, Cin, Cout, relu, cin_log2;
  int items;   // (tiles of pixels) x (blocks of 128 output channels), output-channel block fastest
  int dbg;   // tuning (timing only, debug key 26): 1 no patch DMA after the first, 2 no weight DMA after the prologue, 4 no MFMAs, 8 no epilogue, 16 no barrier in the main loop, 32 no DMA waits
};

typedef __attribute__((address_space(3))) void* lds_ptr_cp;

__device__ __forceinline__ void cp_wait(int n) {      // n wave-uniform, 0..5
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
  }
}

// WO: image width (64 / 32); TR: image rows per tile; the tile's BM = TR * WO pixels are dealt to BM / 64 rows of waves (64 pixels
// each, two 32-pixel MFMA blocks), four waves (32 output channels each) per row: 16 waves at BM = 256, 12 at BM = 192.
//
// PERSISTENT: a workgroup walks items (tile of pixels x block of 128 output channels) w, w + G, ... so that nothing of a tile's start
// and end is paid per tile: the first patch of the next item is fetched during the last channel chunk of the current one (its buffer is
// free by then), its first two weight chunks right behind the main loop.  ROLES: every wave multiplies, but only the LOWER half of the
// waves issue DMA and only the UPPER half run the epilogue's residual loads and output stores -- vector memory operations of a wave
// retire in issue order, so a wave that both stored and fetched would have to see its stores complete before its next fetch could be
// waited for; this way the stores of a tile drain behind the next tile's multiplication.  The epilogue passes the tile through LDS in
// slices of 64 rows (one row of waves), which fit the patch buffer that is idle at that moment.
template <int WO, int TR>
__global__ __launch_bounds__(TR * WO * 4) void conv_patch_kernel(ConvPatchDev a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = TR * WO;
  constexpr int NW = BM / 16;                  // waves
  constexpr int NWM = NW / 4;                  // rows of waves
  constexpr int ND = NW / 2;                   // DMA waves (0 .. ND - 1); the others store
  constexpr int PW = WO + 2, PROWS = TR + 2;
  constexpr int NPIX = PROWS * PW;             // patch pixels per plane
  constexpr int NPP = (NPIX + 15) / 16;        // 1-KiB DMA pieces per plane (16 pixels each)
  constexpr int PLANE_P = NPP * 1024;
  constexpr int PATCHB = 2 * PLANE_P;          // one patch buffer: [hi plane][lo plane]
  constexpr int NPC = 2 * NPP;                 // pieces per patch
  constexpr int NPW = (NPC + ND - 1) / ND;     // most patch pieces per DMA wave
  constexpr int NBW = (16 + ND - 1) / ND;      // most weight pieces per DMA wave and chunk
  constexpr int BSTAGE = 16384;                // weights of one chunk: [hi: 128 rows x 64 B][lo]
  constexpr int NB = 3;
  // LDS: [patch 0][weights 0][weights 1][patch 1][weights 2]; the epilogue's 64-row slice lives in patch 1
  constexpr int OFF_P0 = 0, OFF_B0 = PATCHB, OFF_P1 = PATCHB + 2 * BSTAGE, OFF_B2 = OFF_P1 + PATCHB;
  constexpr int LDS_BYTES = OFF_B2 + BSTAGE;
  constexpr int TP = 132;                      // pitch (floats) of the staged slice
  static_assert(BM % 64 == 0 && LDS_BYTES <= 160 * 1024 && 64 * TP * 4 <= PATCHB, "tile / LDS budget");
  static_assert(NPW <= 9 && NBW <= 3, "one patch piece per chunk; wait table");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const bool dma_wave = wave < ND;
  const int hw = a.H * WO;
  const int NCC = a.Cin >> 5, NQ = NCC * 9;
  const int ntn = a.Cout >> 7;
  const int G = gridDim.x;

  auto sync_lds = [&]() __attribute__((always_inline)) {     // LDS-only workgroup barrier: vector memory stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto stage_w = [&](int st) __attribute__((always_inline)) { return lds + (st == 2 ? OFF_B2 : OFF_B0 + st * BSTAGE); };
  auto patch_buf = [&](int cc) __attribute__((always_inline)) { return lds + ((cc & 1) ? OFF_P1 : OFF_P0); };

  // ---- patch DMA (DMA waves): wave w fetches pieces w, w + ND, ..., one per chunk; the source of a piece is worked out when it is
  // issued (element offset into the plane, or the zero page for pixels outside the image) ----
  const int npw = dma_wave ? (NPC - wave + ND - 1) / ND : 0;  // this wave's pieces (wave-uniform)
  auto issue_patch_piece = [&](int b, int y0, int cc, int i) __attribute__((always_inline)) {      // i wave-uniform, < npw
    const int pc = wave + ND * i;
    const int pp = pc < NPP ? pc : pc - NPP;
    const int pixel = pp * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((pixel >> 2) & 3);
    const int row = pixel / PW, px = pixel - row * PW;
    const int y = y0 - 1 + row, x = px - 1;
    const bool ok = pixel < NPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)WO;
    const _Float16* plane = a.X + (pc < NPP ? 0 : a.x_plane);
    const _Float16* g = ok ? plane + (((long long)b * a.H + y) * WO + x) * a.Cin + chunk * 8 + cc * 32 : a.zeros;
    __builtin_amdgcn_global_load_lds(g, (lds_ptr_cp)(patch_buf(cc) + pc * 1024), 16, 0, 0);
  };
  // ---- weight DMA (DMA waves): pieces w, w + ND, ... of the chunk's 16 (16 rows of one plane each) ----
  const int nbw = dma_wave ? (16 - wave + ND - 1) / ND : 0;
  long long woff[NBW];                         // element offset of this lane's 16 bytes inside the weights, without n0 and k
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int pc = (wave + ND * i) & 15;
    const int rowb = (pc & 7) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((rowb >> 2) & 3);
    woff[i] = (pc >> 3) * a.w_plane + (long long)rowb * a.ldw + chunk * 8;
  }
  auto issue_w = [&](int n0, int st, int koff) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBW; ++i)
      if (i < nbw)
        __builtin_amdgcn_global_load_lds(a.Wh + woff[i] + (long long)n0 * a.ldw + koff, (lds_ptr_cp)(stage_w(st) + (wave + ND * i) * 1024), 16, 0, 0);
  };

  // ---- fragment offsets ----
  int p0[2];                                   // base patch pixel of block i for this lane (tap (0, 0))
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t0 = 64 * wm + 32 * i;           // first tile pixel of the block, raster order
    p0[i] = (t0 / WO) * PW + (t0 % WO) + r;
  }
  const int rowB = wn * 32 + r;
  const int offB = rowB * 64 + ((h ^ ((rowB >> 2) & 3)) << 4);

  auto item_of = [&](int it, int& b, int& y0, int& n0, long long& m0) __attribute__((always_inline)) {
    const int tm = it / ntn;
    n0 = (it - tm * ntn) * 128;
    m0 = (long long)tm * BM;
    b = (int)(m0 / hw);
    y0 = (int)(m0 - (long long)b * hw) / WO;   // first image row of the tile (a multiple of TR)
  };

  int it = blockIdx.x;
  if (it >= a.items) return;
  int b, y0, n0;
  long long m0;
  item_of(it, b, y0, n0, m0);
  if (dma_wave) {
    for (int i = 0; i < npw; ++i) issue_patch_piece(b, y0, 0, i);
    issue_w(n0, 0, 0);
    issue_w(n0, 1, a.Cin);                     // (tap 1, cc 0)
  }

#pragma unroll 1
  for (;;) {
    const int itn = it + G;                    // the next item of this workgroup
    const bool has_next = itn < a.items;

    f32x16 acc[2], acx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = acx[i][e] = 0.f;

    int tap = 0, cc = 0;
    int tap2 = 2, cc2 = 0;                     // (tap, cc) of chunk q + 2
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      // DMA waves: everything up to the weights of chunk q has landed.  Issued behind them and possibly still in flight: the weights
      // of chunk q + 1 and the pieces of the next patch issued in iterations q - 2 and q - 1 (one per iteration, taps 0 .. npw - 1).
      const bool nextp = (cc + 1 < NCC || has_next) && !(LBM_ABL(a.dbg & 1));
      if (dma_wave) {
        int allow = (q + 1 < NQ && !(LBM_ABL(a.dbg & 2))) ? nbw : 0;
        if (nextp) allow += (tap >= 1 && tap - 1 < npw ? 1 : 0) + (tap >= 2 && tap - 2 < npw ? 1 : 0);
        if (!(LBM_ABL(a.dbg & 32))) cp_wait(allow);
      }
      if (!(LBM_ABL(a.dbg & 16))) sync_lds();
      // every wave is done with chunk q - 1 since this chunk's barrier: its weight stage and the other patch buffer are free.  The DMA
      // goes out FIRST: issued behind the MFMAs it lands too late (layer-2 shape 174 -> 188 us)
      if (dma_wave) {
        if (q + 2 < NQ && !(LBM_ABL(a.dbg & 2))) issue_w(n0, (q + 2) % NB, tap2 * a.Cin + cc2 * 32);
        if (nextp && tap < npw) {
          if (cc + 1 < NCC) issue_patch_piece(b, y0, cc + 1, tap);
          else {                                // (NCC is even: chunk 0 of the next item goes to buffer 0, idle now)
            int bn, y0n, n0n;
            long long m0n;
            item_of(itn, bn, y0n, n0n, m0n);       // (worked out where it is used: kept across the loop it costs scalar registers)
            issue_patch_piece(bn, y0n, 0, tap);
          }
        }
      }

      const unsigned char* P = patch_buf(cc);
      const unsigned char* Bs = stage_w(q % NB);
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int tshift = ky * PW + kx;
      // fragments in the order the MFMAs want them (the waits count in issue order): k-step 0 first, weights before pixels
      half8 fah[2][2], fal[2][2], fbh[2], fbl[2];
      int offA[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = p0[i] + tshift;
        offA[i] = p * 64 + ((h ^ ((p >> 2) & 3)) << 4);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        fbh[s] = *reinterpret_cast<const half8*>(Bs + (offB ^ (s << 5)));
        fah[s][0] = *reinterpret_cast<const half8*>(P + (offA[0] ^ (s << 5)));
        fbl[s] = *reinterpret_cast<const half8*>(Bs + 8192 + (offB ^ (s << 5)));
        fal[s][0] = *reinterpret_cast<const half8*>(P + PLANE_P + (offA[0] ^ (s << 5)));
        fah[s][1] = *reinterpret_cast<const half8*>(P + (offA[1] ^ (s << 5)));
        fal[s][1] = *reinterpret_cast<const half8*>(P + PLANE_P + (offA[1] ^ (s << 5)));
      }
      if (!(LBM_ABL(a.dbg & 4)))
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbh[s], acc[i], 0, 0, 0);
          acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[s][i], fbl[s], acx[i], 0, 0, 0);
          acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[s][i], fbh[s], acx[i], 0, 0, 0);
        }
      if (++tap == 9) { tap = 0; ++cc; }
      if (++tap2 == 9) { tap2 = 0; ++cc2; }
    }

    // ---- epilogue: act(acc + bias) through LDS in slices of 64 rows (patch buffer 1: idle since the last chunk), whole rows out,
    // 16 bytes per lane, by the storing waves; meanwhile the DMA waves have the next item's first weight chunks on their way ----
    sync_lds();                                // every wave is done with the last chunk: patch 1 and all weight stages are free
    if (dma_wave && has_next && !(LBM_ABL(a.dbg & 2))) {
      const int n0n = (itn % ntn) * 128;
      issue_w(n0n, 0, 0);
      issue_w(n0n, 1, a.Cin);
    }
    if (!(LBM_ABL(a.dbg & 8))) {
      float* T = reinterpret_cast<float*>(lds + OFF_P1);
      const float bv = a.bias ? a.bias[n0 + wn * 32 + r] : 0.f;
#pragma unroll 1
      for (int j = 0; j < NWM; ++j) {
        if (j > 0) sync_lds();                 // the slice before has been read
        if (wm == j) {
          const int cl = wn * 32 + r;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int rl = i * 32 + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) T[(rl + (e & 3) + 8 * (e >> 2)) * TP + cl] = (acc[i][e] + acx[i][e] * (1.0f / 2048.0f)) + bv;
          }
        }
        sync_lds();
        if (!dma_wave) {
          for (int c = tid - ND * 64; c < 64 * 32; c += ND * 64) {
            const int rl = c >> 5, c4 = c & 31;
            const long long o = (m0 + j * 64 + rl) * a.Cout + n0 + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(T + rl * TP + c4 * 4);
            if (a.res) {
              const float4 p = ld4(a.res + o);
              v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (a.Y) st4(a.Y + o, v);
            if (a.Ys) st4_split(a.Ys, a.ys_plane, o, v);
          }
        }
      }
    }
    if (!has_next) break;
    it = itn;
    item_of(it, b, y0, n0, m0);
    // (the next main loop starts with a barrier: the last slice has been read before chunk 0 lets patch 1 / weight stage 2 be refilled)
  }
#endif
}

// rows of the image per tile: 256 pixels where the rows divide evenly, else 192 (twelve waves)
static int conv_patch_rows(const ConvSSArgs& c, int force) {
  const int r256 = 256 / c.Wd, r192 = 192 / c.Wd;
  if (force == 192) return c.H % r192 == 0 ? r192 : 0;
  if (force == 256) return c.H % r256 == 0 ? r256 : 0;
  // fewest idle CUs in the last round of workgroups (one workgroup per CU)
  int best = 0;
  double best_eff = 0.0;
  for (int rows : {r256, r192}) {
    if (c.H % rows != 0) continue;
    const long long wgs = (long long)c.B * (c.H / rows) * (c.Cout / 128);
    const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    if (best == 0 || eff > best_eff + 0.02) { best_eff = eff; best = rows; }
  }
  return best;
}

bool conv_patch_ok(const ConvSSArgs& c) {
  if (!(c.planes == 2 && c.kh == 3 && c.kw == 3 && c.stride == 1 && c.pad == 1 && c.Wl != nullptr)) return false;
  if (!((c.Cin == 128 || c.Cin == 256) && c.Cout % 128 == 0 && (c.Wd == 64 || c.Wd == 32))) return false;
  if (c.H % (256 / c.Wd) != 0 && c.H % (192 / c.Wd) != 0) return false;
  const long long M = (long long)c.B * c.H * c.Wd;
  if (M * c.Cin >= 0x7fffffffLL) return false;                  // patch offsets are 32-bit element offsets into a plane
  return c.ldw >= 9 * c.Cin && c.ldw % 8 == 0 && c.ys_plane % 4 == 0 && c.x_plane % 8 == 0 && ((uintptr_t)c.Y & 15) == 0 &&
         ((uintptr_t)c.Ys & 7) == 0 && ((uintptr_t)c.res & 15) == 0 && ((uintptr_t)c.X & 15) == 0 && ((uintptr_t)c.Wh & 15) == 0 &&
         ((uintptr_t)c.Wl & 15) == 0 && ((uintptr_t)c.zeros & 15) == 0 && (((const char*)c.Wl - (const char*)c.Wh) % 16) == 0;
}

int conv_patch(const ConvSSArgs& c, hipStream_t stream) {
  LBM_REQUIRE(conv_patch_ok(c) && c.X && c.Wh && (c.Y || c.Ys) && c.zeros, "conv_patch: shape / alignment not supported");
  const long long M = (long long)c.B * c.H * c.Wd;
  ConvPatchDev d;
  d.X = (const _Float16*)c.X; d.x_plane = c.x_plane; d.Wh = (const _Float16*)c.Wh;
  d.w_plane = ((const char*)c.Wl - (const char*)c.Wh) / 2; d.ldw = c.ldw;
  d.bias = c.bias; d.res = c.res; d.Y = c.Y; d.Ys = (_Float16*)c.Ys; d.ys_plane = c.ys_plane; d.zeros = (const _Float16*)c.zeros;
  d.dbg = g_conv_patch_dbg & 255;
  d.H = c.H; d.Cin = c.Cin; d.Cout = c.Cout; d.relu = c.relu; d.cin_log2 = c.Cin == 128 ? 7 : 8;
  const int rows = conv_patch_rows(c, g_conv_patch_tile);
  LBM_REQUIRE(rows > 0, "conv_patch: %d image rows do not divide into tiles", c.H);
  const int BM = rows * c.Wd;
  d.items = (int)(M / BM) * (c.Cout / 128);
  // one workgroup per CU walks its items; the grid divides the items evenly where it can (768 items: 256 x 3, 512: 256 x 2)
  int wgs = d.items < 256 ? d.items : 256;
  const int per = (d.items + wgs - 1) / wgs;
  wgs = (d.items + per - 1) / per;
  const dim3 grid(wgs);
  const double mnk = (double)M * c.Cout * 9.0 * c.Cin;
  ProfScope prof("conv_patch", stream, 2.0 * mnk,
                 4.0 * ((double)M * c.Cin + (double)M * c.Cout * ((c.Y ? 1 : 0) + (c.Ys ? 1 : 0) + (c.res ? 1 : 0)) + 9.0 * c.Cin * c.Cout));
  if (c.Wd == 64 && BM == 256) hipLaunchKernelGGL((conv_patch_kernel<64, 4>), grid, dim3(1024), 0, stream, d);
  else if (c.Wd == 64) hipLaunchKernelGGL((conv_patch_kernel<64, 3>), grid, dim3(768), 0, stream, d);
  else if (BM == 256) hipLaunchKernelGGL((conv_patch_kernel<32, 8>), grid, dim3(1024), 0, stream, d);
  else hipLaunchKernelGGL((conv_patch_kernel<32, 6>), grid, dim3(768), 0, stream, d);
  return check_launch("conv_patch");
}

}  // namespace lbm
