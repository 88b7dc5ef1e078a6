// How fast do the f16 matrix cores go in the access pattern of the split-precision products?  Bare loops, no global memory.
//   mode 0: 3 MFMAs per k-step into TWO accumulators (acc, acx, acx) -- the dependency pattern of gemm16.hip / gemm_rs.hip
//   mode 1: the same 3 MFMAs into three different accumulators (no back-to-back dependency)
//   mode 2: mode 0 with both A fragments of every k-step read from LDS (ds_read_b128), B in registers -- gemm_rs.hip's inner loop
//   mode 3: mode 2 plus B fragments from LDS as well -- gemm16.hip's inner loop (wave tile 64 x 32: 6 reads per 6 MFMAs)
//   mode 4: mode 2 with the reads software-pipelined by hand: groups of two k-steps, the next group's four reads issued before this
//           group's six MFMAs and pinned there with sched_barrier (left alone, the compiler re-uses one register set and waits for
//           every read right before its MFMA)
// usage: mfma_peak [waves per workgroup = 8] [workgroups = 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  half8 b[16], bl[16];
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 8; ++e) {
      b[i][e] = (_Float16)(0.001f * (lane + i + e));
      bl[i][e] = (_Float16)(0.002f * (lane + i - e));
    }
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  const int sw = (r >> 2) & 3;
  const int o0 = r * 64 + ((h ^ sw) << 4), o1 = r * 64 + (((2 + h) ^ sw) << 4);
  f32x16 acc = {0}, acx = {0}, acy = {0};
  half8 a0 = b[3], a1 = bl[5];
  if (MODE == 4) {
    half8 fa[2][4];
    auto ldg = [&](int bsel, int g) __attribute__((always_inline)) {
      const unsigned char* p = lds + g * 4096;
      fa[bsel][0] = *reinterpret_cast<const half8*>(p + o0);
      fa[bsel][1] = *reinterpret_cast<const half8*>(p + 2048 + o0);
      fa[bsel][2] = *reinterpret_cast<const half8*>(p + o1);
      fa[bsel][3] = *reinterpret_cast<const half8*>(p + 2048 + o1);
    };
    ldg(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        ldg((g + 1) & 1, (g + 1) & 7);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ks = 2 * g + q;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], b[ks], acc, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q], bl[ks], acx, 0, 0, 0);
          acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][2 * q + 1], b[ks], acx, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (MODE >= 2) {
        const unsigned char* p = lds + (ks >> 1) * 4096 + ((ks & 1) ? o1 : o0);
        a0 = *reinterpret_cast<const half8*>(p);
        a1 = *reinterpret_cast<const half8*>(p + 2048);
      }
      half8 bh = b[ks], bb = bl[ks];
      if (MODE == 3) {
        const unsigned char* p = lds + 32768 + (ks >> 1) * 4096 + ((ks & 1) ? o1 : o0);
        bh = *reinterpret_cast<const half8*>(p);
        bb = *reinterpret_cast<const half8*>(p + 2048);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bh, acc, 0, 0, 0);
      if (MODE == 1) {
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bb, acx, 0, 0, 0);
        acy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh, acy, 0, 0, 0);
      } else {
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bb, acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh, acx, 0, 0, 0);
      }
    }
    if (MODE < 2) { a0[0] += (_Float16)1e-3f; }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e] + acx[e] + acy[e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
static void run(int waves, int wgs, float* out) {
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(waves * 64), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(waves * 64), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)wgs * waves * iters * 48 * 32.0 * 32 * 16 * 2;
  printf("mode %d: %d waves/WG x %d WGs: %.3f ms, %.0f TFLOP/s MFMA (%.0f algorithmic split TF)\n", MODE, waves, wgs, ms, mf / ms / 1e9, mf / ms / 1e9 / 3);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 8, wgs = argc > 2 ? atoi(argv[2]) : 1024;
  float* out;
  hipMalloc(&out, 4096);
  run<0>(waves, wgs, out);
  run<1>(waves, wgs, out);
  run<2>(waves, wgs, out);
  run<3>(waves, wgs, out);
  run<4>(waves, wgs, out);
  run<0>(4, wgs, out);
  run<4>(4, wgs, out);
  run<2>(4, wgs, out);
  run<3>(4, wgs, out);
  return 0;
}
