// Back-to-back cost of the library's own small kernels from C++ (no Python in the loop): LayerNorm of 1024 rows, a
// 1024x1024x256 split GEMM, and the GEMM -> GEMM(K-split) -> LayerNorm MLP chain of the tracker at 1024 points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <functional>
#include "../../include/lbm_hip.h"

static float time_us(const std::function<void()>& f, int n, hipStream_t s) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(a, s);
  for (int i = 0; i < n; ++i) f();
  (void)hipEventRecord(b, s);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / n;
}

int main() {
  const int R = 1024;
  hipStream_t s;
  (void)hipStreamCreate(&s);
  char* buf;
  (void)hipMalloc(&buf, 256 << 20);
  (void)hipMemset(buf, 0, 256 << 20);
  float* X = (float*)buf;                       // [R,1024] fp32
  float* Y = (float*)(buf + (16 << 20));
  float* g = (float*)(buf + (32 << 20));
  void* As = buf + (48 << 20);                  // split planes
  void* Ws = buf + (64 << 20);
  void* Hs = buf + (80 << 20);
  float* slab = (float*)(buf + (96 << 20));
  const int N = 1000;
  float t = time_us([&] { lbm_hip_layernorm(X, g, g, Y, R, nullptr, nullptr, nullptr, s); }, N, s);
  printf("layernorm 1024 rows           : %6.2f us/launch\n", t);
  lbm_hip_linear_ss_desc d;
  memset(&d, 0, sizeof d);
  d.A = As; d.a_plane = (int64_t)R * 256; d.B = Ws; d.b_plane = 1024 * 256; d.M = R; d.N = 1024; d.K = 256;
  d.lda = 256; d.ldb = 256; d.Ys = Hs; d.ys_plane = (int64_t)R * 1024; d.ldys = 1024; d.ldy = 1024; d.ldres = 256;
  d.res_div = 1; d.res_mod = 0x7fffffff; d.res_scale = 1.f; d.batch = 1; d.act = 1;
  t = time_us([&] { lbm_hip_linear_ss(&d, s); }, N, s);
  printf("gemm 1024x1024x256 (split out): %6.2f us/launch\n", t);
  lbm_hip_linear_ss_desc e;
  memset(&e, 0, sizeof e);
  e.A = Hs; e.a_plane = (int64_t)R * 1024; e.B = Ws; e.b_plane = 1024 * 256; e.M = R; e.N = 256; e.K = 256;
  e.lda = 1024; e.ldb = 1024; e.Y = slab; e.ldy = 256; e.ldys = 256; e.ldres = 256;
  e.res_div = 1; e.res_mod = 0x7fffffff; e.res_scale = 1.f; e.batch = 4; e.sA = 256; e.sB = 256; e.sY = (int64_t)R * 256;
  t = time_us([&] { lbm_hip_linear_ss(&e, s); }, N, s);
  printf("gemm 1024x256x1024 as 4 K-slabs: %6.2f us/launch\n", t);
  t = time_us([&] { lbm_hip_linear_ss(&d, s); lbm_hip_linear_ss(&e, s); lbm_hip_layernorm(slab, g, g, Y, R, nullptr, nullptr, nullptr, s); }, N, s);
  printf("chain gemm -> gemm -> layernorm: %6.2f us per chain of 3\n", t);
  const int flags[] = {0, 16, 8, 7, 6, 3, 5, 1, 2};
  const char* names[] = {"full", "full, direct epilogue", "return at once", "no DMA, no MFMA, no stores", "no MFMA, no stores", "no DMA, no MFMA", "no DMA, no stores", "no DMA", "no MFMA"};
  for (int i = 0; i < 9; ++i) {
    lbm_hip_debug_set(2, flags[i]);
    const float t1 = time_us([&] { lbm_hip_linear_ss(&d, s); }, N, s);
    const float t2 = time_us([&] { lbm_hip_linear_ss(&e, s); }, N, s);
    printf("%-28s: 1024x1024x256 %6.2f   1024x256x(4x256) %6.2f\n", names[i], t1, t2);
  }
  lbm_hip_debug_set(2, 0);
  d.act = 0;
  printf("1024x1024x256 split out, no GELU      : %6.2f\n", time_us([&] { lbm_hip_linear_ss(&d, s); }, N, s));
  d.Y = X; d.Ys = nullptr;
  printf("1024x1024x256 fp32 out, no GELU       : %6.2f\n", time_us([&] { lbm_hip_linear_ss(&d, s); }, N, s));
  d.Ys = Hs;
  printf("1024x1024x256 fp32 + split, no GELU   : %6.2f\n", time_us([&] { lbm_hip_linear_ss(&d, s); }, N, s));
  d.M = 16384; d.Y = nullptr; d.act = 1;   // buffers are large enough: 16384 x 1024 split = 64 MB
  return 0;
}
