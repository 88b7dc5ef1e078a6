// What does an (almost) empty kernel cost back-to-back on one stream, by LDS size, grid, block and kernarg size?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { long long v[32]; };   // 256-byte kernarg

__global__ void k_small(float* p, int n) {
  extern __shared__ unsigned char lds[];
  if (n < 0) p[threadIdx.x] = lds[threadIdx.x];
}
__global__ void k_big(float* p, Big b, int n) {
  extern __shared__ unsigned char lds[];
  if (n < 0) p[threadIdx.x] = lds[threadIdx.x] + (float)b.v[threadIdx.x & 31];
}
// many VGPRs: forces a large register allocation per wave
__global__ __launch_bounds__(256, 1) void k_regs(float* p, int n) {
  extern __shared__ unsigned char lds[];
  float acc[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) acc[i] = p[i] * (float)n;
  if (n < 0) {
    float s = 0;
#pragma unroll
    for (int i = 0; i < 200; ++i) s += acc[i];
    p[threadIdx.x] = s + lds[threadIdx.x];
  }
}
// touches memory: a dependent read-modify-write of 1 MB
__global__ void k_touch(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.0f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
static float time_us(F launch, int n, hipStream_t s) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1e3f / n;
}

int main() {
  float* p;
  CK(hipMalloc(&p, 64 << 20));
  CK(hipMemset(p, 0, 64 << 20));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  CK(hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int N = 2000;
  Big b{};
  printf("back-to-back launches on one stream, us per launch\n");
  for (int lds : {0, 32 * 1024, 64 * 1024, 128 * 1024, 160 * 1024})
    for (int grid : {64, 256, 1024, 4096})
      for (int block : {64, 256, 512}) {
        float t = time_us([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(block), lds, s, p, 1); }, N, s);
        printf("k_small lds %6d grid %5d block %4d : %6.2f\n", lds, grid, block, t);
      }
  for (int lds : {0, 128 * 1024}) {
    float t = time_us([&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), lds, s, p, b, 1); }, N, s);
    printf("k_big (256 B kernarg) lds %6d grid 256 block 256 : %6.2f\n", lds, t);
    t = time_us([&] { hipLaunchKernelGGL(k_regs, dim3(256), dim3(256), lds, s, p, 0); }, N, s);
    printf("k_regs (200 live floats) lds %6d grid 256 block 256 : %6.2f\n", lds, t);
  }
  for (int n : {1 << 12, 1 << 18, 1 << 20, 1 << 22}) {
    float t = time_us([&] { hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, s, p, n); }, N, s);
    printf("k_touch n %8d (dependent RMW of %d KB): %6.2f\n", n, n * 4 / 1024, t);
  }
  return 0;
}
