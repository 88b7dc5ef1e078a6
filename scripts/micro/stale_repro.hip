// Producer / consumer on one HIP stream, with and without a busy second stream: does the consumer's first load ever see
// the buffer's previous contents?  ("stale reads at kernel start".)  No torch, no library code.
//   build: hipcc -O3 --offload-arch=gfx950 stale_repro.hip -o stale_repro
//   run:   ./stale_repro [iterations = 20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int kRows = 1024, kCols = 256;                 // the shape of one activation buffer of the per-point chain

// wave per row, lane = 4 floats: the way the LayerNorm kernel writes its output
__global__ __launch_bounds__(256) void produce(float* x, float v) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  *reinterpret_cast<float4*>(x + (long long)row * kCols + lane * 4) = make_float4(v, v, v, v);
}

// 32 rows per workgroup, every thread's first instructions read 4 x 16 bytes of the tile (as the fused feed-forward does)
__global__ __launch_bounds__(256) void consume(const float* x, float want, unsigned* bad, float* seen) {
  const int m0 = blockIdx.x * 32;
  float4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = it * 256 + threadIdx.x;          // 32 rows x 64 pieces of 16 bytes = 2048 pieces; 4 passes cover half
    const int row = piece >> 5, pc = piece & 31;
    v[it] = *reinterpret_cast<const float4*>(x + (long long)(m0 + row) * kCols + pc * 8);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it)
    if (v[it].x != want || v[it].y != want || v[it].z != want || v[it].w != want) {
      atomicAdd(bad, 1u);
      seen[0] = v[it].x;
      seen[1] = want;
    }
}

// the second queue's work: a streaming read-modify-write over a large buffer (bounded, no spinning)
__global__ __launch_bounds__(256) void busy(float* y, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 q = reinterpret_cast<float4*>(y)[i];
    q.x += 1.f; q.y += 1.f; q.z += 1.f; q.w += 1.f;
    reinterpret_cast<float4*>(y)[i] = q;
  }
}

static int run(int iters, bool second_queue, int filler) {
  float *x, *y, *seen;
  unsigned* bad;
  const long long n4 = 8ll << 20;                         // 128 MB
  CK(hipMalloc(&x, sizeof(float) * kRows * kCols));
  CK(hipMalloc(&y, sizeof(float) * 4 * n4));
  CK(hipMalloc(&bad, sizeof(unsigned)));
  CK(hipMalloc(&seen, 2 * sizeof(float)));
  CK(hipMemset(x, 0, sizeof(float) * kRows * kCols));
  CK(hipMemset(y, 0, sizeof(float) * 4 * n4));
  CK(hipMemset(bad, 0, sizeof(unsigned)));
  CK(hipMemset(seen, 0, 2 * sizeof(float)));
  hipStream_t a, b;
  CK(hipStreamCreate(&a));
  CK(hipStreamCreate(&b));
  CK(hipDeviceSynchronize());
  for (int i = 1; i <= iters; ++i) {
    hipLaunchKernelGGL(produce, dim3(kRows / 4), dim3(256), 0, a, x, (float)i);
    for (int f = 0; f < filler; ++f) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, a, y + 4 * n4 - 65536, 4096ll);   // unrelated small kernels
    hipLaunchKernelGGL(consume, dim3(kRows / 32), dim3(256), 0, a, x, (float)i, bad, seen);
    if (second_queue && (i & 3) == 0) hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, b, y, n4 / 8);
    if ((i & 1023) == 0) CK(hipStreamSynchronize(a));     // keep the queues short
  }
  CK(hipDeviceSynchronize());
  unsigned h = 0;
  float hs[2];
  CK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hs, seen, sizeof(hs), hipMemcpyDeviceToHost));
  std::printf("second queue %d, filler %d: %d producer/consumer pairs, %u stale 16-byte pieces", (int)second_queue, filler, iters, h);
  if (h) std::printf(" (e.g. saw %.0f, wanted %.0f)", hs[0], hs[1]);
  std::printf("\n");
  CK(hipStreamDestroy(a));
  CK(hipStreamDestroy(b));
  CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(bad)); CK(hipFree(seen));
  return h ? 1 : 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
  int rc = 0;
  rc |= run(iters, false, 0);
  rc |= run(iters, true, 0);
  rc |= run(iters, true, 1);
  return rc == 2 ? 2 : 0;
}
