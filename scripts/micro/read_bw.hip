// What does a pure streaming READ deliver on this part, by access shape?  The memory read-out (attn.hip) streams 12..96 KB of contiguous
// slot rows per point, one 1-KB row per wave-load, and sits at 4.6-5.3 TB/s; this prices the alternatives: rows per wave in flight,
// waves per workgroup, persistent grids, 16-byte / 8-byte lanes, nontemporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every wave reads `rows_per_wave` consecutive 1-KB rows, U of them in flight; result folded so that the loads cannot be dropped
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_rows(const float4* __restrict__ x, float* out, long long rows, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long g = w; (g + 1) * rows_per_wave <= rows; g += (long long)gridDim.x * (blockDim.x >> 6)) {      // whole groups only: nothing past the buffer
    const float4* p = x + g * rows_per_wave * 64 + lane;
    for (int r = 0; r < rows_per_wave; r += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (NT) { const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + (long long)(r + u) * 64)); v[u] = make_float4(t[0], t[1], t[2], t[3]); }
        else v[u] = p[(long long)(r + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
static float time_us(F launch, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3f * ms / n;
}

int main() {
  const long long bytes = 512ll << 20;             // 512 MB: beyond the 256 MB Infinity Cache
  const long long rows = bytes / 1024;
  float4* x; float* out;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 1, bytes));
  struct Cfg { int rpw, grid_mult; };
  for (int rpw : {12, 48, 96}) {
    for (int persistent : {0, 1}) {
      const long long waves = rows / rpw;
      const int grid = persistent ? 256 * 8 : (int)((waves + 3) / 4);
#define RUN(U, NT) { const float t = time_us([&] { hipLaunchKernelGGL((read_rows<U, NT>), dim3(grid), dim3(256), 0, 0, x, out, rows, rpw); }, 20); \
      printf("rows/wave %3d  %-10s  U=%d %s  %7.1f us  %5.2f TB/s\n", rpw, persistent ? "persistent" : "one-shot", U, NT ? "nt" : "  ", t, bytes / (t * 1e6)); }
      RUN(4, false) RUN(4, true) RUN(6, false) RUN(12, false) RUN(12, true)
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
