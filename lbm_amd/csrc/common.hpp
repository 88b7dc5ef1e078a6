// Shared device/host helpers for the LBM gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LBM_C 256          // embedding dim (lbm/configs/demo.yaml:10)
#define LBM_HEADS 4        // attention heads (demo.yaml:11)
#define LBM_HD 64          // head dim
#define LBM_LEVELS 4       // pyramid levels (demo.yaml:12)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Timing-only ablation switches inside kernels (they make results WRONG on purpose) exist only in the tuning build
// (`make tuning` -> liblbm_hip_tuning.so, -DLBM_TUNING); in the product library the expression is the constant 0 and the
// compiler drops the code behind it.  lbm_hip_debug_set refuses the corresponding keys there.
#ifdef LBM_TUNING
#define LBM_ABL(x) (x)
#else
#define LBM_ABL(x) 0
#endif

namespace lbm {

// ---- error plumbing (never throw across the C ABI) --------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define LBM_REQUIRE(cond, ...)                      \
  do {                                              \
    if (!(cond)) {                                  \
      lbm::set_error(__VA_ARGS__);                  \
      return -1;                                    \
    }                                               \
  } while (0)

// ---- optional per-launch timing (lbm_hip_profile_begin / _end): HIP events on the launch stream ----
// flops / bytes are the ALGORITHMIC figures of the launch, used for the roofline report.
struct ProfScope {
  ProfScope(const char* name, hipStream_t stream, double flops, double bytes);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

// stable storage for ProfScope names built at run time (LBM_PROF_SHAPES); safe to call from several host threads
const char* intern_name(const char* s);

// ---- cross-lane helpers (wave = 64 lanes) -----------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over each aligned group of 16 lanes; every lane of the group gets the total.
// row_ror:n rotates within a 16-lane row (dpp_ctrl 0x120 + n).
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x128>(v);
  v += dpp_f<0x124>(v);
  v += dpp_f<0x122>(v);
  v += dpp_f<0x121>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));
  v = fmaxf(v, dpp_f<0x124>(v));
  v = fmaxf(v, dpp_f<0x122>(v));
  v = fmaxf(v, dpp_f<0x121>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// erf to < 1 ulp (checked against scipy in float64 over [-6, 6]: max abs error 5.8e-8), both ranges evaluated and
// selected so a wave never diverges: a minimax polynomial in a^2 below 0.9277 and 1 - exp(p(|a|)) above.  The library
// erff costs ~1.2 us per 1024 x 1024 GELU epilogue on the per-point chain; this form about a third of that.
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}

__device__ __forceinline__ float gelu_erf(float x) {
  // exact GELU, as nn.GELU() default (transformer.py:348, backbone.py:36)
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// ---- split representation of an fp32 value for the f16 matrix cores: x = hi + lo * 2^-11 (see gemm16.hip) ----
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
  const float c = __builtin_amdgcn_fmed3f(x, -65000.f, 65000.f);
  hi = (_Float16)c;
  lo = (_Float16)((c - (float)hi) * 2048.f);
}
__device__ __forceinline__ void split4(const float4 v, half4& hi, half4& lo) {
  _Float16 h, l;
  split1(v.x, h, l); hi[0] = h; lo[0] = l;
  split1(v.y, h, l); hi[1] = h; lo[1] = l;
  split1(v.z, h, l); hi[2] = h; lo[2] = l;
  split1(v.w, h, l); hi[3] = h; lo[3] = l;
}
// store 4 consecutive values into both planes (8-byte aligned destinations)
__device__ __forceinline__ void st4_split(void* hi_plane, long long plane, long long off, const float4 v) {
  half4 h, l;
  split4(v, h, l);
  _Float16* p = reinterpret_cast<_Float16*>(hi_plane) + off;
  *reinterpret_cast<half4*>(p) = h;
  *reinterpret_cast<half4*>(p + plane) = l;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// the same four values from an f16 tensor (8 bytes; the f16-storage mode keeps memories and value maps that way)
__device__ __forceinline__ float4 ld4(const _Float16* p) {
  const half4 h = *reinterpret_cast<const half4*>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
// raw (unconverted) 4-element loads, for code that wants the conversion at the point of USE -- a conversion written next to the load
// makes the compiler wait for the load right there, which defeats a prefetch
__device__ __forceinline__ float4 ldraw4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ half4 ldraw4(const _Float16* p) { return *reinterpret_cast<const half4*>(p); }
__device__ __forceinline__ float4 widen4(const float4 v) { return v; }
__device__ __forceinline__ float4 widen4(const half4 h) { return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
__device__ __forceinline__ void st4h(_Float16* p, const float4 v) {      // round-to-nearest f16, magnitudes clamped like split1
  half4 h;
  h[0] = (_Float16)__builtin_amdgcn_fmed3f(v.x, -65000.f, 65000.f);
  h[1] = (_Float16)__builtin_amdgcn_fmed3f(v.y, -65000.f, 65000.f);
  h[2] = (_Float16)__builtin_amdgcn_fmed3f(v.z, -65000.f, 65000.f);
  h[3] = (_Float16)__builtin_amdgcn_fmed3f(v.w, -65000.f, 65000.f);
  *reinterpret_cast<half4*>(p) = h;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

}  // namespace lbm
