// Head read-outs fused with the state update (transformer.py:119-120, 136-138, 147-149;
// lbm_online.py:143-156).  One wave per point, lane = 4 channels.
#include "final.hpp"
#include "../../include/lbm_hip.h"

#include <mutex>

namespace lbm {

// offset = tanh(W3 · hidden + b3) * stride ;  pred = ref_last + offset
__global__ __launch_bounds__(256) void offset_final_kernel(const float* __restrict__ hidden, const float* __restrict__ w3,
                                                           const float* __restrict__ b3, const float* __restrict__ ref,
                                                           float stride_px, float* __restrict__ offset,
                                                           float* __restrict__ pred, int R) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float4 hv = ld4(hidden + (long long)row * LBM_C + lane * 4);
  const float4 wa = ld4(w3 + lane * 4), wb = ld4(w3 + LBM_C + lane * 4);
  const float s0 = wave_sum(hv.x * wa.x + hv.y * wa.y + hv.z * wa.z + hv.w * wa.w) + b3[0];
  const float s1 = wave_sum(hv.x * wb.x + hv.y * wb.y + hv.z * wb.z + hv.w * wb.w) + b3[1];
  if (lane == 0) {
    const float ox = tanhf(s0) * stride_px, oy = tanhf(s1) * stride_px;
    offset[row * 2] = ox;
    offset[row * 2 + 1] = oy;
    pred[row * 2] = ref[row * 2] + ox;
    pred[row * 2 + 1] = ref[row * 2 + 1] + oy;
  }
}

int offset_final(const float* hidden, const float* w3, const float* b3, const float* ref, float stride_px, float* offset,
                 float* pred, int R, hipStream_t stream) {
  if (R == 0) return 0;
  LBM_REQUIRE(hidden && w3 && b3 && ref && offset && pred, "offset_final: null argument");
  ProfScope prof("offset_final", stream, 0.0, 4.0 * LBM_C * R);
  hipLaunchKernelGGL(offset_final_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, hidden, w3, b3, ref, stride_px, offset, pred, R);
  return check_launch("offset_final");
}

__global__ __launch_bounds__(256) void state_final_kernel(StateFinalArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.R) return;
  const float4 hv = ld4(a.vis_hidden + (long long)row * LBM_C + lane * 4);
  const float4 wa = ld4(a.w3 + lane * 4), wb = ld4(a.w3 + LBM_C + lane * 4);
  const float vis = wave_sum(hv.x * wa.x + hv.y * wa.y + hv.z * wa.z + hv.w * wa.w) + a.b3[0];
  const float rho = wave_sum(hv.x * wb.x + hv.y * wb.y + hv.z * wb.z + hv.w * wb.w) + a.b3[1];
  const bool queried = a.queried ? a.queried[row] != 0 : true;

  // new memory rows: stream = post-layer token; collision = queried ? collision token : initial query
  const long long off = (long long)row * LBM_C + lane * 4;
  const float4 sv = ld4(a.stream_tok + off);
  const float4 cv = queried ? ld4(a.coll_tok + off) : ld4(a.query_init + off);
  if (a.stream_out) st4(a.stream_out + off, sv);
  if (a.coll_out) st4(a.coll_out + off, cv);
  if (a.stream_mem) {
    const long long so = ((long long)row * a.M + a.ring_head) * LBM_C + lane * 4;
    if (a.mem_f16) {
      st4h(reinterpret_cast<_Float16*>(a.stream_mem) + so, sv);
      st4h(reinterpret_cast<_Float16*>(a.coll_mem) + so, cv);
    } else {
      st4(a.stream_mem + so, sv);
      st4(a.coll_mem + so, cv);
    }
  }
  if (lane == 0) {
    a.vis[row] = vis;
    a.rho[row] = rho;
    const float sg = 1.0f / (1.0f + expf(-vis));
    if (a.vis_flag) a.vis_flag[row] = sg > a.vis_threshold ? 1 : 0;
    if (a.rho_prob) a.rho_prob[row] = 1.0f / (1.0f + expf(-rho));
    if (a.vis_mask) {
      a.vis_mask[(long long)row * a.M + a.ring_head] = sg < a.vis_threshold ? 1 : 0;   // lbm_online.py:146
      a.mem_mask[(long long)row * a.M + a.ring_head] = queried ? 0 : 1;                 // lbm_online.py:145
    }
    if (a.coords) {
      // lbm_online.py:150-156: coord = ref + offset, rescaled to the caller's frame size
      const float x = a.ref[row * 2] + a.offset[row * 2];
      const float y = a.ref[row * 2 + 1] + a.offset[row * 2 + 1];
      a.coords[row * 2] = x / a.net_w * a.img_w;
      a.coords[row * 2 + 1] = y / a.net_h * a.img_h;
    }
  }
}

int state_final(const StateFinalArgs& a, hipStream_t stream) {
  if (a.R == 0) return 0;
  LBM_REQUIRE(a.vis_hidden && a.w3 && a.b3 && a.stream_tok && a.coll_tok && a.query_init && a.vis && a.rho,
              "state_final: null argument");
  LBM_REQUIRE(a.stream_mem == nullptr || (a.coll_mem && a.M >= 1 && a.ring_head >= 0 && a.ring_head < a.M),
              "state_final: bad ring description");
  LBM_REQUIRE(a.vis_mask == nullptr || (a.mem_mask && a.M >= 1 && a.ring_head >= 0 && a.ring_head < a.M),
              "state_final: bad mask description");
  LBM_REQUIRE(a.coords == nullptr || (a.ref && a.offset && a.net_w > 0 && a.net_h > 0), "state_final: coords need ref/offset");
  ProfScope prof("state_final", stream, 0.0, 4.0 * LBM_C * a.R * 5.0);
  hipLaunchKernelGGL(state_final_kernel, dim3((a.R + 3) / 4), dim3(256), 0, stream, a);
  return check_launch("state_final");
}

// ---- dynamic point set ---------------------------------------------------------------------------------------------
// one workgroup: src[rank] = i for every kept old row i (rank = number of kept rows before it), src[N_old] = kept count
__global__ __launch_bounds__(1024) void update_scan_kernel(const uint8_t* __restrict__ remove_mask, int N_old, int* __restrict__ src) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < N_old; i0 += 1024) {
    const int i = i0 + tid;
    const int keep = (i < N_old && remove_mask[i] == 0) ? 1 : 0;
    const unsigned long long b = __ballot(keep);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (keep) src[off + before] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) src[N_old] = base;
}

// one wave per row of the new state
__global__ __launch_bounds__(256) void update_gather_kernel(UpdatePointsArgs a) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= a.N_new) return;
  const int n_prev = a.scratch[a.N_old];
  const bool fresh = a.update_mask[d] != 0;
  // old row that moves here, if any.  A surviving row whose point is re-planted (fresh, d < n_prev) gets the new query but
  // KEEPS its memories, as the reference does (lbm_online.py:93-98 copy the memories of every surviving row)
  const int s = d < n_prev ? a.scratch[d] : -1;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 q = zero;
  if (fresh) q = ld4(a.new_query + (long long)d * LBM_C + lane * 4);
  else if (s >= 0) q = ld4(a.old_query + (long long)s * LBM_C + lane * 4);
  st4(a.query + (long long)d * LBM_C + lane * 4, q);
  for (int j = 0; j < a.M; ++j) {
    const long long o = ((long long)d * a.M + j) * LBM_C + lane * 4;
    const long long so = ((long long)s * a.M + j) * LBM_C + lane * 4;
    st4(a.stream + o, s >= 0 ? ld4(a.old_stream + so) : zero);
    st4(a.coll + o, s >= 0 ? ld4(a.old_coll + so) : zero);
  }
  for (int j = lane; j < a.M; j += 64) {
    a.vis[(long long)d * a.M + j] = s >= 0 ? a.old_vis[(long long)s * a.M + j] : 1;   // fresh / empty rows: every slot ignored
    a.mem[(long long)d * a.M + j] = s >= 0 ? a.old_mem[(long long)s * a.M + j] : 1;
  }
}

int update_points(const UpdatePointsArgs& a, hipStream_t stream) {
  LBM_REQUIRE(a.N_old >= 0 && a.N_new >= 0 && a.M >= 1, "update_points: bad sizes");
  if (a.N_new == 0) return 0;
  LBM_REQUIRE(a.update_mask && a.new_query && a.query && a.stream && a.coll && a.vis && a.mem && a.scratch,
              "update_points: null argument");
  LBM_REQUIRE(a.N_old == 0 || (a.remove_mask && a.old_query && a.old_stream && a.old_coll && a.old_vis && a.old_mem),
              "update_points: null old state");
  LBM_REQUIRE(a.query != a.old_query && a.stream != a.old_stream && a.coll != a.old_coll, "update_points: the new state must not alias the old one");
  ProfScope prof("update_points", stream, 0.0, 8.0 * LBM_C * (double)a.N_new * (2.0 * a.M + 1.0));
  hipLaunchKernelGGL(update_scan_kernel, dim3(1), dim3(1024), 0, stream, a.remove_mask, a.N_old, a.scratch);
  if (check_launch("update_scan")) return -1;
  hipLaunchKernelGGL(update_gather_kernel, dim3((a.N_new + 3) / 4), dim3(256), 0, stream, a);
  return check_launch("update_gather");
}

// ---- the same edit with the scan's scratch in a static device array (lbm_hip_update_points_static) -----------------------
// `scratch` below is the HOST-side stand-in of the array: its device address is looked up once per device with
// hipGetSymbolAddress (each GPU holds its own copy) and only that pointer is ever handed to a kernel.
static __device__ __align__(256) int scratch[LBM_UPDATE_STATIC_MAX_POINTS + 1];
static int* g_scratch_ptr[LBM_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

int update_points_static(UpdatePointsArgs a, hipStream_t stream) {
  LBM_REQUIRE(a.N_old >= 0 && a.N_old <= LBM_UPDATE_STATIC_MAX_POINTS,
              "update_points_static: N_old = %d exceeds the static scratch (%d points); use lbm_hip_update_points with a caller-provided scratch",
              a.N_old, LBM_UPDATE_STATIC_MAX_POINTS);
  int dev = -1;
  LBM_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < LBM_MAX_DEVICES, "update_points_static: no current device (or index >= %d)", LBM_MAX_DEVICES);
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
      void* p = nullptr;
      const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      LBM_REQUIRE(e == hipSuccess && p != nullptr, "update_points_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      g_scratch_ptr[dev] = static_cast<int*>(p);
    }
    a.scratch = g_scratch_ptr[dev];
  }
  return update_points(a, stream);
}

}  // namespace lbm
