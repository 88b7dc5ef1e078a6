// The 2x2-block correlation-map kernel built WITH packed-fp32 instructions (the default flags): reproduction aid only
// (lbm_hip_debug_set key 9 = 1..4).  The product path uses the build in pyramid.hip.
#define LBM_CB_NS cb_pk
#include "corr_block.inc"

namespace lbm {
int launch_corr_block_packed(const void* args, int mode, unsigned blocks, hipStream_t stream) {
  const cb_pk::CorrMapArgs a = *static_cast<const cb_pk::CorrMapArgs*>(args);
  switch (mode) {
    case 1: hipLaunchKernelGGL(cb_pk::corr_maps_block_kernel<1>, dim3(blocks), dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(cb_pk::corr_maps_block_kernel<2>, dim3(blocks), dim3(256), 0, stream, a); break;
    case 3: hipLaunchKernelGGL(cb_pk::corr_maps_block_kernel<3>, dim3(blocks), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL(cb_pk::corr_maps_block_kernel<0>, dim3(blocks), dim3(256), 0, stream, a); break;
  }
  return 0;
}
}  // namespace lbm
