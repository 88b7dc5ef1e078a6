// The 2x2-block correlation-map kernel compiled WITHOUT packed-fp32 instructions (Makefile: -target-feature -packed-fp32-ops):
// reproduction aid for DESIGN.md section 8 (debug key 9 = 5).
#define LBM_CB_NS cb_nopk
#include "corr_block.inc"

namespace lbm {
int launch_corr_block_nopk(const void* args, unsigned blocks, hipStream_t stream) {
  const cb_nopk::CorrMapArgs a = *static_cast<const cb_nopk::CorrMapArgs*>(args);
  hipLaunchKernelGGL(cb_nopk::corr_maps_block_kernel<0>, dim3(blocks), dim3(256), 0, stream, a);
  return 0;
}
}  // namespace lbm
