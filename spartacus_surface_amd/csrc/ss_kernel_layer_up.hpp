// ss_kernel_layer_up.hpp - the fused layer + upward kernel (one column unit per lane), shared by the
// instantiation units (ss_inst_body.hpp: longwave experiments) and the units of the shortwave fused kernel
// (ss_inst_fuse_body.hpp)
#pragma once
#include "ss_kernels.hpp"
#include "ss_column_lw.hpp"
#include "ss_column_sw.hpp"
#include "ss_column_sw_sym.hpp"
#include "ss_column_lw_sym.hpp"

namespace ss {
namespace {

// layer matrices + upward sweep of a column unit on one lane (orders that live in registers)
#ifndef SS_LAYER_UP_ATTR
#ifdef SS_LAYER_UP_VGPR
#define SS_LAYER_UP_ATTR __attribute__((amdgpu_num_vgpr(SS_LAYER_UP_VGPR)))
#else
#define SS_LAYER_UP_ATTR
#endif
#endif
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock, 1) SS_LAYER_UP_ATTR k_layer_up(const Launch L) {
  const int cu = blockIdx.x * kBlock + threadIdx.x;
  if (cu >= L.ncu) return;
  if constexpr (fused_order(R * NS, LW)) {
    if constexpr (LW) layer_up_lw<U, R, NS>(L, cu);
    else layer_up_sw_any<U, R, NS>(L, cu);
  }
}

}  // namespace
}  // namespace ss
