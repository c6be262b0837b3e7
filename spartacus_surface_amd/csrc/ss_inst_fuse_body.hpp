// ss_inst_fuse_body.hpp - body of the inst_fuse_{sw,lw}_*.hip units: the fused layer + upward kernel of one
// (tile kind, regions, spectral region) class, every stream count whose order is a fused one.  The including file
// defines SS_INST_URBAN, SS_INST_NREG, SS_INST_LW and SS_INST_FUSE_NAME.
#include "ss_kernel_layer_up.hpp"

namespace ss {
namespace {

template <int NS>
int launch_fused_ns(const Launch& L, hipStream_t stream) {
  if constexpr (fused_order(SS_INST_NREG * NS, SS_INST_LW != 0)) {
    if (L.ncu <= 0 || L.maxlay <= 0) return 0;
    const unsigned gx = (unsigned)((L.ncu + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_layer_up<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gx), dim3(kBlock), 0, stream, L);
    return (int)hipGetLastError();
  } else {
    return (int)hipErrorInvalidValue;
  }
}

}  // namespace

int SS_INST_FUSE_NAME(int ns, const Launch& L, hipStream_t stream) {
#ifdef SS_ONLY_NS   // experiment builds: a single stream count per unit
  if (ns == SS_ONLY_NS) return launch_fused_ns<SS_ONLY_NS>(L, stream);
  return (int)hipErrorInvalidValue;
#else
  switch (ns) {
    case 1: return launch_fused_ns<1>(L, stream);
    case 2: return launch_fused_ns<2>(L, stream);
    case 3: return launch_fused_ns<3>(L, stream);
    case 4: return launch_fused_ns<4>(L, stream);
    case 5: return launch_fused_ns<5>(L, stream);
    case 6: return launch_fused_ns<6>(L, stream);
    default: return (int)hipErrorInvalidValue;
  }
#endif
}

}  // namespace ss
