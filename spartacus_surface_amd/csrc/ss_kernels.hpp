// ss_kernels.hpp - __global__ wrappers of the per-unit bodies and the launch
// interface between the C-ABI translation unit (ss_capi.hip) and the
// per-class instantiation units (inst_*.hip, one per (tile kind, regions,
// spectral region) so that hipcc can build them in parallel).
#pragma once

#include <hip/hip_runtime.h>

#include "ss_common.hpp"
#include "ss_misc_units.hpp"

namespace ss {

constexpr int kBlock = 64;   // one wavefront per workgroup: no LDS sharing, no barriers

// Implemented once per (urban, nreg) in inst_*.hip; ns selects the instance.
// Launches the layer kernel over (ncu x maxlay) units, then the upward and the
// downward column kernels over ncu units, on "stream"; ev[0..3] (optional) are
// recorded before, between and after the three kernels.  Returns 0, the hipError_t
// of the first failing launch / event record, or kLaunchBadRecord when the plan's
// scratch record sizes disagree with the kernels' record types (nothing launched).
// "fallback": the stream the one-lane second pass of the cooperative orders runs on (ordered
// with "stream" by events): that kernel keeps 17 KB of scratch per lane for the units the
// cooperative route declines, and the runtime backs scratch per queue - one queue of a plan
// carries it, whatever the number of pipes.
constexpr int kLaunchBadRecord = -1;
using SolverLauncher = int (*)(int ns, const Launch& L, hipStream_t stream, hipEvent_t* ev, hipStream_t fallback);

#define SS_DECL_LAUNCHER(name) int name(int ns, const Launch& L, hipStream_t stream, hipEvent_t* ev, hipStream_t fallback)
SS_DECL_LAUNCHER(launch_sw_forest_r2);
SS_DECL_LAUNCHER(launch_sw_forest_r3);
SS_DECL_LAUNCHER(launch_sw_urban_r1);
SS_DECL_LAUNCHER(launch_sw_urban_r2);
SS_DECL_LAUNCHER(launch_sw_urban_r3);
SS_DECL_LAUNCHER(launch_lw_forest_r2);
SS_DECL_LAUNCHER(launch_lw_forest_r3);
SS_DECL_LAUNCHER(launch_lw_urban_r1);
SS_DECL_LAUNCHER(launch_lw_urban_r2);
SS_DECL_LAUNCHER(launch_lw_urban_r3);

// The fused layer + upward kernels (k_layer_up: ss_layer_up_sw2.hpp, sym::layer_up_lw2) live in units of their own
// (inst_fuse_{sw,lw}_*.hip): they are compiled WITHOUT machine LICM (csrc/Makefile), which the other kernels of a
// class are better off with.  Launches the kernel for stream count ns over the units of L; hipErrorInvalidValue when the
// order is not a fused one.
#define SS_DECL_FUSED(name) int name(int ns, const Launch& L, hipStream_t stream)
SS_DECL_FUSED(launch_fuse_sw_forest_r2);
SS_DECL_FUSED(launch_fuse_sw_forest_r3);
SS_DECL_FUSED(launch_fuse_sw_urban_r1);
SS_DECL_FUSED(launch_fuse_sw_urban_r2);
SS_DECL_FUSED(launch_fuse_sw_urban_r3);
SS_DECL_FUSED(launch_fuse_lw_forest_r2);
SS_DECL_FUSED(launch_fuse_lw_forest_r3);
SS_DECL_FUSED(launch_fuse_lw_urban_r1);
SS_DECL_FUSED(launch_fuse_lw_urban_r2);
SS_DECL_FUSED(launch_fuse_lw_urban_r3);

}  // namespace ss
