// ss_inst_body.hpp - body shared by the inst_*.hip units.  The including file
// defines SS_INST_URBAN (true/false), SS_INST_NREG (1..3), SS_INST_LW (0/1)
// and SS_INST_NAME (launcher symbol).
#include "ss_kernels.hpp"
#include "ss_column_lw.hpp"
#include "ss_column_sw.hpp"
#include "ss_column_sw_sym.hpp"
#include "ss_column_lw_sym.hpp"
#include "ss_solver_lw.hpp"
#include "ss_solver_sw.hpp"
#include "ss_layer16.hpp"
#include "ss_column16.hpp"
#include "ss_down16.hpp"

#include "ss_kernel_layer_up.hpp"

namespace ss {
namespace {

// Matrix orders above SS_UNROLL_MAX run on runtime loops with their matrices in
// scratch memory: registers do not limit them, latency does, so they are built
// for several waves per SIMD (SS_BIG_WAVES) instead of one.
#ifndef SS_BIG_WAVES
#define SS_BIG_WAVES 4
#endif
#define SS_WAVES(R, NS, small) (((R) * (NS) > SS_UNROLL_MAX) ? SS_BIG_WAVES : (small))

#ifndef SS_MIN_WAVES_LAYER
#define SS_MIN_WAVES_LAYER 1
#endif
#ifndef SS_MIN_WAVES_COLUMN
#define SS_MIN_WAVES_COLUMN 1
#endif
#ifndef SS_MIN_WAVES_DOWN
#define SS_MIN_WAVES_DOWN 1
#endif

// (first pass of the cooperative orders: scalars, element-wise Gamma stores and the small clear-sky
// sub-problem - registers, not scratch: two waves per SIMD with 256 registers each)
#ifndef SS_LAYER_ATTR
#ifdef SS_LAYER_VGPR          // experiment: cap the registers of the layer kernel (VGPRs + AGPRs)
#define SS_LAYER_ATTR __attribute__((amdgpu_num_vgpr(SS_LAYER_VGPR)))
#else
#define SS_LAYER_ATTR
#endif
#endif
template <bool U, int R, int NS, bool LW, int STAGE = 0>
__global__ void __launch_bounds__(kBlock, (STAGE == 1 && r16_layer(R * NS, LW)) ? 2 : SS_WAVES(R, NS, SS_MIN_WAVES_LAYER))
    SS_LAYER_ATTR k_layer(const Launch L) {
#ifndef SS_LAYER_GRID_2D
  // Block -> (tile of 64 column units, layer).  The per-layer inputs of a column
  // sit 8 * nlay bytes apart, so one block uses 8 bytes of every 128-byte line it
  // fetches; the other layers of the same tile use the rest.  Workgroups go to the
  // 8 XCDs round-robin and every XCD has its own L2, so the blocks of one tile get
  // ids that are equal modulo 8 and close together: same L2, same time.
  const int per_group = 8 * L.maxlay;
  const int group = blockIdx.x / per_group, r = blockIdx.x % per_group;
  const int tile = group * 8 + (r & 7), jlay = r >> 3;
#else
  const int tile = blockIdx.x, jlay = blockIdx.y;
#endif
  const int cu = tile * kBlock + threadIdx.x;
  if (cu >= L.ncu) return;
  if constexpr (STAGE == 2 && r16_layer(R * NS, LW)) {
    if (L.eigen_fail && L.eigen_fail[kSlotOneLaneUnits] == 0ULL) return;   // every unit went the cooperative way
  }
  if (LW) layer_unit_lw<U, R, NS, STAGE>(L, cu, jlay);
  else layer_unit_sw<U, R, NS, STAGE>(L, cu, jlay);
}

// cooperative eigen-solver between the two passes of the layer kernel (ss_eigen_coop.hpp):
// block -> (tile of 64 units, layer, quarter of the tile); 4 lanes per unit, 16 units per wave
// (two waves per SIMD since the rotations are no longer accumulated: 84 doubles of state per lane at order 12)
#ifndef SS_EIG_COOP_WAVES
#define SS_EIG_COOP_WAVES (kJacobiAccumulate ? 1 : 2)
#endif
template <int N>
__global__ void __launch_bounds__(kBlock, SS_EIG_COOP_WAVES) k_eig_coop(const Launch L) {
  constexpr int G = coop_lanes(N);
  int cu, jlay;
  if (L.coop_list) {
    if (!coop_list_get(L.coop_list, (unsigned long long)blockIdx.x * (kBlock / G) + threadIdx.x / G, jlay, cu)) return;
  } else {
    const int part = blockIdx.x % G, tl = blockIdx.x / G;
    const int tile = tl / L.maxlay;
    jlay = tl % L.maxlay;
    cu = tile * kBlock + part * (kBlock / G) + (int)threadIdx.x / G;
    if (cu >= L.ncu) return;
    if (jlay >= L.in.nlay[L.cols[cu / L.nspec]]) return;
  }
  if constexpr (coop_order(N))
    coop_eig_unit<N, G>(L.eig + rec_base(jlay, L.maxlay, EigRec<N>::NELEM, cu), kRecStride, (int)threadIdx.x % G,
                        L.eigen_fail ? L.eigen_fail + kSlotOneLaneUnits : nullptr, L.cfg.decline_cooperative == 2);
}

// (k_layer_up: ss_kernel_layer_up.hpp)

// Workgroups of the row-of-16 kernels: 128 threads = 8 units.  The records are tiled for one-lane
// kernels ([element pair][lane][2]: the pairs of 8 neighbouring units make one 128-byte line), and
// workgroups go to the 8 XCDs round-robin, each with its own L2: with 4 units per workgroup the two
// halves of every line were fetched (and written) by two different L2s.
#ifndef SS_BLOCK16
#define SS_BLOCK16 128      // measured on C3 with 4 streams: 64 -> 128 +6 % (whole lines per workgroup), 256 +1.5 %
#endif
constexpr int kBlock16 = SS_BLOCK16;
constexpr int kUnits16 = kBlock16 / r16::kRow;      // units per workgroup
constexpr int kParts16 = kBlock / kUnits16;         // workgroups per tile of 64 units

// Cholesky factors of the cooperative units (W, D L, -D L^-T of the eigen record) on the 16 lanes of a
// DPP row, between the first pass of the layer kernel and the Jacobi kernel; block -> units as k_layer16
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock16, 2) k_pre16(const Launch L) {
  constexpr int N = R * NS;
  constexpr int UPB = kUnits16;
  int cu, jlay;
  if (L.coop_list) {
    if (!coop_list_get(L.coop_list, (unsigned long long)blockIdx.x * UPB + threadIdx.x / r16::kRow, jlay, cu)) return;
  } else {
    const int part = blockIdx.x % kParts16, tl = blockIdx.x / kParts16;
    const int tile = tl / L.maxlay;
    jlay = tl % L.maxlay;
    cu = tile * kBlock + part * UPB + (int)threadIdx.x / r16::kRow;
    if (cu >= L.ncu) return;
    if (jlay >= L.in.nlay[L.cols[cu / L.nspec]]) return;
  }
  if constexpr (coop_order(N)) {
    double* erec = L.eig + rec_base(jlay, L.maxlay, EigRec<N>::NELEM, cu);
    if (!(rec_elem_load(erec, kRecStride, EigRec<N>::STATUS) > 0.5)) return;     // same for the whole row
    r16::eig_pre_scalars<U, R, NS, LW>(L.lg, erec, kRecStride, L.eigen_fail ? L.eigen_fail + kSlotOneLaneUnits : nullptr);
  }
}

// layer matrices of the cooperative units on the 16 lanes of a DPP row (ss_layer16.hpp):
// block -> (tile of 64 units, layer, sixteenth of the tile); 4 units per wave
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock16, 2) k_layer16(const Launch L) {
  constexpr int N = R * NS, D = R;
  constexpr int UPB = kUnits16;
  int cu, jlay;
  if (L.coop_list) {
    if (!coop_list_get(L.coop_list, (unsigned long long)blockIdx.x * UPB + threadIdx.x / r16::kRow, jlay, cu)) return;
  } else {
    const int part = blockIdx.x % kParts16, tl = blockIdx.x / kParts16;
    const int tile = tl / L.maxlay;
    jlay = tl % L.maxlay;
    cu = tile * kBlock + part * UPB + (int)threadIdx.x / r16::kRow;
    if (cu >= L.ncu) return;
    if (jlay >= L.in.nlay[L.cols[cu / L.nspec]]) return;
  }
  if constexpr (r16_layer(N, LW)) {
    const double* erec = L.eig + rec_base(jlay, L.maxlay, EigRec<N>::NELEM, cu);
    if (!(rec_elem_load(erec, kRecStride, EigRec<N>::STATUS) > 0.5)) return;     // same for the whole row
    const r16::EigenvectorsFromScalars<U, R, NS, LW> load_ev{&L.lg};
    if constexpr (LW)
      r16::layer_lw<N>(erec, kRecStride, L.lay + rec_base(jlay, L.maxlay, LwRec<N>::NELEM, cu), load_ev);
    else
      r16::layer_sw<N, D>(erec, kRecStride, L.lay + rec_base(jlay, L.maxlay, SwRec<N, D>::NELEM, cu), load_ev);
  }
}

// upward sweep of the cooperative orders: one column unit on the 16 lanes of a DPP row (ss_column16.hpp)
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock16, 2) k_column16(const Launch L) {
  const int cu = blockIdx.x * kUnits16 + (int)threadIdx.x / r16::kRow;
  if (cu >= L.ncu) return;
  if constexpr (r16_layer(R * NS, LW)) {
    if constexpr (LW) r16::column_up_lw<U, R, NS>(L, cu);
    else r16::column_up_sw<U, R, NS>(L, cu);
  }
}

// upward sweep (adding method)
#ifndef SS_COLUMN_WAVES_ATTR
#define SS_COLUMN_WAVES_ATTR
#endif
#ifndef SS_DOWN_WAVES_ATTR
#define SS_DOWN_WAVES_ATTR
#endif
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock, SS_WAVES(R, NS, SS_MIN_WAVES_COLUMN)) SS_COLUMN_WAVES_ATTR k_column(const Launch L) {
  const int cu = blockIdx.x * kBlock + threadIdx.x;
  if (cu >= L.ncu) return;
  if (LW) column_up_lw<U, R, NS>(L, cu);
  else column_up_sw<U, R, NS>(L, cu);
}

// downward passes of the cooperative orders: one column unit (both passes) on the 16 lanes of a DPP row
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock16, 2) k_down16(const Launch L) {
  const int cu = blockIdx.x * kUnits16 + (int)threadIdx.x / r16::kRow;
  if (cu >= L.ncu) return;
  if constexpr (r16_layer(R * NS, LW)) {
    if constexpr (LW) r16::column_down_lw<U, R, NS>(L, cu);
    else r16::column_down_sw<U, R, NS>(L, cu);
  }
}

// downward flux sweep: vectors only, several waves per SIMD
template <bool U, int R, int NS, bool LW>
__global__ void __launch_bounds__(kBlock, SS_WAVES(R, NS, SS_MIN_WAVES_DOWN)) SS_DOWN_WAVES_ATTR k_down(const Launch L) {
  const int unit = blockIdx.x * kBlock + threadIdx.x;   // (column unit, pass), pass fastest
  const int cu = unit >> 1;
  if (cu >= L.ncu) return;
#if defined(SS_DOWN_STAGGER)
  // experiment: half of the waves start half a layer step late, so that the load phases of one half fall into the
  // arithmetic phases of the other
  if ((blockIdx.x >> 3) & 1) { for (int k = 0; k < SS_DOWN_STAGGER; ++k) __builtin_amdgcn_s_sleep(127); }
#endif
  if (LW) column_down_lw<U, R, NS>(L, cu, unit & 1, nullptr);
  else column_down_sw<U, R, NS>(L, cu, unit & 1);
}

// downward flux sweep of the shortwave with both passes on one lane (ss_down_sw_both.hpp): lane = column unit
template <bool U, int R, int NS>
__global__ void __launch_bounds__(kBlock, 1) k_down_both(const Launch L) {
  const int cu = blockIdx.x * kBlock + threadIdx.x;
  if (cu >= L.ncu) return;
#if defined(SS_DOWN_STAGGER)
  if ((blockIdx.x >> 3) & 1) { for (int k = 0; k < SS_DOWN_STAGGER; ++k) __builtin_amdgcn_s_sleep(127); }
#endif
  if constexpr (sw_down_both(R * NS)) column_down_sw_both<U, R, NS>(L, cu);
}

// the downward kernel of a class: both passes on one lane where the vectors of both fit (shortwave, small orders),
// else one lane per (column unit, pass)
template <int NS>
hipError_t launch_down(const Launch& L, hipStream_t stream) {
  if constexpr (!SS_INST_LW && sw_down_both(SS_INST_NREG * NS)) {
    hipLaunchKernelGGL((k_down_both<SS_INST_URBAN, SS_INST_NREG, NS>), dim3((unsigned)((L.ncu + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, L);
  } else {
    const unsigned gd = (unsigned)((2L * L.ncu + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_down<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gd), dim3(kBlock), 0,
                       stream, L);
  }
  return hipGetLastError();
}

#define SS_TRY(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)

template <int NS>
int launch_ns(const Launch& L, hipStream_t stream, hipEvent_t* ev, hipStream_t fallback) {
  {
    // the plan sized the scratch with its own formula: it must match the records
    constexpr int N = SS_INST_NREG * NS, D = SS_INST_NREG;
    constexpr int lay_ne = fused_order(N, SS_INST_LW != 0) ? 0 : (SS_INST_LW ? LwLayRec<N>::NELEM : SwLayRec<N, D>::NELEM);
    constexpr int col_ne = SS_INST_LW ? LwColRec<N>::NELEM : SwColRec<N, D>::NELEM;
    if (L.lay_nelem != lay_ne || L.col_nelem != col_ne) return kLaunchBadRecord;
  }
  if (L.ncu <= 0 || L.maxlay <= 0) return 0;
  const unsigned gx = (unsigned)((L.ncu + kBlock - 1) / kBlock);
  if (ev) SS_TRY(hipEventRecord(ev[0], stream));
  if constexpr (fused_order(SS_INST_NREG * NS, SS_INST_LW != 0)) {
    // fused form: one kernel for the layer matrices and the upward sweep (the "column" span is empty)
#if defined(SS_INST_FUSE_NAME)
    // (the fused kernel is compiled in a unit of its own, see ss_kernels.hpp)
    { const int rc_ = SS_INST_FUSE_NAME(NS, L, stream); if (rc_ != 0) return rc_; }
#else
    hipLaunchKernelGGL((k_layer_up<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gx), dim3(kBlock), 0,
                       stream, L);
    SS_TRY(hipGetLastError());
#endif
    if (ev) SS_TRY(hipEventRecord(ev[1], stream));
    if (ev) SS_TRY(hipEventRecord(ev[2], stream));
    SS_TRY(launch_down<NS>(L, stream));
    if (ev) SS_TRY(hipEventRecord(ev[3], stream));
    return 0;
  }
#ifndef SS_LAYER_GRID_2D
  const dim3 glayer(((gx + 7) / 8) * 8 * (unsigned)L.maxlay);
#else
  const dim3 glayer(gx, (unsigned)L.maxlay);
#endif
  if constexpr (coop_order(SS_INST_NREG * NS)) {
    // W of every full problem -> cooperative Jacobi -> layer matrices with the eigen-decomposition from the record
    if (L.eig_nelem != EigRec<SS_INST_NREG * NS>::NELEM || !L.eig) return kLaunchBadRecord;
    if (L.coop_list) SS_TRY(hipMemsetAsync(L.coop_list, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL((k_layer<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0, 1>), glayer, dim3(kBlock), 0,
                       stream, L);
    SS_TRY(hipGetLastError());
    if constexpr (r16_layer(SS_INST_NREG * NS, SS_INST_LW != 0)) {
      hipLaunchKernelGGL((k_pre16<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gx * (unsigned)L.maxlay * kParts16), dim3(kBlock16), 0, stream, L);
      SS_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL((k_eig_coop<SS_INST_NREG * NS>), dim3(gx * (unsigned)L.maxlay * coop_lanes(SS_INST_NREG * NS)), dim3(kBlock), 0, stream, L);
    SS_TRY(hipGetLastError());
    if constexpr (r16_layer(SS_INST_NREG * NS, SS_INST_LW != 0)) {
      hipLaunchKernelGGL((k_layer16<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gx * (unsigned)L.maxlay * kParts16),
                         dim3(kBlock16), 0, stream, L);
      SS_TRY(hipGetLastError());
    }
    if (fallback && fallback != stream) {
      // second pass on the plan's fallback stream, between two events of the plan's pool (ev[4], ev[5])
      if (!ev) return (int)hipErrorInvalidValue;
      const hipEvent_t a = ev[4], b = ev[5];
      SS_TRY(hipEventRecord(a, stream));
      SS_TRY(hipStreamWaitEvent(fallback, a, 0));
      hipLaunchKernelGGL((k_layer<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0, 2>), glayer, dim3(kBlock), 0,
                         fallback, L);
      SS_TRY(hipGetLastError());
      SS_TRY(hipEventRecord(b, fallback));
      SS_TRY(hipStreamWaitEvent(stream, b, 0));
    } else {
      hipLaunchKernelGGL((k_layer<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0, 2>), glayer, dim3(kBlock), 0,
                         stream, L);
      SS_TRY(hipGetLastError());
    }
  } else {
    hipLaunchKernelGGL((k_layer<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), glayer, dim3(kBlock), 0,
                       stream, L);
    SS_TRY(hipGetLastError());
  }
  if (ev) SS_TRY(hipEventRecord(ev[1], stream));
  if constexpr (r16_layer(SS_INST_NREG * NS, SS_INST_LW != 0)) {
    constexpr unsigned upb = kUnits16;
    hipLaunchKernelGGL((k_column16<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>),
                       dim3(((unsigned)L.ncu + upb - 1) / upb), dim3(kBlock16), 0, stream, L);
  } else {
    hipLaunchKernelGGL((k_column<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>), dim3(gx), dim3(kBlock), 0,
                       stream, L);
  }
  SS_TRY(hipGetLastError());
  if (ev) SS_TRY(hipEventRecord(ev[2], stream));
  if constexpr (r16_layer(SS_INST_NREG * NS, SS_INST_LW != 0)) {
    constexpr unsigned upb = kUnits16;
    hipLaunchKernelGGL((k_down16<SS_INST_URBAN, SS_INST_NREG, NS, SS_INST_LW != 0>),
                       dim3(((unsigned)L.ncu + upb - 1) / upb), dim3(kBlock16), 0, stream, L);
    SS_TRY(hipGetLastError());
  } else {
    SS_TRY(launch_down<NS>(L, stream));
  }
  if (ev) SS_TRY(hipEventRecord(ev[3], stream));
  return 0;
}

}  // namespace

int SS_INST_NAME(int ns, const Launch& L, hipStream_t stream, hipEvent_t* ev, hipStream_t fallback) {
#ifdef SS_ONLY_NS   // experiment builds: a single stream count per unit
  if (ns == SS_ONLY_NS) return launch_ns<SS_ONLY_NS>(L, stream, ev, fallback);
  return (int)hipErrorInvalidValue;
#else
  switch (ns) {
    case 1: return launch_ns<1>(L, stream, ev, fallback);
    case 2: return launch_ns<2>(L, stream, ev, fallback);
    case 3: return launch_ns<3>(L, stream, ev, fallback);
    case 4: return launch_ns<4>(L, stream, ev, fallback);
    case 5: return launch_ns<5>(L, stream, ev, fallback);
    case 6: return launch_ns<6>(L, stream, ev, fallback);
    case 7: return launch_ns<7>(L, stream, ev, fallback);
    case 8: return launch_ns<8>(L, stream, ev, fallback);
    default: return (int)hipErrorInvalidValue;
  }
#endif
}

}  // namespace ss
