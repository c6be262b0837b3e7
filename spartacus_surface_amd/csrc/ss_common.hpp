// ss_common.hpp - shared definitions of the gfx950 SPARTACUS-Surface kernels.
//
// Execution model: one wavefront lane owns one independent
// problem for its whole lifetime - a (column, band, layer) unit in the layer
// kernels, a (column, band) unit in the column kernels.  All small dense
// algebra of a unit lives in that lane's registers (compile-time matrix orders,
// fully unrolled for n <= 6); units exchange data only through HBM scratch laid
// out [layer][element][unit] so that the 64 lanes of a wave always touch 64
// consecutive doubles (512 B, coalesced).
//
// The per-unit bodies are __host__ __device__ so that a TEST-ONLY host build
// (tests/emul) can run them under sanitizers / gdb; the product library only
// ever launches them as HIP kernels.
#pragma once

#include <cmath>
#include <cstdint>
#include <cfloat>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SS_HD __host__ __device__ __forceinline__
#define SS_HD_NOINLINE __host__ __device__ __noinline__
#else
#define SS_HD inline
#define SS_HD_NOINLINE
#endif

// Loop unrolling policy: orders up to SS_UNROLL_MAX are fully unrolled so that
// every matrix element has a compile-time register; larger orders keep rolled
// loops (arrays then live in scratch memory).
#define SS_UNROLL_MAX 6
// Smallest matrix order that runs with one problem on the 16 lanes of a DPP row (ss_row16.hpp).
// 7 = everything that does not fit the registers of one lane.  -DSS_COOP_MIN_ORDER=6 is the
// measured experiment "multi-lane cooperative kernels for the headline order" (DESIGN.md section 3).
#ifndef SS_COOP_MIN_ORDER
#define SS_COOP_MIN_ORDER (SS_UNROLL_MAX + 1)
#endif
// Shortwave, orders 3 .. SS_UNROLL_MAX: layer matrices and adding method of a column unit on ONE lane, no layer
// record (ss_layer_up_sw2.hpp; round 4: 3.3 + 2.9 ms as two kernels -> 5.4 ms, 15 KB per column less traffic).
// -DSS_NO_SYM_FUSE_SW2 builds the three-kernel form of round 3.
#if !defined(SS_NO_SYM_FUSE_SW2) && !defined(SS_SYM_FUSE_SW) && !defined(SS_FUSE_SW) && !defined(SS_FUSE_LW) \
    && !defined(SS_NO_SYM_LAYER) && !defined(SS_NO_SYM_PACK_SW) && !defined(SS_NO_SYM_PACK_COL_SW) && !defined(SS_NO_SW_DIRVEC)
#ifndef SS_SYM_FUSE_SW2
#define SS_SYM_FUSE_SW2 1
#endif
// ... and the longwave likewise (sym::layer_up_lw2, ss_column_lw_sym.hpp: 3.05 + 2.0 ms -> 4.96 ms, 14 KB per
// column less traffic); -DSS_NO_SYM_FUSE_LW2: the three-kernel form
#if !defined(SS_NO_SYM_FUSE_LW2) && !defined(SS_SYM_FUSE_LW) && !defined(SS_SYM_FUSE_LW2)
#define SS_SYM_FUSE_LW2 1
#endif
#endif
#define SS_PRAGMA(x) _Pragma(#x)
#define SS_UNROLL_N(N) SS_PRAGMA(unroll((N) <= SS_UNROLL_MAX ? (N) : 1))
#define SS_UNROLL_NN(N) SS_PRAGMA(unroll((N) <= SS_UNROLL_MAX ? (N) * (N) : 1))
#define SS_UNROLL_FULL SS_PRAGMA(unroll)
// Loops of the symmetric routines (ss_layer_sym.hpp, ss_eigen_sym.hpp), which serve the register-resident orders
// AND, with rolled loops and their matrices in scratch, the orders above 16: full unrolling up to SS_UNROLL_MAX
// (no trip count exceeds N * N there), none beyond
#define SS_SYM_UNROLL(N) SS_PRAGMA(unroll((N) <= SS_UNROLL_MAX ? (N) * (N) : 1))

// Compiler-level fence that keeps the memory accesses written before it before
// it in the instruction stream (no hardware wait): used to issue a whole group
// of loads ahead of the arithmetic that overlaps with their latency.
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SS_SCHED_FENCE() do {} while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_LOAD_FENCE() asm volatile("" ::: "memory")
#else
#define SS_LOAD_FENCE() do {} while (0)
#endif

// Ordering point for ONE value: the (empty) volatile asm "modifies" x, so everything that
// defines x comes before it, everything that uses x after it, and volatile asms keep their program
// order among themselves.  A chain of them pins the order of otherwise independent steps of fully
// unrolled code - the scheduler then cannot run the loads / broadcasts of ten steps ahead of the
// arithmetic of the first (it does, with a 512-register budget, and the register file overflows).
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_PIN(x) asm volatile("" : "+v"(x))
#else
#define SS_PIN(x) do {} while (0)
#endif

namespace ss {

// f(std::integral_constant<int, B>), ..., f(std::integral_constant<int, E - 1>): a loop whose index is a
// compile-time constant in the body (packed-triangle index arithmetic must never reach the register allocator
// as a run-time subscript: an array indexed that way is placed in scratch memory)
template <int B, int E, class F>
SS_HD void static_loop(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_loop<B + 1, E>(f);
  }
}

// Per-lane staging space in LDS for the downward kernels: kLaneLdsSlots doubles
// per lane, lane-interleaved (a wave's access to one slot is one conflict-free
// 512-byte row).  Their blocks are one wave (kBlock = 64) and they run one wave
// per SIMD, so four blocks share the 160 KB of a CU: 40 KB = 80 doubles per lane.
// Host builds (the emulator of tests/emul) use a plain array.
#ifndef SS_LANE_LDS_SLOTS
#define SS_LANE_LDS_SLOTS 80      // (experiment builds: 40 = two waves per SIMD in the kernels that stage in LDS)
#endif
constexpr int kLaneLdsSlots = SS_LANE_LDS_SLOTS;
template <bool ON>
struct LaneLds {
  static constexpr int kSlots = kLaneLdsSlots;
#if defined(__HIP_DEVICE_COMPILE__)
  double* p;
  SS_HD LaneLds() {
    __shared__ double lds[kLaneLdsSlots * 64];
    p = lds + threadIdx.x;
  }
  SS_HD void put(int slot, double v) const { p[slot * 64] = v; }
  SS_HD double get(int slot) const { return p[slot * 64]; }
  // the same slot of the NEXT lane (an even lane borrowing what its odd neighbour leaves unused)
  SS_HD void put_next(int slot, double v) const { p[slot * 64 + 1] = v; }
  SS_HD double get_next(int slot) const { return p[slot * 64 + 1]; }
  // a slot of ANY lane of the wave (the transposed write-out of the downward kernels)
  SS_HD double get_at(int slot, int lane) const { return (p - threadIdx.x)[slot * 64 + lane]; }
#else
  mutable double a[kLaneLdsSlots] = {};
  mutable double b[kLaneLdsSlots] = {};
  SS_HD void put(int slot, double v) const { a[slot] = v; }
  SS_HD double get(int slot) const { return a[slot]; }
  SS_HD void put_next(int slot, double v) const { b[slot] = v; }
  SS_HD double get_next(int slot) const { return b[slot]; }
#endif
};
// A smaller per-lane scratch of SLOTS doubles, same addressing (the graded-order permutation of the
// symmetrised layer route, ss_layer_sym.hpp, in kernels that stage nothing else in LDS: 36 slots =
// 18 KB per wave, two waves per SIMD fit the 160 KB of a CU)
template <int SLOTS>
struct LaneLdsN {
  static constexpr int kSlots = SLOTS;
#if defined(__HIP_DEVICE_COMPILE__)
  double* p;
  SS_HD LaneLdsN() {
    __shared__ double lds[SLOTS * 64];
    p = lds + threadIdx.x;
  }
  SS_HD void put(int slot, double v) const { p[slot * 64] = v; }
  SS_HD double get(int slot) const { return p[slot * 64]; }
#else
  mutable double a[SLOTS] = {};
  SS_HD void put(int slot, double v) const { a[slot] = v; }
  SS_HD double get(int slot) const { return a[slot]; }
#endif
};
// Do the (column, pass) lanes of this wave - a full wave of 32 column units x 2 passes, one band - own one
// CONTIGUOUS block of the caller's per-layer arrays: consecutive columns, equal layer counts, layers packed in
// column order?  Then the staged per-layer outputs of the downward kernels leave through transposed stores -
// every store instruction writes 512 consecutive bytes, four complete lines - instead of 8 bytes per lane at a
// stride of nlay x 8 bytes, which L2 has to merge and, under the pressure of the record traffic, evicts half
// merged.  first_l0: the block's first packed layer.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool wave_owns_contiguous_block(int col, int nlay, int l0, int nspec, int maxlay, int* first_l0) {
  const int lane = (int)(threadIdx.x & 63);
  const bool full = __popcll(__ballot(1)) == 64;
  const int col0 = __builtin_amdgcn_readfirstlane(col), l00 = __builtin_amdgcn_readfirstlane(l0);
  const bool mine = nspec == 1 && nlay == maxlay && col == col0 + (lane >> 1) && l0 == l00 + (lane >> 1) * nlay;
  *first_l0 = l00;
  return full && __all(mine ? 1 : 0);
}
#endif

// Orders above SS_UNROLL_MAX run four waves per SIMD: no LDS to spare, nothing staged
template <>
struct LaneLds<false> {
  static constexpr int kSlots = 0;
  SS_HD void put(int, double) const {}
  SS_HD double get(int) const { return 0.0; }
  SS_HD void put_next(int, double) const {}
  SS_HD double get_next(int) const { return 0.0; }
};

// doubles per (unit, layer) of the input pack (ss_pack.hpp)
constexpr int kPackSw = 16, kPackLw = 18;
constexpr double kPackNoLayer = -2.0;   // element 15 of a shortwave pack record (else cos_sza), element 0 of a
                                        // longwave one (else dz): the column has no such layer

constexpr double kPi = 3.14159265358979323846;   // radtool/radiation_constants.F90:24
constexpr double kStefanBoltzmann = 5.67037321e-8;  // radiation_constants.F90:26
constexpr int kDevMaxStreams = 8;                // instantiated stream counts: 1,2,4,8

// Legendre-Gauss table as kernel constant (radtool_legendre_gauss.F90:25-48)
struct LG {
  double mu[kDevMaxStreams];
  double tan_ang[kDevMaxStreams];
  double weight[kDevMaxStreams];
  double hweight[kDevMaxStreams];
  double vweight[kDevMaxStreams];
  double sqrt_wmu[kDevMaxStreams];   // sqrt(weight * mu): stream part of the symmetrising scaling (ss_eigen_sym.hpp)
  double rsqrt_wmu[kDevMaxStreams];  // its reciprocal
  double vadjustment;
  double vadjustment2;
};

// Scalars of config_type the solvers read (radsurf_config.F90:32-113)
struct Cfg {
  double min_vegetation_fraction;
  double min_building_fraction;
  double vegetation_isolation_factor;   // of the tile kind (forest / urban)
  int use_symmetric_vegetation_scale;   // of the tile kind
  int do_vegetation;
  // diagnostic (env SPARTACUS_DECLINE_COOP=1): every full problem of the orders 7..16 is declined by the
  // cooperative route and takes the second pass of the layer kernel (one lane, general eigen-solver) -
  // the path of units without the symmetric structure, which well-formed inputs hardly ever reach
  int decline_cooperative;              // (2: the cooperative route runs and is then treated as not converged)
  // diagnostic (env SPARTACUS_DECLINE_SYM=1): every unit of the orders 3..6 is declined by the symmetrised route and
  // takes the out-of-line general route (layer_unit_sw_cold / layer_unit_lw_cold) - the path of units whose matrices
  // are not positive definite, which no input of the reference's test-suite reaches
  int decline_symmetric;
};

#if defined(SS_NO_DECLINE_SYM_SWITCH)      // (experiment builds: what the diagnostic switch costs the hot path)
#define SS_DECLINE_SYM(L) false
#else
#define SS_DECLINE_SYM(L) ((L).cfg.decline_symmetric != 0)
#endif

// The launch descriptor of the running kernel for an OUT-OF-LINE callee.  Every kernel of the solver takes the
// descriptor as its only argument, by value, so it IS the kernel-argument segment.  The pointer has to be formed in
// the kernel (after inlining, that is where this is called): inside a callee __builtin_amdgcn_kernarg_segment_ptr()
// folds to NULL - the implicit argument is not passed on - and the callee reads its descriptor from address 0
// ("Memory access fault ... on address (nil)": what the cold routes did from round 3 until an edge-input sweep
// reached them in round 4).  Handing the callee &L of a by-value kernel argument instead makes every wave copy the
// 800-byte descriptor to scratch at kernel entry (measured in round 3: the write traffic of the layer kernel doubled).
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_LAUNCH_PTR_FOR_CALLEE(L) ((const ::ss::Launch*)__builtin_amdgcn_kernarg_segment_ptr())
#else
#define SS_LAUNCH_PTR_FOR_CALLEE(L) (&(L))
#endif

// canopy_flux_type members by index (order of radsurf_canopy_flux.F90:27-91,
// = order of ss_flux in the C-ABI header)
enum FluxField {
  F_GROUND_DN = 0, F_GROUND_DN_DIR, F_GROUND_NET, F_TOP_DN, F_TOP_DN_DIR, F_TOP_NET,
  F_ROOF_IN, F_ROOF_IN_DIR, F_ROOF_NET, F_WALL_IN, F_WALL_IN_DIR, F_WALL_NET,
  F_ROOF_SUNLIT_FRAC, F_WALL_SUNLIT_FRAC, F_GROUND_SUNLIT_FRAC, F_GROUND_VERTICAL_DIFF,
  F_CLEAR_AIR_ABS, F_VEG_ABS, F_VEG_AIR_ABS, F_VEG_ABS_DIR, F_VEG_SUNLIT_FRAC,
  F_FLUX_DN_LAYER_TOP, F_FLUX_DN_DIR_LAYER_TOP, F_FLUX_UP_LAYER_TOP,
  F_FLUX_DN_LAYER_BASE, F_FLUX_DN_DIR_LAYER_BASE, F_FLUX_UP_LAYER_BASE,
  F_NFIELDS
};

struct FluxDev {
  double* f[F_NFIELDS];   // nullptr = member not allocated
};

// Device mirrors of the input arrays, same layout as the host (Fortran) arrays
struct Inputs {
  const int32_t* nlay;        // (ncol)
  const int32_t* istartlay;   // (ncol) 1-based
  const double* cos_sza;      // (ncol)
  const double* dz;
  const double* building_fraction;
  const double* veg_fraction;
  const double* building_scale;
  const double* veg_scale;
  const double* veg_ext;
  const double* veg_fsd;
  const double* veg_contact_fraction;
  // spectral, (nspec, ntotlay) or (nspec, ncol)
  const double* air_ext;
  const double* air_ssa;
  const double* veg_ssa;
  // SW
  const double* ground_albedo;
  const double* ground_albedo_dir;   // resolved: = ground_albedo unless use_sw_direct_albedo
  const double* roof_albedo;
  const double* roof_albedo_dir;     // resolved: = roof_albedo when not allocated
  const double* wall_albedo;
  const double* wall_specular_frac;
  // LW
  const double* clear_air_planck;
  const double* veg_planck;
  const double* veg_air_planck;
  const double* ground_emissivity;
  const double* roof_emissivity;
  const double* wall_emissivity;
  const double* ground_emission;
  const double* roof_emission;
  const double* wall_emission;
};

// Everything one launch of a layer / column kernel needs (passed by value).
struct Launch {
  Inputs in;
  LG lg;
  Cfg cfg;
  const int32_t* cols;   // column indices (0-based) of this chunk of the class
  int ncols;             // columns in the chunk
  int nspec;             // spectral bands (nsw or nlw)
  int ncu;               // column units = ncols * nspec
  int maxlay;            // largest nlay in the chunk
  int lay_nelem, col_nelem;   // record sizes the plan sized the scratch for (checked at launch)
  double* lay;           // layer-matrix scratch  [tile][layer][elem pair][lane][2]
  double* col;           // adding-method scratch, same tiling
  double* eig;           // eigen records of the cooperative eigen-solver (ss_eigen_coop.hpp), same tiling
  int eig_nelem;
  // Dense list of the cooperative unit-layers of this launch (coop_list_push, ss_eigen_coop.hpp): [0] = their
  // number, [1 + m] = (layer << 32) | column unit.  Zeroed before the first pass, filled by it, read by the factor,
  // Jacobi and layer kernels of the orders 7..16; null: those kernels find their units by position.
  unsigned long long* coop_list;
  // Per-layer inputs of the class, re-laid out once per upload in the tiling of the
  // scratch records (ss_pack.hpp): unit cu of this launch is unit pack_cu0 + cu of the pack
  const double* pack;
  int pack_maxlay;       // layer slots per tile of the pack (largest nlay of the class)
  int pack_cu0;
  double* top1;          // bc_out: sw_albedo / lw_emissivity   (nspec, ncol)
  double* top2;          // bc_out: sw_albedo_dir / lw_emission (nspec, ncol)
  FluxDev flux_a;        // sw_norm_dir  / lw_internal
  FluxDev flux_b;        // sw_norm_diff / lw_norm
  unsigned long long* eigen_fail;   // count of non-converged eigenproblems
};
// The launch descriptor as the kernel-argument segment holds it, through a pointer the optimiser cannot see
// through: what is read from it HERE is loaded here (s_load from the constant address space) instead of being
// hoisted to the kernel entry and kept - or spilled, lane by lane, into VGPRs - across a loop that only needs it
// at its end.  The downward kernels read the 54 member pointers of the two flux objects this way; every kernel
// that uses it takes the descriptor as its only argument, by value.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) Launch* ss_kernarg_launch_ptr;
#define SS_LAUNCH_HERE(L, name) ss_kernarg_launch_ptr name##_p_ = (ss_kernarg_launch_ptr)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(name##_p_)); \
  const __attribute__((address_space(4))) Launch& name = *name##_p_
// the same as a plain reference (what functions that take const Launch& / const LG& accept); after inlining the
// loads are still scalar loads from the constant address space
#define SS_LAUNCH_HERE_REF(L, name) ss_kernarg_launch_ptr name##_p_ = (ss_kernarg_launch_ptr)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(name##_p_)); \
  const Launch& name = *(const Launch*)name##_p_
#else
#define SS_LAUNCH_HERE(L, name) const Launch& name = (L)
#define SS_LAUNCH_HERE_REF(L, name) const Launch& name = (L)
#endif

// Counter slot (of the buffer behind Launch::eigen_fail, zeroed at the start of every solve) that the
// first pass of the layer kernel of the cooperative orders sets when it meets a unit the cooperative
// route does not take (a sub-problem, or no symmetric structure): as long as it is zero the second
// pass (one lane per unit) has nothing to do and its blocks leave at once.
constexpr int kSlotOneLaneUnits = 31;

// In-kernel section stamps (profiling builds only, -DSS_STAMPS): lane 0 of every
// wave adds the s_memtime ticks spent since the previous stamp to counter k of
// the buffer behind Launch::eigen_fail (slot 0 stays the eigen-failure count;
// print with SPARTACUS_PRINT_STAMPS=1).  The stamps perturb the schedule
// (k_layer +45 %), read them as proportions.
#if defined(SS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define SS_STAMP_DECL(ptr) unsigned long long* ss_stamp_buf_ = (ptr); \
  unsigned long long ss_stamp_t_ = __builtin_amdgcn_s_memtime()
#define SS_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (ss_stamp_buf_ && (threadIdx.x & 63) == 0) atomicAdd(ss_stamp_buf_ + 1 + (k), t_ - ss_stamp_t_); \
    ss_stamp_t_ = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SS_STAMP_DECL(ptr) do {} while (0)
#define SS_STAMP(k) do {} while (0)
#endif

// element (i,j) of a column-major matrix with R rows
#define SS_E(a, i, j, R) ((a)[(i) + (j) * (R)])

SS_HD double ss_sign(double a, double b) { return copysign(fabs(a), b); }
SS_HD double ss_max(double a, double b) { return a > b ? a : b; }   // Fortran max
SS_HD double ss_min(double a, double b) { return a < b ? a : b; }

// 1 / x and sqrt(x), 1 / sqrt(x) from the hardware estimates and two Newton / Goldschmidt steps
// (relative error of a few 1e-16, no denormal or division-by-zero fix-up).  For the kernels of the
// orders 7..16: rotation angles of the Jacobi kernel (a rotation built from cs = 1 / sqrt(1 + t^2),
// sn = cs t is orthogonal to rounding whatever the error of t), reciprocal pivots of the unpivoted LU and
// of the symmetric elimination (ss_row16.hpp, ss_layer16.hpp) - an IEEE division is 12 instructions, a
// square root plus a division 40.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return fma(r, fma(-x, r, 1.0), r);
}
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& rs) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  s = fma(g, e, g);
  rs = 2.0 * fma(h, e, h);
}
#else
inline double fast_rcp(double x) { return 1.0 / x; }
inline void fast_sqrt_rsqrt(double x, double& s, double& rs) { s = std::sqrt(x); rs = 1.0 / s; }
#endif

}  // namespace ss
